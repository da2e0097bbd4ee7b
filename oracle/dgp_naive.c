/* dgp_naive.c -- TEST INFRASTRUCTURE ONLY (part of oracle/).
 *
 * Independent, loop-level restatement of the hot-path operators with double accumulation.  It shares
 * no code with oracle/dgp_oracle.py (which leans on torch-CPU convolutions) and exists to cross-check
 * that restatement at small sizes: TF padding arithmetic, no-flip cross-correlation, transposed-conv
 * index map, TF SAME max-pool, DGP soft-argmax.
 *
 * Reference semantics followed (relative to the reference root):
 *   naive_conv2d            slim.conv2d / resnet_utils.conv2d_same as called from
 *                           src/DeepLabCut/deeplabcut/pose_estimation_tensorflow/nnet/pose_net.py:46-52
 *   naive_conv2d_transpose  slim.conv2d_transpose(3x3, stride 2, SAME)   pose_net.py:18-26,
 *                           src/deepgraphpose/models/fitdgp_util.py:58-73
 *   naive_maxpool_same      slim.max_pool2d(3, 2, SAME) of resnet_v1's root block
 *   naive_soft_argmax       src/deepgraphpose/models/fitdgp_util.py:281-402
 * Parity status: unpinned against TF itself (TF is not installable here); see oracle/dgp_oracle.py.
 */
#include <math.h>
#include <stdlib.h>

/* x NHWC [N,H,W,Ci], w HWIO [KH,KW,Ci,Co], zero padding (pad_t, pad_l) before; y [N,Ho,Wo,Co] */
void naive_conv2d(const float* x, int N, int H, int W, int Ci, const float* w, int KH, int KW, int Co, int stride,
                  int rate, int pad_t, int pad_l, int Ho, int Wo, float* y) {
    for (int n = 0; n < N; ++n)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo)
                for (int co = 0; co < Co; ++co) {
                    double acc = 0.0;
                    for (int kh = 0; kh < KH; ++kh) {
                        const int hi = ho * stride - pad_t + kh * rate;
                        if (hi < 0 || hi >= H) continue;
                        for (int kw = 0; kw < KW; ++kw) {
                            const int wi = wo * stride - pad_l + kw * rate;
                            if (wi < 0 || wi >= W) continue;
                            const float* xp = x + (((size_t)n * H + hi) * W + wi) * Ci;
                            const float* wp = w + ((size_t)(kh * KW + kw) * Ci) * Co + co;
                            for (int ci = 0; ci < Ci; ++ci) acc += (double)xp[ci] * (double)wp[(size_t)ci * Co];
                        }
                    }
                    y[(((size_t)n * Ho + ho) * Wo + wo) * Co + co] = (float)acc;
                }
}

/* TF conv2d_transpose, SAME: gradient of the stride-s SAME forward conv on a (s*H x s*W) image.
 * x [N,H,W,Ci], w [KH,KW,Co,Ci], y [N,s*H,s*W,Co]:  y[o] += x[i]*w[k] for every (i,k) with s*i + k - pad = o */
void naive_conv2d_transpose(const float* x, int N, int H, int W, int Ci, const float* w, int KH, int KW, int Co,
                            const float* bias, int s, float* y) {
    const int OH = s * H, OW = s * W;
    int tot_h = (H - 1) * s + KH - OH; if (tot_h < 0) tot_h = 0;
    int tot_w = (W - 1) * s + KW - OW; if (tot_w < 0) tot_w = 0;
    const int pt = tot_h / 2, pl = tot_w / 2;
    double* acc = (double*)calloc((size_t)N * OH * OW * Co, sizeof(double));
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j)
                for (int kh = 0; kh < KH; ++kh) {
                    const int oh = s * i + kh - pt;
                    if (oh < 0 || oh >= OH) continue;
                    for (int kw = 0; kw < KW; ++kw) {
                        const int ow = s * j + kw - pl;
                        if (ow < 0 || ow >= OW) continue;
                        const float* xp = x + (((size_t)n * H + i) * W + j) * Ci;
                        for (int co = 0; co < Co; ++co) {
                            const float* wp = w + ((size_t)(kh * KW + kw) * Co + co) * Ci;
                            double a = 0.0;
                            for (int ci = 0; ci < Ci; ++ci) a += (double)xp[ci] * (double)wp[ci];
                            acc[(((size_t)n * OH + oh) * OW + ow) * Co + co] += a;
                        }
                    }
                }
    for (size_t e = 0; e < (size_t)N * OH * OW * Co; ++e) y[e] = (float)(acc[e] + (bias ? (double)bias[e % Co] : 0.0));
    free(acc);
}

void naive_maxpool_same(const float* x, int N, int H, int W, int C, int k, int s, float* y) {
    const int Ho = (H + s - 1) / s, Wo = (W + s - 1) / s;
    int th = (Ho - 1) * s + k - H; if (th < 0) th = 0;
    int tw = (Wo - 1) * s + k - W; if (tw < 0) tw = 0;
    const int pt = th / 2, pl = tw / 2;
    for (int n = 0; n < N; ++n)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo)
                for (int c = 0; c < C; ++c) {
                    float m = -INFINITY;
                    for (int a = 0; a < k; ++a)
                        for (int b = 0; b < k; ++b) {
                            const int hi = ho * s - pt + a, wi = wo * s - pl + b;
                            if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                            const float v = x[(((size_t)n * H + hi) * W + wi) * C + c];
                            if (v > m) m = v;
                        }
                    y[(((size_t)n * Ho + ho) * Wo + wo) * C + c] = m;
                }
}

/* scmap [N,H,W,C] -> mu [N,C,2] (row, col); softmax(gamma*s) -> zero-padded Gaussian blur (sigma =
 * radius = gauss_len) -> renormalise -> expectation.  All in double. */
void naive_soft_argmax(const float* s, int N, int H, int W, int C, double gamma, int glen, double* mu) {
    const int r = glen, K = 2 * r + 1;
    double* g = (double*)malloc(sizeof(double) * K);
    double gs = 0.0;
    for (int i = -r; i <= r; ++i) { g[i + r] = exp(-0.5 * ((double)i / glen) * ((double)i / glen)); gs += g[i + r]; }
    for (int i = 0; i < K; ++i) g[i] /= gs;
    double* p = (double*)malloc(sizeof(double) * H * W);
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            double mx = -INFINITY, se = 0.0;
            for (int i = 0; i < H * W; ++i) { const double v = gamma * s[((size_t)n * H * W + i) * C + c]; if (v > mx) mx = v; }
            for (int i = 0; i < H * W; ++i) { p[i] = exp(gamma * s[((size_t)n * H * W + i) * C + c] - mx); se += p[i]; }
            for (int i = 0; i < H * W; ++i) p[i] /= se;
            double t0 = 0, th = 0, tw = 0;
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    double b = 0.0;
                    for (int a = -r; a <= r; ++a)
                        for (int d = -r; d <= r; ++d) {
                            const int hh = h + a, ww = w + d;
                            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
                            b += g[a + r] * g[d + r] * p[hh * W + ww];
                        }
                    t0 += b; th += b * h; tw += b * w;
                }
            mu[((size_t)n * C + c) * 2 + 0] = th / t0;
            mu[((size_t)n * C + c) * 2 + 1] = tw / t0;
        }
    free(p); free(g);
}
