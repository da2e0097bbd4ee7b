// DGP loss: forward + backward w.r.t. the two head outputs, as wavefront-reduction kernels (gfx950).
//
// Restates dgp_loss (DGP/models/fitdgp.py:946-1076) on [nt, H, W, nj] scoremaps:
//   mu = soft-argmax(pred)                      fitdgp_util.py:342-402   (kernel in dgp_kernels.hip)
//   t  = labels for visible markers, mu for hidden ones (no stop_gradient)      fitdgp.py:946-961
//   G  = exp(-|alpha - t|^2 / 2l^2) / (max G + 1e-5)                           :964-976
//   visible CE, hidden CE (gm2 in {0,1,2}, gm3 in {0,3})                       :979-1039
//   locref Huber on visible markers                                            :1041-1055
//   spatial clique ("skeleton graph smoothness")                               :1060-1076
// and their gradient d total / d pred, d total / d locref_pred, including the path through the hidden
// targets (Gaussian target + clique -> mu -> softmax).  Maps are tiny (<= 14 400 px), so every kernel is one
// workgroup per marker map with LDS-resident probabilities and fp64 wave reductions (larger maps stream: loss_ce_backward<true>).
//
//   temporal clique ("temporal graph smoothness", wt > 0)                     :1079-1124
// The temporal term follows the reference's specification of intent (its TF code passes float crop sizes to
// tf.image.crop_and_resize and cannot build): frame-to-frame displacement of every marker,
// weighted by min(1/mean-flow, 1)^3 of the optical-flow magnitude inside the +-10 px box of the two positions
// (bilinear crop_and_resize to the full frame, then mean), Frobenius norm.  The flow weight is treated as a
// constant in the backward pass (no gradient through the crop boxes).
#include "dgp_internal.h"

namespace dgp {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmaxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide helpers for 256 threads; `red` is 8 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wsum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wmaxf(v);
    __syncthreads();
    if (lane == 0) red[wave] = (double)v;
    __syncthreads();
    return fmaxf(fmaxf((float)red[0], (float)red[1]), fmaxf((float)red[2], (float)red[3]));
}

// the same for NW waves (loss_ce_backward runs 16: a marker map is ONE workgroup on the critical path between the forward and the backward pass);
// `red` is NW doubles, summed in wave order
template <int NW>
__device__ __forceinline__ double block_sum_n(double v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wsum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += red[w];
    return t;
}
// K sums with ONE pair of barriers (red: K x NW doubles); each sum is formed exactly as block_sum_n forms it
template <int NW, int K>
__device__ __forceinline__ void block_sums_n(double (&v)[K], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wsum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[k * NW + wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t = red[k * NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) t += red[k * NW + w];
        v[k] = t;
    }
}
template <int NW>
__device__ __forceinline__ float block_max_n(float v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wmaxf(v);
    __syncthreads();
    if (lane == 0) red[wave] = (double)v;
    __syncthreads();
    float t = (float)red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = fmaxf(t, (float)red[w]);
    return t;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float ce_logits(float z, float x) {      // tf.nn.sigmoid_cross_entropy_with_logits
    return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

// Out-of-line on purpose (loss_ce_backward): every call site executes the SAME instructions, so a value recomputed in a later pass is bit
// for bit the value an earlier pass took the maximum of.
__device__ __attribute__((noinline)) float loss_gauss_cell(float h, float w, float t0, float t1, float inv2l2) {
    const float dh = h - t0, dw = w - t1;
    return expf(-(dh * dh + dw * dw) * inv2l2);
}
__device__ __attribute__((noinline)) float loss_sigmoid_cell(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __attribute__((noinline)) float loss_softmax_cell(float x, float gamma, float mx) {      // exp(gamma x - max): the product is rounded first
    float v = x * gamma;
    asm volatile("" : "+v"(v));
    return expf(v - mx);
}

// ------------------------------------------------------------------------------------------------
// K1: marker assembly + spatial clique (single workgroup).
//   t_all[m] = label (visible) | mu (hidden); kind[m] = 0 visible / 1 hidden
//   ws_loss and dL/dt (clique part) for every marker.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_assemble_clique(LossArgs a) {
    __shared__ double red[8];
    const int nm = a.nt * a.nj;
    for (int m = threadIdx.x; m < nm; m += 256) {
        a.kind[m] = -1;
        a.t_all[2 * m] = 0.f; a.t_all[2 * m + 1] = 0.f;
        a.dLdt[2 * m] = 0.f; a.dLdt[2 * m + 1] = 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_h; i += 256) {
        const int m = a.hidden_marker[i];
        a.kind[m] = 1;
        a.t_all[2 * m] = a.mu[2 * m]; a.t_all[2 * m + 1] = a.mu[2 * m + 1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_v; i += 256) {
        const int m = a.visible_marker[i], s = a.visible_in_targets[i];
        a.kind[m] = 0;
        a.t_all[2 * m] += a.targets[2 * s]; a.t_all[2 * m + 1] += a.targets[2 * s + 1];   // scatter_nd adds
    }
    __syncthreads();
    double loss = 0.0;
    if (a.nl > 0) {
        const float C = a.clique_scale;      // 1/(H W) * n_vis_tot / n_vis_batch / (n_vis_tot + n_hid_tot) / wn_visible
        for (int e = threadIdx.x; e < a.nl * a.nt; e += 256) {
            const int l = e / a.nt, n = e - l * a.nt;
            float d0 = 0.f, d1 = 0.f;
            for (int j = 0; j < a.nj; ++j) {
                const float s = a.S0[l * a.nj + j];
                if (s != 0.f) {
                    d0 += s * (a.t_all[2 * (n * a.nj + j)] * a.stride + 0.5f * a.stride);
                    d1 += s * (a.t_all[2 * (n * a.nj + j) + 1] * a.stride + 0.5f * a.stride);
                }
            }
            const float dist = sqrtf(d0 * d0 + d1 * d1);
            const float wmax = a.ws_max[l], wsl = a.ws[l];
            loss += (double)((fmaxf(dist - wmax, 0.f) + wmax) * wsl * C);
            if (dist > wmax) {
                const float g0 = C * wsl * d0 / dist * a.stride, g1 = C * wsl * d1 / dist * a.stride;
                for (int j = 0; j < a.nj; ++j) {
                    const float s = a.S0[l * a.nj + j];
                    if (s != 0.f) {
                        atomicAdd(&a.dLdt[2 * (n * a.nj + j)], s * g0);
                        atomicAdd(&a.dLdt[2 * (n * a.nj + j) + 1], s * g1);
                    }
                }
            }
        }
    }
    loss = block_sum(loss, red);
    if (threadIdx.x == 0) a.losses[3] = (float)loss;     // ws_loss
}

// ------------------------------------------------------------------------------------------------
// K1b: temporal-clique flow weights, one workgroup per (frame pair, joint).
//   box = bbox of the two marker positions (px) +- 10, clipped to the frame; tf.image.crop_and_resize
//   (bilinear, extrapolation 0) of the flow-magnitude field to [Hin, Win]; mean; w = min(min(1/(m+1e-10),1)^3,1)
//   * wt_batch[t] / H / W                                                          fitdgp.py:1085-1118
// The boxes are functions of the (differentiable) hidden targets and TF's crop_and_resize has a gradient with respect to them
// (CropAndResizeGradBoxes: d sample / d in_y = bottom - top, d in_y / d y1 = (Hin - 1) - iy, d in_y / d y2 = iy at crop height
// = image height; likewise in x), so the weight is NOT a constant of the backward pass: wt_w[ne + 4 e + {0,1,2,3}] =
// dw / d (r0, c0, r1, c1), through min / max of the two positions (a tie shares the gradient evenly, like tf.reduce_min), the
// clamps to the frame (no gradient where the clamp is active) and min(., 1)^3 (none while the mean flow is <= 1).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_temporal_weights(LossArgs a) {
    __shared__ double red[8];
    const int e = blockIdx.x, n = e / a.nj, j = e - n * a.nj;
    const int ne = (a.nt - 1) * a.nj;
    const float* vf = a.vector_field + (long long)n * a.Hin * a.Win;
    const float r0 = a.t_all[2 * (n * a.nj + j)] * a.stride + 0.5f * a.stride;
    const float c0 = a.t_all[2 * (n * a.nj + j) + 1] * a.stride + 0.5f * a.stride;
    const float r1 = a.t_all[2 * ((n + 1) * a.nj + j)] * a.stride + 0.5f * a.stride;
    const float c1 = a.t_all[2 * ((n + 1) * a.nj + j) + 1] * a.stride + 0.5f * a.stride;
    const float win = 10.f, Hf = (float)a.Hin, Wf = (float)a.Win;
    const float y1 = fmaxf(0.f, fminf(r0, r1) - win) / Hf, y2 = fminf(Hf, fmaxf(r0, r1) + win) / Hf;
    const float x1 = fmaxf(0.f, fminf(c0, c1) - win) / Wf, x2 = fminf(Wf, fmaxf(c0, c1) + win) / Wf;
    const float hs = a.Hin > 1 ? (y2 - y1) * (Hf - 1.f) / (Hf - 1.f) : 0.f;      // crop_h == Hin
    const float wsx = a.Win > 1 ? (x2 - x1) * (Wf - 1.f) / (Wf - 1.f) : 0.f;
    double acc = 0.0, gy1 = 0.0, gy2 = 0.0, gx1 = 0.0, gx2 = 0.0;
    const int total = a.Hin * a.Win;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int iy = i / a.Win, ix = i - iy * a.Win;
        const float in_y = a.Hin > 1 ? y1 * (Hf - 1.f) + iy * hs : 0.5f * (y1 + y2) * (Hf - 1.f);
        const float in_x = a.Win > 1 ? x1 * (Wf - 1.f) + ix * wsx : 0.5f * (x1 + x2) * (Wf - 1.f);
        if (in_y < 0.f || in_y > Hf - 1.f || in_x < 0.f || in_x > Wf - 1.f) continue;     // extrapolation_value = 0
        const int ty = (int)floorf(in_y), by = (int)ceilf(in_y), lx = (int)floorf(in_x), rx = (int)ceilf(in_x);
        const float fy = in_y - ty, fx = in_x - lx;
        const float tl = vf[ty * a.Win + lx], tr = vf[ty * a.Win + rx], bl = vf[by * a.Win + lx], br = vf[by * a.Win + rx];
        const float top = tl + (tr - tl) * fx;
        const float bot = bl + (br - bl) * fx;
        acc += (double)(top + (bot - top) * fy);
        const float dvy = bot - top, dvx = (1.f - fy) * (tr - tl) + fy * (br - bl);
        if (a.Hin > 1) { gy1 += (double)(dvy * ((Hf - 1.f) - (float)iy)); gy2 += (double)(dvy * (float)iy); }
        else { gy1 += 0.5 * (double)(dvy * (Hf - 1.f)); gy2 += 0.5 * (double)(dvy * (Hf - 1.f)); }
        if (a.Win > 1) { gx1 += (double)(dvx * ((Wf - 1.f) - (float)ix)); gx2 += (double)(dvx * (float)ix); }
        else { gx1 += 0.5 * (double)(dvx * (Wf - 1.f)); gx2 += 0.5 * (double)(dvx * (Wf - 1.f)); }
    }
    acc = block_sum(acc, red);
    gy1 = block_sum(gy1, red); gy2 = block_sum(gy2, red); gx1 = block_sum(gx1, red); gx2 = block_sum(gx2, red);
    if (threadIdx.x == 0) {
        const float m = (float)(acc / (double)total);
        const float inv0 = 1.f / (m + 1e-10f);
        float inv = fminf(inv0, 1.f);
        inv = fminf(expf(logf(inv) * 3.f), 1.f);
        const float k = a.wt_batch[n] / (float)a.H / (float)a.W;
        a.wt_w[e] = inv * k;
        // dw / dm: only where 1 / (m + eps) < 1 (both minima pass the gradient to their first argument there)
        const float dwdm = inv0 < 1.f ? -3.f * inv0 * inv0 * inv0 * inv0 * k : 0.f;
        const float sy1 = dwdm * (float)(gy1 / (double)total), sy2 = dwdm * (float)(gy2 / (double)total);
        const float sx1 = dwdm * (float)(gx1 / (double)total), sx2 = dwdm * (float)(gx2 / (double)total);
        // box -> positions: y1 = max(0, min(r0, r1) - win) / Hin, y2 = min(Hin, max(r0, r1) + win) / Hin (x alike)
        const float ay1 = (fminf(r0, r1) - win > 0.f) ? sy1 / Hf : 0.f, ay2 = (fmaxf(r0, r1) + win < Hf) ? sy2 / Hf : 0.f;
        const float ax1 = (fminf(c0, c1) - win > 0.f) ? sx1 / Wf : 0.f, ax2 = (fmaxf(c0, c1) + win < Wf) ? sx2 / Wf : 0.f;
        const float rmin0 = r0 < r1 ? 1.f : (r0 > r1 ? 0.f : 0.5f), cmin0 = c0 < c1 ? 1.f : (c0 > c1 ? 0.f : 0.5f);
        float* g = a.wt_w + ne + 4 * e;
        g[0] = ay1 * rmin0 + ay2 * (1.f - rmin0);            // d w / d r0 (r0 is the max exactly where it is not the min)
        g[1] = ax1 * cmin0 + ax2 * (1.f - cmin0);            // d w / d c0
        g[2] = ay1 * (1.f - rmin0) + ay2 * rmin0;            // d w / d r1
        g[3] = ax1 * (1.f - cmin0) + ax2 * cmin0;            // d w / d c1
    }
}

// K1c: temporal clique loss + d/dt (single workgroup): || (relu(D - wt_max) + wt_max) * w ||_F * scale, with the gradient through
// the distances D AND through the flow weights w (boxes of the crop: see K1b)
__global__ __launch_bounds__(256) void loss_temporal(LossArgs a) {
    __shared__ double red[8];
    const int ne = (a.nt - 1) * a.nj;
    double ss = 0.0;
    for (int e = threadIdx.x; e < ne; e += 256) {
        const int n = e / a.nj, j = e - n * a.nj;
        const float dr = (a.t_all[2 * ((n + 1) * a.nj + j)] - a.t_all[2 * (n * a.nj + j)]) * a.stride;
        const float dc = (a.t_all[2 * ((n + 1) * a.nj + j) + 1] - a.t_all[2 * (n * a.nj + j) + 1]) * a.stride;
        const float D = sqrtf(dr * dr + dc * dc);
        const float v = (fmaxf(D - a.wt_max, 0.f) + a.wt_max) * a.wt_w[e];
        ss += (double)v * v;
    }
    ss = block_sum(ss, red);
    const float F = (float)sqrt(ss);
    const float C = a.temporal_scale;       // n_vis_tot / n_vis_batch / (n_vis_tot + n_hid_tot) / wn_visible
    if (threadIdx.x == 0) a.losses[6] = F * C;
    if (F <= 0.f) return;
    for (int e = threadIdx.x; e < ne; e += 256) {
        const int n = e / a.nj, j = e - n * a.nj;
        const float dr = (a.t_all[2 * ((n + 1) * a.nj + j)] - a.t_all[2 * (n * a.nj + j)]) * a.stride;
        const float dc = (a.t_all[2 * ((n + 1) * a.nj + j) + 1] - a.t_all[2 * (n * a.nj + j) + 1]) * a.stride;
        const float D = sqrtf(dr * dr + dc * dc);
        const float w = a.wt_w[e];
        const float gD = fmaxf(D - a.wt_max, 0.f) + a.wt_max;
        // through the weight: dL/dw = C v / F * gD, positions = t * stride + stride / 2
        const float dLdw = C * (gD * w) / F * gD * a.stride;
        const float* gw = a.wt_w + ne + 4 * e;
        if (dLdw != 0.f) {
            if (gw[0] != 0.f) atomicAdd(&a.dLdt[2 * (n * a.nj + j)], dLdw * gw[0]);
            if (gw[1] != 0.f) atomicAdd(&a.dLdt[2 * (n * a.nj + j) + 1], dLdw * gw[1]);
            if (gw[2] != 0.f) atomicAdd(&a.dLdt[2 * ((n + 1) * a.nj + j)], dLdw * gw[2]);
            if (gw[3] != 0.f) atomicAdd(&a.dLdt[2 * ((n + 1) * a.nj + j) + 1], dLdw * gw[3]);
        }
        if (!(D > a.wt_max) || D <= 0.f) continue;
        const float v = D * w;
        const float g = C * v / F * w / D * a.stride;        // dL/dD * dD/d(dr) = g * dr
        atomicAdd(&a.dLdt[2 * ((n + 1) * a.nj + j)], g * dr);
        atomicAdd(&a.dLdt[2 * ((n + 1) * a.nj + j) + 1], g * dc);
        atomicAdd(&a.dLdt[2 * (n * a.nj + j)], -g * dr);
        atomicAdd(&a.dLdt[2 * (n * a.nj + j) + 1], -g * dc);
    }
}

// ------------------------------------------------------------------------------------------------
// K2: c = max sigmoid(x) per marker (needed across markers for the gm3 = 3 normaliser).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loss_marker_stats(LossArgs a) {
    __shared__ double red[8];
    const int m = blockIdx.x, n = m / a.nj, cj = m - n * a.nj;
    const int HW = a.H * a.W;
    const float* x = a.pred + (long long)n * HW * a.nj + cj;
    float qmax = -1.f;
    for (int i = threadIdx.x; i < HW; i += 256) qmax = fmaxf(qmax, loss_sigmoid_cell(x[(long long)i * a.nj]));
    qmax = block_max(qmax, red);
    if (threadIdx.x == 0) a.stats[m] = qmax;
}

// K3: normalisers that need all markers (single workgroup): present = H*W*#{hidden: 1 - c != 0}
__global__ __launch_bounds__(256) void loss_normalisers(LossArgs a) {
    __shared__ double red[8];
    double cnt = 0;
    for (int i = threadIdx.x; i < a.n_h; i += 256)
        if (1.f - a.stats[a.hidden_marker[i]] != 0.f) cnt += 1;
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        const double HW = (double)a.H * a.W;
        a.norm[0] = a.n_v > 0 ? (float)(1.0 / (a.n_v * HW)) : 0.f;                        // visible mean
        a.norm[1] = a.n_h > 0 ? (float)(a.hidden_scale / (a.n_h * HW)) : 0.f;             // hidden, gm3 = 0
        a.norm[2] = cnt > 0 ? (float)(a.hidden_scale / (cnt * HW)) : 0.f;                 // hidden, gm3 = 3
        a.losses[0] = 0.f; a.losses[1] = 0.f; a.losses[2] = 0.f;
        a.norm[3] = 0.f;   // locref nonzero count (published by loss_locref_backward)
    }
}

// ------------------------------------------------------------------------------------------------
// K4: per-marker CE loss + d/dpred (direct) + d/dt, then (hidden markers) back through the soft-argmax.
// G and sigmoid(x) are computed ONCE into LDS so that the "is this the max element" tests of the
// reduce_max gradients (ties share the gradient evenly, like tf.reduce_max) are exact.
// LARGE (round 5): maps beyond 19 200 cells do not fit two LDS arrays (the reference's placeholders are [None, None, None, nj],
// DGP/models/fitdgp.py:1130-1142: any size).  G, sigmoid(x) and the softmax probabilities are then RECOMPUTED wherever they are read --
// by the same out-of-line functions on the same inputs, so a recomputed value is bit for bit the value the first pass took the maximum
// of and the equality tests stay exact; every output equals what the LDS variant gives (tests force both on one map).  A fallback for
// frames beyond ~960 x 1280, not a fast path: (2 gauss_len + 1)^2 exps per cell in the blur.
// ------------------------------------------------------------------------------------------------
constexpr int LOSS_CE_THREADS = 1024;      // (256 until round 6: 107 us per step for 44 maps of 60 x 80, every pass 19 iterations per thread)
template <bool LARGE>
__global__ __launch_bounds__(LOSS_CE_THREADS) void loss_ce_backward(LossArgs a) {
    constexpr int NT = LOSS_CE_THREADS, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sG = reinterpret_cast<float*>(smem);      // H*W Gaussian bump (later: softmax probabilities)
    float* sq = sG + (LARGE ? 0 : a.H * a.W);        // H*W sigmoid(x)
    __shared__ double red[3 * NW];
    __shared__ float gk[16];
    const int m = blockIdx.x, n = m / a.nj, cj = m - n * a.nj;
    const int kind = a.kind[m];
    const int HW = a.H * a.W;
    const long long base = (long long)n * HW * a.nj + cj;
    const float* x = a.pred + base;
    float* dx = a.dpred + base;
    if (kind < 0) {       // marker in neither list: no loss term
        for (int i = threadIdx.x; i < HW; i += NT) dx[(long long)i * a.nj] = 0.f;
        return;
    }
    const float t0 = a.t_all[2 * m], t1 = a.t_all[2 * m + 1];
    const float l2 = a.lengthscale * a.lengthscale, inv2l2 = 1.f / (2.f * l2);
    const bool hidden = kind == 1;
    const bool use_c = hidden && a.gm2 != 0;           // confidence c enters the hidden term
    const bool scale_target = hidden && a.gm2 == 1;
    const bool scaled_logit = hidden && a.gm3 == 3;
    const float e20 = 1e-20f;

    auto Gat = [&](int i) -> float { if (!LARGE) return sG[i]; const int h = i / a.W; return loss_gauss_cell((float)h, (float)(i - h * a.W), t0, t1, inv2l2); };
    auto Qat = [&](int i) -> float { return LARGE ? loss_sigmoid_cell(x[(long long)i * a.nj]) : sq[i]; };
    float gmax = -1.f, c = -1.f;
    for (int i = threadIdx.x; i < HW; i += NT) {
        const int h = i / a.W, w = i - h * a.W;
        const float G = loss_gauss_cell((float)h, (float)w, t0, t1, inv2l2);
        const float q = loss_sigmoid_cell(x[(long long)i * a.nj]);
        if (!LARGE) { sG[i] = G; sq[i] = q; }
        gmax = fmaxf(gmax, G); c = fmaxf(c, q);
    }
    gmax = block_max_n<NW>(gmax, red);
    c = block_max_n<NW>(c, red);
    double ngq[2] = {0, 0};
    for (int i = threadIdx.x; i < HW; i += NT) { if (Gat(i) == gmax) ngq[0] += 1; if (Qat(i) == c) ngq[1] += 1; }
    block_sums_n<NW>(ngq, red);
    const float ng = (float)ngq[0], nq = (float)ngq[1];
    const float gden = gmax + 1e-5f;
    const float A = !hidden ? a.norm[0] : (scaled_logit ? a.norm[2] * (1.f - c) : a.norm[1]);
    const float Araw = !hidden ? a.norm[0] : (scaled_logit ? a.norm[2] : a.norm[1]);

    // pass 1: loss value, dL/dc, dL/dGmax accumulators
    double ce_sum = 0, dLdc = 0, dLdgmax = 0;
    for (int i = threadIdx.x; i < HW; i += NT) {
        const float G = Gat(i), g = G / gden, q = Qat(i);
        const float xi = x[(long long)i * a.nj];
        const float z = scale_target ? g * c : g;
        float xe = xi, dxe_ds = 0.f;
        if (scaled_logit) {
            const float sv = q * c;
            xe = -logf(1.f - sv + e20) + logf(sv + e20);
            dxe_ds = 1.f / (sv + e20) + 1.f / (1.f - sv + e20);
        }
        ce_sum += (double)ce_logits(z, xe);
        const float u = A * (sigmoidf(xe) - z);            // dL/dxe
        const float dLdz = A * (-xe);
        if (use_c) {
            if (scaled_logit) dLdc += (double)(u * dxe_ds * q);
            if (scale_target) dLdc += (double)(dLdz * g);
        }
        const float dLdg = scale_target ? dLdz * c : dLdz;
        dLdgmax += (double)(-dLdg * G / (gden * gden));
    }
    {
        double v3[3] = {ce_sum, dLdc, dLdgmax};
        block_sums_n<NW>(v3, red);
        ce_sum = v3[0]; dLdc = v3[1]; dLdgmax = v3[2];
    }
    if (scaled_logit) dLdc += -(double)Araw * ce_sum;      // d/dc of the (1 - c) weight
    if (threadIdx.x == 0) atomicAdd(&a.losses[hidden ? 1 : 0], (float)((double)A * ce_sum));

    // pass 2: dL/dx (direct) and dL/dt
    double gt0 = 0, gt1 = 0;
    for (int i = threadIdx.x; i < HW; i += NT) {
        const int h = i / a.W, w = i - h * a.W;
        const float dh = (float)h - t0, dw = (float)w - t1;
        const float G = Gat(i), g = G / gden, q = Qat(i);
        const float xi = x[(long long)i * a.nj];
        const float z = scale_target ? g * c : g;
        float xe = xi, dxe_ds = 0.f;
        if (scaled_logit) {
            const float sv = q * c;
            xe = -logf(1.f - sv + e20) + logf(sv + e20);
            dxe_ds = 1.f / (sv + e20) + 1.f / (1.f - sv + e20);
        }
        const float u = A * (sigmoidf(xe) - z);
        float gx;
        if (scaled_logit) {
            float dLdq = u * dxe_ds * c;
            if (q == c) dLdq += (float)(dLdc / nq);
            gx = dLdq * q * (1.f - q);
        } else {
            gx = u;
            if (use_c && q == c) gx += (float)(dLdc / nq) * q * (1.f - q);
        }
        dx[(long long)i * a.nj] = gx;
        if (hidden) {                 // targets of hidden markers are functions of mu
            const float dLdz = A * (-xe);
            const float dLdg = scale_target ? dLdz * c : dLdz;
            float dLdG = dLdg / gden;
            if (G == gmax) dLdG += (float)(dLdgmax / ng);
            const float k = dLdG * G / l2;
            gt0 += (double)(k * dh);
            gt1 += (double)(k * dw);
        }
    }
    if (!hidden) return;
    {
        double v2[2] = {gt0, gt1};
        block_sums_n<NW>(v2, red);
        gt0 = v2[0]; gt1 = v2[1];
    }
    const float g_h = (float)gt0 + a.dLdt[2 * m], g_w = (float)gt1 + a.dLdt[2 * m + 1];

    // ---- back through mu = soft-argmax(x): p = softmax(gamma x), b = blur(p), mu = sum b (h,w) / sum b
    float* sp = sG;
    const int r = a.gauss_len;
    if (threadIdx.x == 0) {
        float sacc = 0.f;
        for (int i = -r; i <= r; ++i) { const float xs = (float)i / (float)a.gauss_len; gk[i + r] = expf(-0.5f * xs * xs); sacc += gk[i + r]; }
        for (int i = 0; i <= 2 * r; ++i) gk[i] /= sacc;
    }
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < HW; i += NT) { float v = x[(long long)i * a.nj] * a.gamma; asm volatile("" : "+v"(v)); mx = fmaxf(mx, v); }
    mx = block_max_n<NW>(mx, red);
    double se = 0;
    for (int i = threadIdx.x; i < HW; i += NT) { const float e = loss_softmax_cell(x[(long long)i * a.nj], a.gamma, mx); if (!LARGE) sp[i] = e; se += (double)e; }
    se = block_sum_n<NW>(se, red);
    const float den = (float)se;
    if (!LARGE) for (int i = threadIdx.x; i < HW; i += NT) sp[i] /= den;
    __syncthreads();
    auto Pat = [&](int i) -> float { return LARGE ? loss_softmax_cell(x[(long long)i * a.nj], a.gamma, mx) / den : sp[i]; };
    double B = 0, Bh = 0, Bw = 0;
    for (int i = threadIdx.x; i < HW; i += NT) {
        const int h = i / a.W, w = i - h * a.W;
        float bsum = 0.f;
        for (int aa = -r; aa <= r; ++aa) {
            const int hh = h + aa;
            if ((unsigned)hh >= (unsigned)a.H) continue;
            for (int bb = -r; bb <= r; ++bb) {
                const int ww = w + bb;
                if ((unsigned)ww >= (unsigned)a.W) continue;
                bsum += gk[aa + r] * gk[bb + r] * Pat(hh * a.W + ww);
            }
        }
        B += (double)bsum; Bh += (double)bsum * h; Bw += (double)bsum * w;
    }
    {
        double v3[3] = {B, Bh, Bw};
        block_sums_n<NW>(v3, red);
        B = v3[0]; Bh = v3[1]; Bw = v3[2];
    }
    const float mu_h = (float)(Bh / B), mu_w = (float)(Bw / B), invB = (float)(1.0 / B);
    // dL/db_i = r_i = (g_h (h - mu_h) + g_w (w - mu_w)) / B inside the map, 0 outside; dL/dp = blur(r)
    auto dLdp = [&](int h, int w) {
        float sacc = 0.f;
        for (int aa = -r; aa <= r; ++aa) {
            const int hh = h + aa;
            if ((unsigned)hh >= (unsigned)a.H) continue;
            for (int bb = -r; bb <= r; ++bb) {
                const int ww = w + bb;
                if ((unsigned)ww >= (unsigned)a.W) continue;
                sacc += gk[aa + r] * gk[bb + r] * (g_h * ((float)hh - mu_h) + g_w * ((float)ww - mu_w)) * invB;
            }
        }
        return sacc;
    };
    double inner = 0;
    for (int i = threadIdx.x; i < HW; i += NT) { const int h = i / a.W, w = i - h * a.W; inner += (double)(Pat(i) * dLdp(h, w)); }
    inner = block_sum_n<NW>(inner, red);
    for (int i = threadIdx.x; i < HW; i += NT) {
        const int h = i / a.W, w = i - h * a.W;
        dx[(long long)i * a.nj] += a.gamma * Pat(i) * (dLdp(h, w) - (float)inner);
    }
}

// ------------------------------------------------------------------------------------------------
// locref Huber (k = 1) on visible markers: count of non-zero mask entries, then loss + gradient.
// One workgroup per visible marker; channel pair (2j, 2j+1) of frame n.
// ------------------------------------------------------------------------------------------------
constexpr int LOSS_LOCREF_THREADS = 1024;
// (round 6: ONE launch.  The count of non-zero mask entries over all visible markers -- small integers, exact in any order -- is formed by every workgroup
//  itself instead of by a launch of its own in front; 1024 threads: a marker is one workgroup on the critical path between forward and backward pass)
__global__ __launch_bounds__(LOSS_LOCREF_THREADS) void loss_locref_backward(LossArgs a) {
    constexpr int NT = LOSS_LOCREF_THREADS, NW = NT / 64;
    __shared__ double red[NW];
    const int HW = a.H * a.W;
    double cnt = 0;
    for (int v = 0; v < a.n_v; ++v) {
        const int mv = a.visible_marker[v], nv = mv / a.nj, cv = mv - nv * a.nj;
        const float* mk = a.locref_mask + (long long)nv * HW * 2 * a.nj + 2 * cv;
        for (int i = threadIdx.x; i < 2 * HW; i += NT) {
            const int px = i >> 1, k = i & 1;
            if (mk[(long long)px * 2 * a.nj + k] != 0.f) cnt += 1;
        }
    }
    const float nz = (float)block_sum_n<NW>(cnt, red);
    if (blockIdx.x == 0 && threadIdx.x == 0) a.norm[3] = nz;
    const int m = a.visible_marker[blockIdx.x], n = m / a.nj, cj = m - n * a.nj;
    const long long base = (long long)n * HW * 2 * a.nj + 2 * cj;
    const float wgt = nz > 0.f ? a.locref_weight / nz : 0.f;
    double ls = 0;
    for (int i = threadIdx.x; i < 2 * HW; i += NT) {
        const int px = i >> 1, k = i & 1;
        const long long o = base + (long long)px * 2 * a.nj + k;
        const float d = a.locref_pred[o] - a.locref_map[o], mk = a.locref_mask[o];
        const float ad = fabsf(d);
        float el, de;
        if (a.huber) { el = ad < 1.f ? 0.5f * d * d : ad - 0.5f; de = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f); }
        else { el = d * d; de = 2.f * d; }
        ls += (double)(el * mk);
        a.dlocref[o] = wgt * mk * de;
    }
    ls = block_sum_n<NW>(ls, red);
    if (threadIdx.x == 0) atomicAdd(&a.losses[2], (float)(ls * wgt));
}

__global__ void loss_finalize(LossArgs a) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        a.losses[4] = a.losses[0] + a.losses[1] + a.losses[2] + (a.nl > 0 ? a.losses[3] : 0.f) +
                      (a.use_wt ? a.losses[6] : 0.f);                                           // total_loss
        a.losses[5] = a.losses[0] + a.losses[2];                                                 // total_loss_visible
    }
}

// ------------------------------------------------------------------------------------------------
// DLC step-0 loss (pose_net.train, DeepLabCut pose_estimation_tensorflow/nnet/pose_net.py:159-190):
//   part_loss   = tf.losses.sigmoid_cross_entropy(targets, logits, weights)  -> sum(w*ce) / #nonzero(w)
//   locref_loss = locref_loss_weight * huber(targets, pred, mask)            -> sum(mask*h) / #nonzero(mask)
// acc[0] = sum w*ce, acc[1] = #nonzero w, acc[2] = sum mask*huber, acc[3] = #nonzero mask   (doubles)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dlc_loss_reduce(DlcLossArgs a) {
    __shared__ double red[4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x, i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = i0; i < a.n_part; i += stride) {
        const float x = a.pred[i], z = a.part_targets[i];
        const float w = a.part_weights ? a.part_weights[i] : 1.f;
        // max(x,0) - x*z + log1p(exp(-|x|))  (tf.nn.sigmoid_cross_entropy_with_logits)
        const float ce = fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
        s0 += (double)(w * ce);
        s1 += w != 0.f ? 1.0 : 0.0;
    }
    if (a.locref_pred)
        for (long long i = i0; i < a.n_loc; i += stride) {
            const float mk = a.locref_mask[i];
            const float d = a.locref_pred[i] - a.locref_targets[i], ad = fabsf(d);
            const float h = a.huber ? (ad < 1.f ? 0.5f * d * d : ad - 0.5f) : d * d;
            s2 += (double)(mk * h);
            s3 += mk != 0.f ? 1.0 : 0.0;
        }
    s0 = block_sum(s0, red); s1 = block_sum(s1, red); s2 = block_sum(s2, red); s3 = block_sum(s3, red);
    if (threadIdx.x == 0) {
        atomicAdd(&a.acc[0], s0); atomicAdd(&a.acc[1], s1); atomicAdd(&a.acc[2], s2); atomicAdd(&a.acc[3], s3);
    }
}

__global__ __launch_bounds__(256) void dlc_loss_backward(DlcLossArgs a) {
    const float inv_p = a.acc[1] > 0.0 ? (float)(1.0 / a.acc[1]) : 0.f;
    const float inv_l = a.acc[3] > 0.0 ? (float)((double)a.locref_loss_weight / a.acc[3]) : 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x, i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = i0; i < a.n_part; i += stride) {
        const float x = a.pred[i], z = a.part_targets[i];
        const float w = a.part_weights ? a.part_weights[i] : 1.f;
        const float sg = 1.f / (1.f + expf(-x));
        a.dpred[i] = w * (sg - z) * inv_p;
    }
    if (a.locref_pred)
        for (long long i = i0; i < a.n_loc; i += stride) {
            const float d = a.locref_pred[i] - a.locref_targets[i];
            const float de = a.huber ? (fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f)) : 2.f * d;
            a.dlocref[i] = a.locref_mask[i] * de * inv_l;
        }
    if (i0 == 0) {
        const float pl = a.acc[1] > 0.0 ? (float)(a.acc[0] / a.acc[1]) : 0.f;
        const float ll = a.acc[3] > 0.0 ? (float)(a.acc[2] / a.acc[3]) * a.locref_loss_weight : 0.f;
        a.losses[0] = pl; a.losses[1] = ll; a.losses[2] = pl + ll; a.losses[3] = 0.f;
    }
}

hipError_t launch_dlc_loss(const DlcLossArgs& a, hipStream_t s) {
    hipError_t e = hipMemsetAsync(a.acc, 0, 4 * sizeof(double), s);
    if (e != hipSuccess) return e;
    long long n = a.n_part > a.n_loc ? a.n_part : a.n_loc;
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    // (every workgroup of the reduction ends in four fp64 atomics on four fixed addresses: one workgroup per CU keeps that queue short -- see sumsq_kernel)
    hipLaunchKernelGGL(dlc_loss_reduce, dim3(grid > 256 ? 256 : grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL(dlc_loss_backward, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_loss(const LossArgs& a, hipStream_t s) {
    const int nm = a.nt * a.nj;
    hipError_t e = hipMemsetAsync(a.dlocref, 0, (size_t)a.nt * a.H * a.W * 2 * a.nj * sizeof(float), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(loss_assemble_clique, dim3(1), dim3(256), 0, s, a);
    if (a.use_wt && a.nt > 1) {
        hipLaunchKernelGGL(loss_temporal_weights, dim3((a.nt - 1) * a.nj), dim3(256), 0, s, a);
        hipLaunchKernelGGL(loss_temporal, dim3(1), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(loss_marker_stats, dim3(nm), dim3(256), 0, s, a);
    hipLaunchKernelGGL(loss_normalisers, dim3(1), dim3(256), 0, s, a);
    const size_t smem = (size_t)2 * a.H * a.W * sizeof(float);
    const char* force = getenv("DGP_LOSS_STREAM");            // tests: the streaming variant on a map the LDS variant also takes (read per call)
    if (smem > LOSS_LDS_LIMIT || (force && atoi(force) != 0)) {
        hipLaunchKernelGGL(loss_ce_backward<true>, dim3(nm), dim3(LOSS_CE_THREADS), 0, s, a);
    } else {
        static bool attr_dev[16] = {};
        bool& attr = attr_dev[dgp_device_slot()];
        if (!attr) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(loss_ce_backward<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e != hipSuccess) return e;
            attr = true;
        }
        hipLaunchKernelGGL(loss_ce_backward<false>, dim3(nm), dim3(LOSS_CE_THREADS), smem, s, a);
    }
    if (a.n_v > 0) {
        hipLaunchKernelGGL(loss_locref_backward, dim3(a.n_v), dim3(LOSS_LOCREF_THREADS), 0, s, a);
    }
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace dgp
