// Training step of the DGP hot path on gfx950: forward with retained activations, backward through the
// heads / ResNet (data gradients through the same implicit-GEMM kernel, weight gradients through an
// fp32-MFMA reduction-over-pixels GEMM), BN-affine / bias gradients, global-norm clip + momentum SGD.
//
// Reference: sess.run([loss, train_op]) in fit_dgp (DGP/models/fitdgp.py:708-713,818): TF autodiff of
// dgp_loss w.r.t. ALL trainable variables (conv weights, BN gamma/beta -- moving statistics stay frozen because
// the net is built with is_training=False, PET/nnet/pose_net.py:52 -- head weights/biases),
// clip_by_global_norm(10), MomentumOptimizer(0.9).
#include "dgp_engine.h"
#include "dgp_device.h"
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>

using namespace dgp;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr unsigned OOBT = 0xFFFFFFF0u;

__device__ __forceinline__ float4 bload16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

// ------------------------------------------------------------------------------------------------
// weight (re)packing on the device, from the flat master parameter buffer
// ------------------------------------------------------------------------------------------------
// forward panels [nk*8][CoutP][4] from HWIO [KH*KW][Cin_real][Cout]
// max |v| of what a pack kernel wrote -> the panel's range slots (fp16-split convs of the training step); non-negative floats
// order like their bit patterns, slots spread the atomics
__device__ __forceinline__ void pack_track(float* rng, float m) {
    if (!rng) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        atomicMax(reinterpret_cast<unsigned*>(rng) + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (ABSMAX_SLOTS - 1)), __float_as_uint(m));
}
__device__ __forceinline__ float amax4(float m, const float4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

__global__ void pack_fwd_kernel(const float* __restrict__ w, int taps, int cin_real, int cin, int cout, int coutP,
                                int nchunks, float* __restrict__ packed, float* __restrict__ rng) {
    float mx = 0.f;
    const long long total = (long long)nchunks * coutP;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(g % coutP);
        const int q = (int)(g / coutP);
        const int cin4 = cin >> 2;
        const int tap = q / cin4, ch = (q - tap * cin4) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tap < taps && co < cout) {
            float* pv = &v.x;
            for (int e = 0; e < 4; ++e)
                if (ch + e < cin_real) pv[e] = w[((long long)tap * cin_real + ch + e) * cout + co];
        }
        *reinterpret_cast<float4*>(packed + g * 4) = v;
        mx = amax4(mx, v);
    }
    pack_track(rng, mx);
}

// head forward panels: W'[khp][kwp][ci][(a,b),c] = w[a+2-2khp][b+2-2kwp][c][ci] (w is [3,3,njt,Cin])
__global__ void pack_head_fwd_kernel(const float* __restrict__ w, int njt, int cin, int coutP, int nchunks,
                                     float* __restrict__ packed) {
    const long long total = (long long)nchunks * coutP;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(g % coutP);
        const int q = (int)(g / coutP);
        const int cin4 = cin >> 2;
        const int tap = q / cin4, ch = (q - tap * cin4) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tap < 4 && co < 4 * njt) {
            const int khp = tap >> 1, kwp = tap & 1, ph = co / njt, c = co - ph * njt;
            const int ka = (ph >> 1) + 2 - 2 * khp, kb = (ph & 1) + 2 - 2 * kwp;
            if (ka <= 2 && kb <= 2) {
                const float* src = w + (((long long)ka * 3 + kb) * njt + c) * cin + ch;
                v = make_float4(src[0], src[1], src[2], src[3]);
            }
        }
        *reinterpret_cast<float4*>(packed + g * 4) = v;
    }
}

// head weights as the pointwise panel of the engine's run_head_pointwise: row ci, column (tap, phase, joint) -> [2048 / 4][coutP][4];
// W'[khp][kwp][ci][(a,b),c] = w[a+2-2khp][b+2-2kwp][c][ci] (w is [3,3,njt,Cin]); tracks max |panel| for the fp16 cells
__global__ void pack_head_pw_kernel(const float* __restrict__ w, int njt, int cin, int coutP, float* __restrict__ packed,
                                    float* __restrict__ rng) {
    float mx = 0.f;
    const long long total = (long long)(cin >> 2) * coutP;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(g % coutP);
        const int ch = (int)(g / coutP) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < 16 * njt) {
            const int tap = col / (4 * njt), r = col - tap * 4 * njt, ph = r / njt, c = r - ph * njt;
            const int khp = tap >> 1, kwp = tap & 1;
            const int ka = (ph >> 1) + 2 - 2 * khp, kb = (ph & 1) + 2 - 2 * kwp;
            if (ka <= 2 && kb <= 2) {
                const float* src = w + (((long long)ka * 3 + kb) * njt + c) * cin + ch;
                v = make_float4(src[0], src[1], src[2], src[3]);
            }
        }
        *reinterpret_cast<float4*>(packed + g * 4) = v;
        mx = amax4(mx, v);
    }
    pack_track(rng, mx);
}

__global__ void fold_bn_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps, int C,
                               float* __restrict__ scale, float* __restrict__ bias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float inv = gamma[c] / sqrtf(var[c] + eps);
        scale[c] = inv;
        bias[c] = beta[c] - mean[c] * inv;
    }
}

__global__ void head_bias_kernel(const float* __restrict__ b, int njt, float* __restrict__ bias4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * njt) bias4[i] = b[i % njt];
}

// data-gradient panels: Wd[tap'][co][ci] = W[taps-1-tap'][ci][co] * scale[co]; K' = taps*Cout (co fastest), N' = Cin
__global__ void pack_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ scale, int taps, int cin,
                                  int cout, int cinP, int nchunks, float* __restrict__ packed, float* __restrict__ rng) {
    float mx = 0.f;
    const long long total = (long long)nchunks * cinP;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(g % cinP);
        const int q = (int)(g / cinP);
        const int cout4 = cout >> 2;
        const int tapp = q / cout4, co = (q - tapp * cout4) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tapp < taps && ci < cin) {
            const int tap = taps - 1 - tapp;          // spatial flip (row-major taps: flipping both axes reverses the index)
            const float* src = w + ((long long)tap * cin + ci) * cout + co;
            v = make_float4(src[0] * scale[co], src[1] * scale[co + 1], src[2] * scale[co + 2], src[3] * scale[co + 3]);
        }
        *reinterpret_cast<float4*>(packed + g * 4) = v;
        mx = amax4(mx, v);
    }
    pack_track(rng, mx);
}

// All non-head layers in ONE launch (the per-layer kernels above cost ~180 dispatches of 4 us + their gaps per step):
// blockIdx.y = layer, blockIdx.z = job (0: forward panel, 1: data-gradient panel, 2: folded BN).  The data-gradient job folds
// scale = gamma / sqrt(var + eps) itself (the same expression as fold_bn_kernel, so the same bits) instead of waiting for job 2.
struct PackDesc {
    long long w_off, g_off, b_off, mean_off, var_off;
    int taps, cin_real, cin, cout, coutP, nchunks_f, cinP, nchunks_b;
    float *d_w, *rng_f, *d_wT, *rng_b, *d_scale, *d_bias;
};
__global__ __launch_bounds__(256) void pack_all_kernel(const PackDesc* __restrict__ table, const float* __restrict__ params,
                                                       const float* __restrict__ stats, float eps) {
    const PackDesc d = table[blockIdx.y];
    const float* w = params + d.w_off;
    if (blockIdx.z == 2) {
        for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d.cout; c += gridDim.x * blockDim.x) {
            const float inv = params[d.g_off + c] / sqrtf(stats[d.var_off + c] + eps);
            d.d_scale[c] = inv;
            d.d_bias[c] = params[d.b_off + c] - stats[d.mean_off + c] * inv;
        }
        return;
    }
    float mx = 0.f;
    if (blockIdx.z == 0) {
        const long long total = (long long)d.nchunks_f * d.coutP;
        if ((long long)blockIdx.x * blockDim.x >= total) return;
        const int cin4 = d.cin >> 2;
        for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
            const int co = (int)(g % d.coutP);
            const int q = (int)(g / d.coutP);
            const int tap = q / cin4, ch = (q - tap * cin4) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tap < d.taps && co < d.cout) {
                float* pv = &v.x;
                for (int e = 0; e < 4; ++e)
                    if (ch + e < d.cin_real) pv[e] = w[((long long)tap * d.cin_real + ch + e) * d.cout + co];
            }
            *reinterpret_cast<float4*>(d.d_w + g * 4) = v;
            mx = amax4(mx, v);
        }
        pack_track(d.rng_f, mx);
    } else {
        if (!d.d_wT) return;
        const long long total = (long long)d.nchunks_b * d.cinP;
        if ((long long)blockIdx.x * blockDim.x >= total) return;
        const int cout4 = d.cout >> 2;
        for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
            const int ci = (int)(g % d.cinP);
            const int q = (int)(g / d.cinP);
            const int tapp = q / cout4, co = (q - tapp * cout4) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tapp < d.taps && ci < d.cin) {
                const int tap = d.taps - 1 - tapp;
                const float* src = w + ((long long)tap * d.cin + ci) * d.cout + co;
                float sc[4];
                for (int e = 0; e < 4; ++e) sc[e] = params[d.g_off + co + e] / sqrtf(stats[d.var_off + co + e] + eps);
                v = make_float4(src[0] * sc[0], src[1] * sc[1], src[2] * sc[2], src[3] * sc[3]);
            }
            *reinterpret_cast<float4*>(d.d_wT + g * 4) = v;
            mx = amax4(mx, v);
        }
        pack_track(d.rng_b, mx);
    }
}

// head data-gradient panels: "input" channels = phase-major 4*njt padded to cpad, taps 2x2 flipped, N' = Cin
__global__ void pack_head_dgrad_kernel(const float* __restrict__ w, int njt, int cin, int cpad, int cinP, int nchunks,
                                       float* __restrict__ packed) {
    const long long total = (long long)nchunks * cinP;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(g % cinP);
        const int q = (int)(g / cinP);
        const int c4 = cpad >> 2;
        const int tapp = q / c4, cob = (q - tapp * c4) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tapp < 4 && ci < cin) {
            const int tap = 3 - tapp, khp = tap >> 1, kwp = tap & 1;
            float* pv = &v.x;
            for (int e = 0; e < 4; ++e) {
                const int co = cob + e;
                if (co >= 4 * njt) continue;
                const int ph = co / njt, c = co - ph * njt;
                const int ka = (ph >> 1) + 2 - 2 * khp, kb = (ph & 1) + 2 - 2 * kwp;
                if (ka <= 2 && kb <= 2) pv[e] = w[(((long long)ka * 3 + kb) * njt + c) * cin + ci];
            }
        }
        *reinterpret_cast<float4*>(packed + g * 4) = v;
    }
}

// 16-bit tier: BOTH heads' data-gradient panels side by side -- "input" channels [0, cpad0) the part head's phases, [cpad0, cpad0 +
// cpad1) the locref head's, zero up to CT -- so that one launch over head_gather_h1_kernel's tensor gives d features of both heads
__global__ void pack_heads_dgrad_kernel(const float* __restrict__ w0, int njt0, int cpad0, const float* __restrict__ w1, int njt1, int cpad1,
                                        int cin, int CT, int cinP, int nchunks, float* __restrict__ packed, float* __restrict__ rng) {
    const long long total = (long long)nchunks * cinP;
    float mx = 0.f;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(g % cinP);
        const int q = (int)(g / cinP);
        const int c4 = CT >> 2;
        const int tapp = q / c4, cob = (q - tapp * c4) << 2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tapp < 4 && ci < cin) {
            const int tap = 3 - tapp, khp = tap >> 1, kwp = tap & 1;
            float* pv = &v.x;
            for (int e = 0; e < 4; ++e) {
                int co = cob + e;
                const float* w = w0;
                int njt = njt0;
                if (co >= cpad0) { co -= cpad0; w = w1; njt = njt1; if (co >= cpad1) continue; }
                if (co >= 4 * njt) continue;
                const int ph = co / njt, c = co - ph * njt;
                const int ka = (ph >> 1) + 2 - 2 * khp, kb = (ph & 1) + 2 - 2 * kwp;
                if (ka <= 2 && kb <= 2) pv[e] = w[(((long long)ka * 3 + kb) * njt + c) * cin + ci];
            }
        }
        *reinterpret_cast<float4*>(packed + g * 4) = v;
        mx = amax4(mx, v);
    }
    pack_track(rng, mx);
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dWraw[k][co] += sum_m A[m][k] * dY[m][co]   (k = (tap, ci), HWIO order)
// fp32 MFMA with the output tile's rows = k, cols = co and the reduction over pixels m.  LDS images are the
// natural [pixel][k] / [pixel][co] row-major tiles (no transposes): an MFMA operand register is one
// ds_read_b32 of 32 consecutive floats.  The pixel range is split across workgroups (grid.z), partial sums
// are combined with float atomics (256-B wave shapes).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;       // NHWC [N,H,W,Cin]  (forward input of the conv)
    const float* dy;      // [M, Cdy] masked upstream gradient (M = N*Ho*Wo)
    float* dw;            // [Kchunks*4][Cdy] accumulated
    float* colsum;        // [Cdy] accumulated column sums of dY (d beta / d bias), or nullptr
    int N, H, W, Cin, log2cin4, Ho, Wo, Cdy;
    int KW, stride, dil, pad_t, pad_l, ntaps, kchunks;
    int M, m_per_block;
    unsigned x_bytes, dy_bytes;
    // wgrad_dma only: the operands' fp16 high / low copies (ConvArgs::shadow, same geometry and byte extents as x / dy) and the range
    // slots that decide whether they may be read: previous step (the scale they were written with) and this step (what they hold)
    const void* xs; const void* dys;
    const float *x_prev, *x_cur, *dy_prev, *dy_cur;
    int* fail_flag;       // x is an H2-only tensor (no fp32 twin, x == nullptr): unusable copies raise this flag instead of falling back
};

template <int T>      // tile = 64T (k) x 64T (co); 4 waves as 2 x 2, wave tile 32T x 32T
__device__ __forceinline__ void wgrad_f32_body(const WgradArgs& p) {
    constexpr int BR = 64 * T;               // tile extent (both dims)
    constexpr int CH = BR / 4;               // 16-byte chunks per tile row
    constexpr int NLD = 32 * CH / 256;       // staged chunks per thread per operand (T=1: 2, T=2: 4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sA = reinterpret_cast<float*>(smem);          // [2][32][BR]
    float* sB = sA + 2 * 32 * BR;                        // [2][32][BR]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int q0 = blockIdx.x * CH;          // first k-chunk of this tile
    const int n0 = blockIdx.y * BR;          // first output column
    const int m_lo = blockIdx.z * p.m_per_block;
    const int m_hi = min(p.M, m_lo + p.m_per_block);
    const int nsteps = (m_hi - m_lo + 31) / 32;
    if (nsteps <= 0) return;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

    // this thread stages chunk `cc` of pixel rows pr + (256/CH)*i
    const int cc = t % CH, pr = t / CH;
    const int q = q0 + cc;
    const int cin4m1 = (p.Cin >> 2) - 1;
    const int tap = q >> p.log2cin4, ch = (q & cin4m1) << 2;
    const bool qok = q < p.kchunks && tap < p.ntaps;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int dh = kh * p.dil - p.pad_t, dw = kw * p.dil - p.pad_l;
    const int cob = n0 + 4 * cc;
    const bool cok = cob < p.Cdy;
    const int HoWo = p.Ho * p.Wo;

    float4 ra[NLD], rb[NLD];
    auto gload = [&](int step) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = m_lo + step * 32 + pr + (256 / CH) * i;
            unsigned offa = OOBT, offb = OOBT;
            if (m < m_hi) {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const int hi = ho * p.stride + dh, wi = wo * p.stride + dw;
                if (qok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    offa = (unsigned)(((n * p.H + hi) * p.W + wi) * p.Cin + ch) << 2;
                if (cok) offb = (unsigned)(m * p.Cdy + cob) << 2;
            }
            ra[i] = bload16(rs_x, offa);
            rb[i] = bload16(rs_dy, offb);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = pr + (256 / CH) * i;
            *reinterpret_cast<float4*>(sA + (buf * 32 + row) * BR + 4 * cc) = ra[i];
            *reinterpret_cast<float4*>(sB + (buf * 32 + row) * BR + 4 * cc) = rb[i];
        }
    };
    floatx16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();
    const bool do_colsum = p.colsum != nullptr && blockIdx.x == 0 && t < BR;     // one k-tile sums dY's columns
    float csum = 0.f;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        if (do_colsum) {
            const float* col = sB + buf * 32 * BR + t;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) csum += col[r * BR];
        }
        const float* a_base = sA + buf * 32 * BR + wm * 32 * T + l31;
        const float* b_base = sB + buf * 32 * BR + wn * 32 * T + l31;
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const int row = 2 * pp + half;
            float af[T], bf[T];
#pragma unroll
            for (int i = 0; i < T; ++i) af[i] = a_base[row * BR + 32 * i];
#pragma unroll
            for (int j = 0; j < T; ++j) bf[j] = b_base[row * BR + 32 * j];
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    if (do_colsum && n0 + t < p.Cdy) atomicAdd(p.colsum + n0 + t, csum);
    // C/D layout: col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*half (k)
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int co = n0 + wn * 32 * T + 32 * j + l31;
        if (co >= p.Cdy) continue;
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = q0 * 4 + wm * 32 * T + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (k < p.kchunks * 4) {
#if defined(DGP_WX) && DGP_WX == 6
                    p.dw[(long long)k * p.Cdy + co] = acc[i][j][r];
#else
                    atomicAdd(p.dw + (long long)k * p.Cdy + co, acc[i][j][r]);
#endif
                }
            }
    }
}

template <int T>
__global__ __launch_bounds__(256) void wgrad_f32(const WgradArgs p) { wgrad_f32_body<T>(p); }

// ------------------------------------------------------------------------------------------------
// Weight gradient on the 16-bit matrix pipe (fp16 high/low split, 3 MFMAs per fp32-class product; the split and the range
// scaling are those of conv_igemm_split_ls, EXPERIMENTS.md section 2).  Same tiling as wgrad_f32<2>: 128 (k) x 128 (co) tile,
// reduction over 32-pixel steps, pixel range split over grid.z, fp32 atomics.  Both MFMA operands want 8 consecutive
// PIXELS of one channel -- a column of the natural [pixel][channel] tile -- so the LDS images stay row-major (fp16 planes,
// 256-byte pixel rows, 16-byte chunks XOR-swizzled) and the operands come in through ds_read_b64_tr_b16, the hardware
// transposed read (scripts/micro/tr_b16.hip checks its lane map): lane 4q+p of a 16-lane group supplies row q / columns
// 4p..4p+3 of a 4 x 16 block and lane i receives column i.
// ------------------------------------------------------------------------------------------------
typedef _Float16 half8t __attribute__((ext_vector_type(8)));
typedef _Float16 half2t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float rng_scale(const float* slots, int lane) {     // 2^(14 - E) for max = m 2^E; 1 for zero / untracked
    const float4 v = reinterpret_cast<const float4*>(slots)[lane];
    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int be = (int)((__float_as_uint(m) >> 23) & 0xFF);
    if (be == 0 || be == 0xFF) return 1.f;
    int se = 127 + 14 - (be - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}

__device__ __forceinline__ void split_hl(const float4 v, const float s, uint2& ph, uint2& pl) {
    // same 10-VALU sequence as split2_f16 (dgp_kernels.hip): h = f16(s x) written into its half of the packed register,
    // r = s x - h exact in fp32 with h read as an fp16 operand, l = f16(r) packed
    unsigned h01, h23;
    float r0, r1, r2, r3;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(s), "v"(v.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(s), "v"(v.y));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(s), "v"(v.z));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(s), "v"(v.w));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(s), "v"(v.x), "v"(h01));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(s), "v"(v.y), "v"(h01));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r2) : "v"(s), "v"(v.z), "v"(h23));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r3) : "v"(s), "v"(v.w), "v"(h23));
    const half2t l01 = {(_Float16)r0, (_Float16)r1}, l23 = {(_Float16)r2, (_Float16)r3};
    ph.x = h01; ph.y = h23;
    pl.x = __builtin_bit_cast(unsigned, l01); pl.y = __builtin_bit_cast(unsigned, l23);
}

// byte offset of 16-byte chunk ch (0..15) of pixel row `row` in a [32][128 halves] plane (conflict-free for the 8-byte
// row-wise writes and the transposed reads: cdna_hip_programming.md T10, image (b))
__device__ __forceinline__ unsigned swz_off(int row, int ch) { return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

struct WgradRanges { const float* x_absmax; const float* dy_absmax; };

#ifdef DGP_DIAG
// diagnostic build only: s_memtime stamps fence the schedule; read SHARES, not totals (scripts/diag_wgrad.py)
#define WG_STAMP(x)                                                                     \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
__device__ unsigned long long g_wgrad_diag[16];
#else
#define WG_STAMP(x)
#endif

__global__ __launch_bounds__(256) void wgrad_h3(const WgradArgs p, const WgradRanges rg) {
    constexpr int BR = 128, CH = 32, NLD = 4;
    constexpr unsigned PLANE = 32 * 256;          // bytes of one fp16 plane of a 32-pixel tile
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 buffers][A hi, A lo, B hi, B lo][PLANE]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int q0 = blockIdx.x * CH;
    const int n0 = blockIdx.y * BR;
    const int m_lo = blockIdx.z * p.m_per_block;
    const int m_hi = min(p.M, m_lo + p.m_per_block);
    const int nsteps = (m_hi - m_lo + 31) / 32;
    if (nsteps <= 0) return;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
    const float sx = rng_scale(rg.x_absmax, lane), sy = rng_scale(rg.dy_absmax, lane);
    const float post = 1.f / (sx * sy);           // exact power of two

    const int cc = t % CH, pr = t / CH;           // this thread stages 4-channel chunk cc of pixel rows pr + 8 i
    const int q = q0 + cc;
    const int cin4m1 = (p.Cin >> 2) - 1;
    const int tap = q >> p.log2cin4, ch = (q & cin4m1) << 2;
    const bool qok = q < p.kchunks && tap < p.ntaps;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int dh = kh * p.dil - p.pad_t, dw = kw * p.dil - p.pad_l;
    const int cob = n0 + 4 * cc;
    const bool cok = cob < p.Cdy;
    const int HoWo = p.Ho * p.Wo;

    float4 ra[NLD], rb[NLD];
    // pixel coordinates of this thread's NLD rows, advanced by 32 pixels per step (no integer divisions in the loop: with three
    // 16-bit MFMAs per product the staging arithmetic, not the matrix pipe, sets the pace of this kernel)
    const bool pointwise = p.ntaps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.H == p.Ho && p.W == p.Wo;
    int pn[NLD], pho[NLD], pwo[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int m = m_lo + pr + 8 * i;
        pn[i] = m / HoWo;
        const int rem = m - pn[i] * HoWo;
        pho[i] = rem / p.Wo;
        pwo[i] = rem - pho[i] * p.Wo;
    }
    auto gload = [&](int step) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = m_lo + step * 32 + pr + 8 * i;
            unsigned offa = OOBT, offb = OOBT;
            if (m < m_hi) {
                if (pointwise) {
                    if (qok) offa = (unsigned)(m * p.Cin + ch) << 2;
                } else {
                    const int hi = pho[i] * p.stride + dh, wi = pwo[i] * p.stride + dw;
                    if (qok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        offa = (unsigned)(((pn[i] * p.H + hi) * p.W + wi) * p.Cin + ch) << 2;
                }
                if (cok) offb = (unsigned)(m * p.Cdy + cob) << 2;
            }
            ra[i] = bload16(rs_x, offa);
            rb[i] = bload16(rs_dy, offb);
            if (!pointwise) {                      // next step: 32 pixels further
                pwo[i] += 32;
                while (pwo[i] >= p.Wo) { pwo[i] -= p.Wo; if (++pho[i] == p.Ho) { pho[i] = 0; ++pn[i]; } }
            }
        }
    };
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);          // column sums of dY (d beta / d bias) from the staging registers
    const bool do_colsum = p.colsum != nullptr && blockIdx.x == 0;
    auto lstore = [&](int buf) {
        char* base = smem + buf * (4 * PLANE);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = pr + 8 * i;
            const unsigned o = swz_off(row, cc >> 1) + 8u * (cc & 1);
            uint2 h, l;
            split_hl(ra[i], sx, h, l);
            *reinterpret_cast<uint2*>(base + o) = h;
            *reinterpret_cast<uint2*>(base + PLANE + o) = l;
            split_hl(rb[i], sy, h, l);
            *reinterpret_cast<uint2*>(base + 2 * PLANE + o) = h;
            *reinterpret_cast<uint2*>(base + 3 * PLANE + o) = l;
            if (do_colsum) { cs4.x += rb[i].x; cs4.y += rb[i].y; cs4.z += rb[i].z; cs4.w += rb[i].w; }
        }
    };
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read addressing: 16-lane group g: columns 16 (g & 1) .. +15 of the 32-wide block, pixel group g >> 1
    const int g16 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned smem_base = (unsigned)(size_t)smem;
    auto tr_addr = [&](int buf, int plane, int col0 /* first channel of the 32-wide block */, int pix0) {
        const int row = pix0 + 8 * (g16 >> 1) + qq;
        const int col = col0 + 16 * (g16 & 1) + 4 * pp;             // in halves
        return smem_base + (unsigned)(buf * 4 * PLANE + plane * PLANE) + swz_off(row, col >> 3) + 8u * ((col >> 2) & 1);
    };

#ifdef DGP_DIAG
    unsigned long long T0 = 0, T1 = 0, T2[2] = {0, 0}, T3[2] = {0, 0}, T4 = 0, T5 = 0, TB = 0, TE = 0, dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    WG_STAMP(TB);
#endif
    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        WG_STAMP(T0);
        if (s + 1 < nsteps) gload(s + 1);
        WG_STAMP(T1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // operands: [operand A/B][block][piece hi/lo] = 8 halves = two transposed reads of 4 pixels each
            unsigned long long lo4[2][2][2], hi4[2][2][2];
#pragma unroll
            for (int op = 0; op < 2; ++op)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const int col0 = (op == 0 ? wm : wn) * 64 + 32 * b;
                        const unsigned a0 = tr_addr(buf, 2 * op + pc, col0, 16 * kk);          // pixels 0-3 of this lane's k-group
                        const unsigned a1 = tr_addr(buf, 2 * op + pc, col0, 16 * kk + 4);      // pixels 4-7 (the row enters the swizzle)
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo4[op][b][pc]) : "v"(a0) : "memory");
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi4[op][b][pc]) : "v"(a1) : "memory");
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int op = 0; op < 2; ++op)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) asm volatile("" : "+v"(lo4[op][b][pc]), "+v"(hi4[op][b][pc]));
            WG_STAMP(T2[kk]);
            auto frag = [&](int op, int b, int pc) {
                const uint4 u = make_uint4((unsigned)lo4[op][b][pc], (unsigned)(lo4[op][b][pc] >> 32), (unsigned)hi4[op][b][pc],
                                           (unsigned)(hi4[op][b][pc] >> 32));
                return __builtin_bit_cast(half8t, u);
            };
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    floatx16 a = acc[i][j];
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(0, i, 1), frag(1, j, 0), a, 0, 0, 0);      // lo * hi
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(0, i, 0), frag(1, j, 1), a, 0, 0, 0);      // hi * lo
                    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(0, i, 0), frag(1, j, 0), a, 0, 0, 0);      // hi * hi
                    acc[i][j] = a;
                }
            WG_STAMP(T3[kk]);
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        WG_STAMP(T4);
        __syncthreads();
#ifdef DGP_DIAG
        WG_STAMP(T5);
        dsum[0] += T1 - T0; dsum[1] += T2[0] - T1; dsum[2] += T3[0] - T2[0]; dsum[3] += T2[1] - T3[0];
        dsum[4] += T3[1] - T2[1]; dsum[5] += T4 - T3[1]; dsum[6] += T5 - T4;
#endif
    }
#ifdef DGP_DIAG
    WG_STAMP(TE);
#endif
    if (do_colsum && cok) {
        atomicAdd(p.colsum + cob, cs4.x); atomicAdd(p.colsum + cob + 1, cs4.y);
        atomicAdd(p.colsum + cob + 2, cs4.z); atomicAdd(p.colsum + cob + 3, cs4.w);
    }
    // C/D layout: col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*half (k)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = n0 + wn * 64 + 32 * j + l31;
        if (co >= p.Cdy) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = q0 * 4 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (k < p.kchunks * 4) {
#if defined(DGP_WX) && DGP_WX == 6
                    p.dw[(long long)k * p.Cdy + co] = acc[i][j][r] * post;
#else
                    atomicAdd(p.dw + (long long)k * p.Cdy + co, acc[i][j][r] * post);
#endif
                }
            }
    }
#ifdef DGP_DIAG
    if (lane == 0 && wave == 0) {
        unsigned long long TF;
        WG_STAMP(TF);
        for (int k = 0; k < 7; ++k) atomicAdd(&g_wgrad_diag[k], dsum[k]);
        atomicAdd(&g_wgrad_diag[7], T0 ? (unsigned long long)nsteps : 0ull);
        atomicAdd(&g_wgrad_diag[8], TE - TB);        // prologue + loop
        atomicAdd(&g_wgrad_diag[9], TF - TE);        // atomics epilogue (issue + colsum)
        atomicAdd(&g_wgrad_diag[10], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// wgrad_h3p: wgrad_h3 as an explicit software pipeline.  scripts/diag_wgrad.py stamps on wgrad_h3 (b4.conv2, cycles per 32-pixel
// step of 4400): staging-load issue 1630, transposed reads 690, the 24 MFMAs 850, split + LDS stores 1100, barrier 130 -- the
// phases of a wave ran one after the other and the four waves of a workgroup ran them in lock-step, so each shared unit (texture
// addresser, LDS, matrix pipe) was busy in bursts and idle between them (PMC: TA busy 22 %, L2 hit 83 %, L1->L2 latency 173 cycles,
// MFMA busy 23 %).  Here every MFMA is followed by a fixed slice of the other work:
//   first half of a step  : 12 MFMAs on pixel group 0 | transposed reads of group 1 | split + store of the rows loaded one step ago
//   barrier
//   second half           : 12 MFMAs on pixel group 1 | transposed reads of the NEXT step's group 0 | global loads for step + 2
// MFMAs go round the four accumulator blocks (dependent MFMAs are four issues apart).  LDS addresses are 10 lane registers plus
// instruction offsets (the swizzle XOR splits into a lane part and the constants 4 b and h), staging offsets advance by adds.
// ------------------------------------------------------------------------------------------------
template <unsigned OFF>
__device__ __forceinline__ unsigned long long trr(unsigned base) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory");
    return v;
}

struct WgFrag { unsigned long long v[2][2]; };          // [piece hi/lo][pixels 0-3 / 4-7] of one 32-channel block, one k-group

__device__ __forceinline__ half8t wg_op(const WgFrag& f, int pc) {
    const uint4 u = make_uint4((unsigned)f.v[pc][0], (unsigned)(f.v[pc][0] >> 32), (unsigned)f.v[pc][1], (unsigned)(f.v[pc][1] >> 32));
    return __builtin_bit_cast(half8t, u);
}

__global__ __launch_bounds__(256) void wgrad_h3p(const WgradArgs p, const WgradRanges rg) {
    constexpr int CH = 32;
    constexpr unsigned PLANE = 8192u, BUF = 32768u;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 buffers][A hi, A lo, B hi, B lo][32 pixels x 256 B]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int q0 = blockIdx.x * CH;
    const int n0 = blockIdx.y * 128;
    const int m_lo = blockIdx.z * p.m_per_block;
    const int m_hi = min(p.M, m_lo + p.m_per_block);
    const int nsteps = (m_hi - m_lo + 31) / 32;
    if (nsteps <= 0) return;
    const bool pointwise = p.ntaps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.H == p.Ho && p.W == p.Wo;
    // rows at and beyond m_hi read as zeros through the descriptors' range check where the offset grows with the pixel index
    const unsigned x_lim = pointwise ? (unsigned)min((long long)p.x_bytes, (long long)m_hi * p.Cin * 4) : p.x_bytes;
    const unsigned dy_lim = (unsigned)min((long long)p.dy_bytes, (long long)m_hi * p.Cdy * 4);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)x_lim, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)dy_lim, 0x00020000);
    const float sx = rng_scale(rg.x_absmax, lane), sy = rng_scale(rg.dy_absmax, lane);
    const float post = 1.f / (sx * sy);

    const int cc = t % CH, pr = t / CH;           // this thread stages 4-channel chunk cc of pixel rows pr + 8 i
    const int q = q0 + cc;
    const int cin4m1 = (p.Cin >> 2) - 1;
    const int tap = q >> p.log2cin4, ch = (q & cin4m1) << 2;
    const bool qok = q < p.kchunks && tap < p.ntaps;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int dh = kh * p.dil - p.pad_t, dw = kw * p.dil - p.pad_l;
    const int cob = n0 + 4 * cc;
    const bool cok = cob < p.Cdy;
    const int HoWo = p.Ho * p.Wo;

    // staging walkers: byte offsets of this thread's four pixel rows, advanced by 32 pixels per load
    unsigned offb[4], offa[4];
    int whi[4], wwi[4], mleft[4];
    const unsigned stepb = cok ? 32u * (unsigned)p.Cdy * 4u : 0u;
    const unsigned stepa_pw = qok ? 32u * (unsigned)p.Cin * 4u : 0u;
    const int hlim = p.Ho * p.stride + dh, wlim = p.Wo * p.stride + dw;
    const unsigned d_px = (unsigned)(32 * p.stride * p.Cin * 4);
    const unsigned d_row = (unsigned)((p.stride * p.W - p.Wo * p.stride) * p.Cin * 4);
    const unsigned d_img = (unsigned)((p.H * p.W - p.Ho * p.stride * p.W) * p.Cin * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m_lo + pr + 8 * i;
        offb[i] = cok ? (unsigned)(m * p.Cdy + cob) << 2 : OOBT;
        if (pointwise) {
            offa[i] = qok ? (unsigned)(m * p.Cin + ch) << 2 : OOBT;
            whi[i] = wwi[i] = mleft[i] = 0;
        } else {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            whi[i] = ho * p.stride + dh;
            wwi[i] = wo * p.stride + dw;
            offa[i] = (unsigned)(((n * p.H + whi[i]) * p.W + wwi[i]) * p.Cin + ch) << 2;
            mleft[i] = m_hi - m;
        }
    }
    float4 ra[4], rb[4];
    auto load_a = [&](int i) {
        if (pointwise) {
            ra[i] = bload16(rs_x, offa[i]);
            offa[i] += stepa_pw;
        } else {
            const bool ok = qok && mleft[i] > 0 && (unsigned)whi[i] < (unsigned)p.H && (unsigned)wwi[i] < (unsigned)p.W;
            ra[i] = bload16(rs_x, ok ? offa[i] : OOBT);
            mleft[i] -= 32;
            wwi[i] += 32 * p.stride;
            offa[i] += d_px;
            while (wwi[i] >= wlim) {
                wwi[i] -= p.Wo * p.stride;
                whi[i] += p.stride;
                offa[i] += d_row;
                if (whi[i] >= hlim) { whi[i] -= p.Ho * p.stride; offa[i] += d_img; }
            }
        }
    };
    auto load_b = [&](int i) {
        rb[i] = bload16(rs_dy, offb[i]);
        offb[i] += stepb;
    };
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_colsum = p.colsum != nullptr && blockIdx.x == 0;

    // LDS store bases (even / odd i): row pr + 8 i, 16-byte chunk cc >> 1 swizzled, 8-byte half cc & 1
    const int lw = (cc >> 1) ^ (((pr & 3) << 2) | (pr >> 2));
    char* wbase[2];
    wbase[0] = smem + 256 * pr + 16 * lw + 8 * (cc & 1);
    wbase[1] = smem + 256 * pr + 16 * (lw ^ 2) + 8 * (cc & 1);
    // transposed-read bases [operand][block b][pixel half h]
    const int g16 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int f_lane = (qq << 2) | (2 * (g16 >> 1));
    const unsigned rrow = (unsigned)(size_t)smem + 256u * (8 * (g16 >> 1) + qq) + 8u * (pp & 1);
    unsigned rbase[2][2][2];
#pragma unroll
    for (int op = 0; op < 2; ++op) {
        const int L = (8 * (op == 0 ? wm : wn) + 2 * (g16 & 1) + (pp >> 1)) ^ f_lane;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) rbase[op][b][h] = rrow + 16u * (unsigned)(L ^ (4 * b) ^ h);
    }

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    WgFrag F0[2][2], F1[2][2];          // [operand][block]

#define WG_FENCE() __builtin_amdgcn_sched_barrier(0)
    // one split + store unit: u < 4 -> x row u (planes 0, 1), else dY row u - 4 (planes 2, 3)
#define WG_UNIT(SB, U)                                                                                               \
    do {                                                                                                             \
        constexpr int i_ = (U) & 3;                                                                                  \
        char* d_ = wbase[i_ & 1] + (SB) * BUF + 2048 * i_ + ((U) < 4 ? 0 : 2 * PLANE);                               \
        uint2 h_, l_;                                                                                                \
        if ((U) < 4) split_hl(ra[i_], sx, h_, l_);                                                                   \
        else {                                                                                                       \
            split_hl(rb[i_], sy, h_, l_);                                                                            \
            if (do_colsum) { cs4.x += rb[i_].x; cs4.y += rb[i_].y; cs4.z += rb[i_].z; cs4.w += rb[i_].w; }           \
        }                                                                                                            \
        *reinterpret_cast<uint2*>(d_) = h_;                                                                          \
        *reinterpret_cast<uint2*>(d_ + PLANE) = l_;                                                                  \
    } while (0)
#define WG_LOAD(U) do { if ((U) < 4) load_a((U) & 3); else load_b((U) & 3); } while (0)
    // read slice R (0..7) of a fragment set: two of its sixteen transposed reads
#define WG_READ2(F, BUFI, KK, R)                                                                                     \
    do {                                                                                                             \
        constexpr int op_ = ((R) >> 2) & 1, b_ = ((R) >> 1) & 1, pc_ = (R) & 1;                                      \
        constexpr unsigned o_ = (BUFI) * BUF + 4096u * (KK) + (2 * op_ + pc_) * PLANE;                               \
        F[op_][b_].v[pc_][0] = trr<o_>(rbase[op_][b_][0]);                                                           \
        F[op_][b_].v[pc_][1] = trr<o_ + 1024u>(rbase[op_][b_][1]);                                                   \
    } while (0)
    // MFMA number G (0..11) of a half step: term G / 4 (lo*hi, hi*lo, hi*hi), block G % 4
#define WG_MMA(F, G)                                                                                                 \
    do {                                                                                                             \
        constexpr int blk_ = (G) & 3, i_ = blk_ >> 1, j_ = blk_ & 1, term_ = (G) >> 2;                               \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wg_op(F[0][i_], term_ == 0 ? 1 : 0),                    \
                                                             wg_op(F[1][j_], term_ == 1 ? 1 : 0), acc[i_][j_], 0, 0, 0); \
    } while (0)
#define WG_WAIT_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    // prologue: step 0 -> buffer 0, step 1 -> registers, fragments of step 0 / pixel group 0
#pragma unroll
    for (int u = 0; u < 8; ++u) WG_LOAD(u);
    WG_UNIT(0, 0); WG_UNIT(0, 1); WG_UNIT(0, 2); WG_UNIT(0, 3); WG_UNIT(0, 4); WG_UNIT(0, 5); WG_UNIT(0, 6); WG_UNIT(0, 7);
#pragma unroll
    for (int u = 0; u < 8; ++u) WG_LOAD(u);
    __syncthreads();
    WG_READ2(F0, 0, 0, 0); WG_READ2(F0, 0, 0, 1); WG_READ2(F0, 0, 0, 2); WG_READ2(F0, 0, 0, 3);
    WG_READ2(F0, 0, 0, 4); WG_READ2(F0, 0, 0, 5); WG_READ2(F0, 0, 0, 6); WG_READ2(F0, 0, 0, 7);
    WG_WAIT_LDS();
    WG_FENCE();

#if defined(DGP_WX)      // timing-only ablations (results are garbage): scripts/ablate_wgrad.sh
#if DGP_WX == 1         // no global loads inside the loop
#undef WG_LOAD
#define WG_LOAD(U) do { if ((U) < 4) asm volatile("" : "+v"(ra[(U) & 3].x), "+v"(ra[(U) & 3].y), "+v"(ra[(U) & 3].z), "+v"(ra[(U) & 3].w)); else asm volatile("" : "+v"(rb[(U) & 3].x), "+v"(rb[(U) & 3].y), "+v"(rb[(U) & 3].z), "+v"(rb[(U) & 3].w)); } while (0)
#elif DGP_WX == 2       // no MFMAs
#undef WG_MMA
#define WG_MMA(F, G) do { asm volatile("" :: "v"(F[0][0].v[0][0]), "v"(F[0][1].v[0][0]), "v"(F[1][0].v[0][0]), "v"(F[1][1].v[0][0]), "v"(F[0][0].v[1][1]), "v"(F[0][1].v[1][1]), "v"(F[1][0].v[1][1]), "v"(F[1][1].v[1][1])); } while (0)
#elif DGP_WX == 3       // no transposed reads inside the loop
#undef WG_READ2
#define WG_READ2(F, BUFI, KK, R) do { constexpr int op_ = ((R) >> 2) & 1, b_ = ((R) >> 1) & 1, pc_ = (R) & 1; asm volatile("" : "+v"(F[op_][b_].v[pc_][0]), "+v"(F[op_][b_].v[pc_][1])); } while (0)
#elif DGP_WX == 4       // no split + store inside the loop
#undef WG_UNIT
#define WG_UNIT(SB, U) do { asm volatile("" :: "v"(ra[(U) & 3].x), "v"(rb[(U) & 3].x)); } while (0)
#elif DGP_WX == 5       // no mid-step barrier
#define __syncthreads() do { } while (0)
#endif
#endif
    // one step on compute buffer CB: stores go to the other buffer, the next step's first fragments come from it
#define WG_STEP(CB)                                                                                                  \
    do {                                                                                                             \
        WG_MMA(F0, 0);  WG_FENCE(); WG_READ2(F1, CB, 1, 0); WG_FENCE();                                              \
        WG_MMA(F0, 1);  WG_FENCE(); WG_READ2(F1, CB, 1, 1); WG_FENCE();                                              \
        WG_MMA(F0, 2);  WG_FENCE(); WG_READ2(F1, CB, 1, 2); WG_FENCE();                                              \
        WG_MMA(F0, 3);  WG_FENCE(); WG_READ2(F1, CB, 1, 3); WG_FENCE();                                              \
        WG_MMA(F0, 4);  WG_FENCE(); WG_READ2(F1, CB, 1, 4); WG_UNIT(1 - (CB), 0); WG_FENCE();                        \
        WG_MMA(F0, 5);  WG_FENCE(); WG_READ2(F1, CB, 1, 5); WG_UNIT(1 - (CB), 1); WG_FENCE();                        \
        WG_MMA(F0, 6);  WG_FENCE(); WG_READ2(F1, CB, 1, 6); WG_UNIT(1 - (CB), 2); WG_FENCE();                        \
        WG_MMA(F0, 7);  WG_FENCE(); WG_READ2(F1, CB, 1, 7); WG_UNIT(1 - (CB), 3); WG_FENCE();                        \
        WG_MMA(F0, 8);  WG_FENCE(); WG_UNIT(1 - (CB), 4); WG_FENCE();                                                \
        WG_MMA(F0, 9);  WG_FENCE(); WG_UNIT(1 - (CB), 5); WG_FENCE();                                                \
        WG_MMA(F0, 10); WG_FENCE(); WG_UNIT(1 - (CB), 6); WG_FENCE();                                                \
        WG_MMA(F0, 11); WG_FENCE(); WG_UNIT(1 - (CB), 7); WG_FENCE();                                                \
        WG_WAIT_LDS();                                                                                               \
        __syncthreads();                                                                                             \
        WG_FENCE();                                                                                                  \
        WG_MMA(F1, 0);  WG_FENCE(); WG_READ2(F0, 1 - (CB), 0, 0); WG_FENCE();                                        \
        WG_MMA(F1, 1);  WG_FENCE(); WG_READ2(F0, 1 - (CB), 0, 1); WG_FENCE();                                        \
        WG_MMA(F1, 2);  WG_FENCE(); WG_READ2(F0, 1 - (CB), 0, 2); WG_FENCE();                                        \
        WG_MMA(F1, 3);  WG_FENCE(); WG_READ2(F0, 1 - (CB), 0, 3); WG_FENCE();                                        \
        WG_MMA(F1, 4);  WG_FENCE(); WG_READ2(F0, 1 - (CB), 0, 4); WG_LOAD(0); WG_FENCE();                            \
        WG_MMA(F1, 5);  WG_FENCE(); WG_READ2(F0, 1 - (CB), 0, 5); WG_LOAD(4); WG_FENCE();                            \
        WG_MMA(F1, 6);  WG_FENCE(); WG_READ2(F0, 1 - (CB), 0, 6); WG_LOAD(1); WG_FENCE();                            \
        WG_MMA(F1, 7);  WG_FENCE(); WG_READ2(F0, 1 - (CB), 0, 7); WG_LOAD(5); WG_FENCE();                            \
        WG_MMA(F1, 8);  WG_FENCE(); WG_LOAD(2); WG_FENCE();                                                          \
        WG_MMA(F1, 9);  WG_FENCE(); WG_LOAD(6); WG_FENCE();                                                          \
        WG_MMA(F1, 10); WG_FENCE(); WG_LOAD(3); WG_FENCE();                                                          \
        WG_MMA(F1, 11); WG_FENCE(); WG_LOAD(7); WG_FENCE();                                                          \
        WG_WAIT_LDS();                                                                                               \
        WG_FENCE();                                                                                                  \
    } while (0)

    for (int s = 0; s < nsteps; s += 2) {
        WG_STEP(0);
        if (s + 1 < nsteps) WG_STEP(1);
    }
#undef WG_STEP
#undef WG_MMA
#undef WG_READ2
#undef WG_LOAD
#undef WG_UNIT
#undef WG_FENCE
#undef WG_WAIT_LDS

    if (do_colsum && cok) {
        atomicAdd(p.colsum + cob, cs4.x); atomicAdd(p.colsum + cob + 1, cs4.y);
        atomicAdd(p.colsum + cob + 2, cs4.z); atomicAdd(p.colsum + cob + 3, cs4.w);
    }
    // C/D layout: col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*half (k)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = n0 + wn * 64 + 32 * j + l31;
        if (co >= p.Cdy) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = q0 * 4 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (k < p.kchunks * 4) {
#if defined(DGP_WX) && DGP_WX == 6
                    p.dw[(long long)k * p.Cdy + co] = acc[i][j][r] * post;
#else
                    atomicAdd(p.dw + (long long)k * p.Cdy + co, acc[i][j][r] * post);
#endif
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad_dma: the 128 x 128 weight-gradient tile with BOTH operands arriving as fp16 high / low cells by LDS-DMA.
// wgrad_h3p's per-step demand (EXPERIMENTS.md section 6a (4)) was four units at 30-45 % each: 8 staging loads + their address walk, the
// fp16 split of 32 values, 16 ds_write_b64, 32 transposed reads and 24 MFMAs per thread.  Here the producers of x and dY have
// already written the split values (ConvArgs::shadow: the H2 cell layout, [pixel][8 channels: 16 B high | 16 B low]), so a step is
// 4 DMA instructions (no VGPR round trip, no VALU, no ds_write), the same transposed reads and the same MFMAs.
//   LDS        4 stages of 16 pixels x [A | B] x 512 B (per pixel row: 16 high chunks, then 16 low chunks; chunk slot XOR-swizzled by
//              f(row) on the SOURCE side -- the DMA destination is lane-linear -- exactly wgrad_h3p's conflict-free image at twice
//              the row pitch).  64 KB: two workgroups per CU.
//   step u     every wave: 4 DMA instructions (2 x 1 KB of A rows, 2 x 1 KB of B rows) of step u + 4 into stage u & 3 (read out during
//              step u - 1); 12 MFMAs (v_mfma_f32_32x32x16_f16, 3 per product) on the fragments of step u, between them the 16
//              transposed reads of step u + 1 from stage (u + 1) & 3; s_waitcnt vmcnt(8) (own loads of step u + 2 have landed);
//              one barrier.  Three steps (48 KB per workgroup) are in flight.
//   validity   the copies were written with scales predicted from the previous step's ranges; every workgroup checks this step's
//              ranges against them (shadow_usable) and, where the prediction failed (first step, a jump of more than 2^5 up or 2^7 down),
//              runs the fp32-MFMA tile on the fp32 tensors instead (same grid, same LDS size) -- slower, never wrong.
// ------------------------------------------------------------------------------------------------
// H1 (round 5, the 16-bit trainer): both operands are H1 tensors IN PLACE -- the retained activation and the gradient tensor themselves, plain
// NHWC fp16 with predicted power-of-two scales, no copies and no fp32 twins.  A pixel row of a stage is 16 cells of 16 bytes (256 B), ONE
// LDS-DMA instruction per operand, wave and step covers four pixel rows (lane >> 4) x 16 chunk positions (lane & 15), the transposed reads
// are the high-piece reads of the H2 image at half the row pitch, and a step is 4 MFMAs (one per product) instead of 12.  A tensor that
// left its predicted range raises fail_flag (the host repeats the step on the parity path): there is nothing to fall back to.
template <bool H1>
__device__ __forceinline__ void wgrad_dma_t(const WgradArgs& p) {
    constexpr unsigned ROW = H1 ? 256u : 512u, OPB = 16u * ROW, STG = 2u * OPB;      // pixel row, one operand of a stage, a stage (H2: 16 KB)
    constexpr unsigned EB = H1 ? 2u : 4u;         // bytes per channel of the operand tensors
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), half = lane >> 5, l31 = lane & 31;
    const float sx = shadow_scale_for(p.x_prev, lane), sy = shadow_scale_for(p.dy_prev, lane);
    if (!(shadow_usable(sx, p.x_cur, lane) && shadow_usable(sy, p.dy_cur, lane))) {
        if (!H1 && p.x && p.dy) wgrad_f32_body<2>(p);
        else if (p.fail_flag && threadIdx.x == 0) atomicOr(p.fail_flag, 2);       // the host repeats the step on fp32 tensors
        return;
    }
    const int wm = wave >> 1, wn = wave & 1;
    const int q0c = blockIdx.x * 16;               // first 8-channel cell of this k-tile
    const int n0 = blockIdx.y * 128;
    const int m_lo = blockIdx.z * p.m_per_block;
    const int m_hi = min(p.M, m_lo + p.m_per_block);
    if (m_hi <= m_lo) return;
    const int nsub = (((m_hi - m_lo + 15) >> 4) + 3) & ~3;      // 16-pixel steps, rounded up to the unroll (rows past m_hi read as zeros)
    const bool pointwise = p.ntaps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.H == p.Ho && p.W == p.Wo;
    const unsigned xb = H1 ? p.x_bytes >> 1 : p.x_bytes, dyb = H1 ? p.dy_bytes >> 1 : p.dy_bytes;
    const unsigned x_lim = pointwise ? (unsigned)min((long long)xb, (long long)m_hi * p.Cin * EB) : xb;
    const unsigned dy_lim = (unsigned)min((long long)dyb, (long long)m_hi * p.Cdy * EB);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xs), 0, (int)x_lim, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dys), 0, (int)dy_lim, 0x00020000);
    const float post = 1.f / (sx * sy);

    // ---- DMA lane map.  Instruction e (0, 1) of this wave fills stage rows R_e, R_e + 1 with R_e = 4 (2 (wave >> 1) + e) + 2 (wave & 1):
    // lanes 0-31 the first row, 32-63 the second; slot = lane & 31 -> plane (high / low) = slot >> 4, chunk position cs = slot & 15,
    // which holds logical cell cs ^ f(row), f(row) = ((row & 3) << 2) | ((row >> 2) & 3).  e only flips bit 0 of f: the two cells of a
    // lane are neighbours (same tap), its two pixel rows are 4 apart.
    // (H1: one instruction per operand covers rows 4 wave + (lane >> 4), chunk position lane & 15; f(row) = ((lane >> 4) << 2) | wave; e = 0 only)
    const int slot = lane & 31, plane = H1 ? 0 : slot >> 4, cs_ = H1 ? (lane & 15) : (slot & 15);
    const int f0 = H1 ? (((lane >> 4) << 2) | wave) : (((2 * (wave & 1) + half) << 2) | (2 * (wave >> 1)));
    const int row0 = H1 ? 4 * wave + (lane >> 4) : 4 * (2 * (wave >> 1)) + 2 * (wave & 1) + half;      // stage-local pixel row of instruction 0 (H2: instruction 1: + 4)
    constexpr int NE = H1 ? 1 : 2;                // DMA instructions per operand, wave and step
    const int lc8 = p.log2cin4 - 1;                                              // log2(Cin / 8)
    const int kcells = p.kchunks >> 1;
    int cellA[2], cellB[2];
    bool qok[2], cok[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c = cs_ ^ f0 ^ e;
        cellA[e] = q0c + c; cellB[e] = c;
        qok[e] = cellA[e] < kcells && (cellA[e] >> lc8) < p.ntaps;
        cok[e] = n0 + 8 * c < p.Cdy;
    }
    const int tap = cellA[0] >> lc8;               // (cells c, c ^ 1 lie in the same tap: Cin / 8 is even)
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int dh = kh * p.dil - p.pad_t, dw = kw * p.dil - p.pad_l;
    const int HoWo = p.Ho * p.Wo;
    const int cmask = (1 << lc8) - 1;
    unsigned offa[2], offb[2];
    int whi[2], wwi[2], mleft[2];
    const unsigned stepb = 16u * (unsigned)p.Cdy * EB, stepa_pw = 16u * (unsigned)p.Cin * EB;
    const int hlim = p.Ho * p.stride + dh, wlim = p.Wo * p.stride + dw;
    const unsigned d_px = (unsigned)(16 * p.stride * p.Cin) * EB;
    const unsigned d_row = (unsigned)((p.stride * p.W - p.Wo * p.stride) * p.Cin) * EB;
    const unsigned d_img = (unsigned)((p.H * p.W - p.Ho * p.stride * p.W) * p.Cin) * EB;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int m = m_lo + row0 + 4 * e;
        const unsigned cha = (unsigned)((cellA[e] & cmask) * 8) * EB + 16u * (unsigned)plane;
        offb[e] = cok[e] ? (unsigned)(m * p.Cdy + n0 + 8 * cellB[e]) * EB + 16u * (unsigned)plane : OOBT;
        if (pointwise) {
            offa[e] = qok[e] ? (unsigned)(m * p.Cin) * EB + cha : OOBT;
            whi[e] = wwi[e] = mleft[e] = 0;
        } else {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            whi[e] = ho * p.stride + dh;
            wwi[e] = wo * p.stride + dw;
            offa[e] = (unsigned)(((n * p.H + whi[e]) * p.W + wwi[e]) * p.Cin) * EB + cha;
            mleft[e] = m_hi - m;
        }
    }
    typedef __attribute__((address_space(3))) void lds_void;
    char* const dst0 = smem + (unsigned)__builtin_amdgcn_readfirstlane((H1 ? 4 * wave : 4 * (2 * (wave >> 1)) + 2 * (wave & 1)) * (int)ROW);
    // the four DMA instructions of one step into stage ST (A e = 0, A e = 1, B e = 0, B e = 1), then the walkers move on by 16 pixels
    auto dma_a = [&](char* stage, int e) {
        unsigned off;
        if (pointwise) {
            off = offa[e];
            if (qok[e]) offa[e] += stepa_pw;
        } else {
            const bool ok = qok[e] && mleft[e] > 0 && (unsigned)whi[e] < (unsigned)p.H && (unsigned)wwi[e] < (unsigned)p.W;
            off = ok ? offa[e] : OOBT;
            mleft[e] -= 16;
            wwi[e] += 16 * p.stride;
            offa[e] += d_px;
            while (wwi[e] >= wlim) {
                wwi[e] -= p.Wo * p.stride;
                whi[e] += p.stride;
                offa[e] += d_row;
                if (whi[e] >= hlim) { whi[e] -= p.Ho * p.stride; offa[e] += d_img; }
            }
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(stage + e * (4 * ROW)), 16, (int)off, 0, 0, 0);
    };
    auto dma_b = [&](char* stage, int e) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_void*)(stage + OPB + e * (4 * ROW)), 16, (int)offb[e], 0, 0, 0);
        if (cok[e]) offb[e] += stepb;
    };

    // transposed-read bases [operand][block b][pixel half h] (wgrad_h3p's map at the 512-byte row pitch)
    const int g16 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int f_lane = (qq << 2) | (2 * (g16 >> 1));
    const unsigned rrow = (unsigned)(size_t)smem + ROW * (unsigned)(8 * (g16 >> 1) + qq) + 8u * (pp & 1);
    unsigned rbase[2][2][2];
#pragma unroll
    for (int op = 0; op < 2; ++op) {
        const int L = (8 * (op == 0 ? wm : wn) + 2 * (g16 & 1) + (pp >> 1)) ^ f_lane;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) rbase[op][b][h] = rrow + 16u * (unsigned)(L ^ (4 * b) ^ h);
    }
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    WgFrag F0[2][2], F1[2][2];          // [operand][block]

    // column sums of dY (d beta), workgroups of the first k-tile only: thread t owns stage row t >> 4, chunk position t & 15
    const bool do_colsum = p.colsum != nullptr && blockIdx.x == 0;
    float cs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned cs_off = OPB + ROW * (unsigned)(t >> 4) + 16u * (unsigned)(t & 15);
    auto colsum_stage = [&](const char* stage) {
        const half8 h = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(stage + cs_off));
        if constexpr (H1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) cs8[k] += (float)h[k];
        } else {
        const half8 l = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(stage + cs_off + 256));
#pragma unroll
        for (int k = 0; k < 8; ++k) cs8[k] += (float)h[k] + (float)l[k];
        }
    };

#define WD_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WD_READ2(F, ST, R)                                                                                           \
    do {                                                                                                             \
        constexpr int op_ = ((R) >> 2) & 1, b_ = ((R) >> 1) & 1, pc_ = (R) & 1;                                      \
        constexpr unsigned o_ = (ST) * STG + op_ * OPB + pc_ * 256u;                                                 \
        F[op_][b_].v[pc_][0] = trr<o_>(rbase[op_][b_][0]);                                                           \
        F[op_][b_].v[pc_][1] = trr<o_ + 4u * ROW>(rbase[op_][b_][1]);                                                \
    } while (0)
#define WD_MMA(F, G)                                                                                                 \
    do {                                                                                                             \
        constexpr int blk_ = (G) & 3, i_ = blk_ >> 1, j_ = blk_ & 1, term_ = (G) >> 2;                               \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wg_op(F[0][i_], term_ == 0 ? 1 : 0),                    \
                                                             wg_op(F[1][j_], term_ == 1 ? 1 : 0), acc[i_][j_], 0, 0, 0); \
    } while (0)

    // prologue: steps 0..3 in flight, steps 0 and 1 landed, fragments of step 0
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        char* st = dst0 + k * STG;
        dma_a(st, 0); if constexpr (!H1) dma_a(st, 1);
        dma_b(st, 0); if constexpr (!H1) dma_b(st, 1);
    }
    if constexpr (H1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    WD_READ2(F0, 0, 0); WD_READ2(F0, 0, 2); WD_READ2(F0, 0, 4); WD_READ2(F0, 0, 6);
    if constexpr (!H1) { WD_READ2(F0, 0, 1); WD_READ2(F0, 0, 3); WD_READ2(F0, 0, 5); WD_READ2(F0, 0, 7); }
    if (do_colsum) colsum_stage(smem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // stage 0 has been read out: step 4 may land in it
    WD_FENCE();

    // one step on the fragments F (step u, u & 3 == S): DMA of step u + 4 into stage S, fragments of step u + 1 from stage S + 1 into Fn
#define WD_STEP(S, F, Fn)                                                                                            \
    do {                                                                                                             \
        constexpr int NS_ = ((S) + 1) & 3;                                                                           \
        char* st_ = dst0 + (S) * STG;                                                                                \
        dma_a(st_, 0); WD_FENCE();                                                                                   \
        WD_MMA(F, 0);  WD_FENCE(); WD_READ2(Fn, NS_, 0); WD_FENCE();                                                 \
        dma_a(st_, 1); WD_FENCE();                                                                                   \
        WD_MMA(F, 1);  WD_FENCE(); WD_READ2(Fn, NS_, 1); WD_FENCE();                                                 \
        dma_b(st_, 0); WD_FENCE();                                                                                   \
        WD_MMA(F, 2);  WD_FENCE(); WD_READ2(Fn, NS_, 2); WD_FENCE();                                                 \
        dma_b(st_, 1); WD_FENCE();                                                                                   \
        WD_MMA(F, 3);  WD_FENCE(); WD_READ2(Fn, NS_, 3); WD_FENCE();                                                 \
        WD_MMA(F, 4);  WD_FENCE(); WD_READ2(Fn, NS_, 4); WD_FENCE();                                                 \
        WD_MMA(F, 5);  WD_FENCE(); WD_READ2(Fn, NS_, 5); WD_FENCE();                                                 \
        WD_MMA(F, 6);  WD_FENCE(); WD_READ2(Fn, NS_, 6); WD_FENCE();                                                 \
        WD_MMA(F, 7);  WD_FENCE(); WD_READ2(Fn, NS_, 7); WD_FENCE();                                                 \
        WD_MMA(F, 8);  WD_FENCE();                                                                                   \
        if (do_colsum) colsum_stage(smem + NS_ * STG);                                                               \
        WD_FENCE();                                                                                                  \
        WD_MMA(F, 9);  WD_FENCE();                                                                                   \
        WD_MMA(F, 10); WD_FENCE();                                                                                   \
        WD_MMA(F, 11); WD_FENCE();                                                                                   \
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_s_barrier();                                                                                \
        WD_FENCE();                                                                                                  \
    } while (0)

    // H1: the same step with half of everything but the barrier: 2 DMA instructions, the 4 high-piece read pairs, 4 MFMAs (high x high)
#define WD_STEP1(S, F, Fn)                                                                                           \
    do {                                                                                                             \
        constexpr int NS_ = ((S) + 1) & 3;                                                                           \
        char* st_ = dst0 + (S) * STG;                                                                                \
        dma_a(st_, 0); WD_FENCE();                                                                                   \
        WD_MMA(F, 8);  WD_FENCE(); WD_READ2(Fn, NS_, 0); WD_FENCE();                                                 \
        dma_b(st_, 0); WD_FENCE();                                                                                   \
        WD_MMA(F, 9);  WD_FENCE(); WD_READ2(Fn, NS_, 2); WD_FENCE();                                                 \
        WD_MMA(F, 10); WD_FENCE(); WD_READ2(Fn, NS_, 4); WD_FENCE();                                                 \
        WD_MMA(F, 11); WD_FENCE(); WD_READ2(Fn, NS_, 6); WD_FENCE();                                                 \
        if (do_colsum) colsum_stage(smem + NS_ * STG);                                                               \
        WD_FENCE();                                                                                                  \
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                                  \
        __builtin_amdgcn_s_barrier();                                                                                \
        WD_FENCE();                                                                                                  \
    } while (0)
    for (int u = 0; u < nsub; u += 4) {
        if constexpr (H1) {
            WD_STEP1(0, F0, F1);
            WD_STEP1(1, F1, F0);
            WD_STEP1(2, F0, F1);
            WD_STEP1(3, F1, F0);
        } else {
        WD_STEP(0, F0, F1);
        WD_STEP(1, F1, F0);
        WD_STEP(2, F0, F1);
        WD_STEP(3, F1, F0);
        }
    }
#undef WD_STEP1
#undef WD_STEP
#undef WD_MMA
#undef WD_READ2
#undef WD_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the loads past the end (zeros) still write LDS

    if (do_colsum) {
        // (the colsum of steps nsub.. read zeros; step 0 was added in the prologue)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);       // [16 rows][128 columns]
        const int r = t >> 4, c = (t & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
#pragma unroll
        for (int k = 0; k < 8; ++k) red[r * 128 + 8 * c + k] = cs8[k];
        __syncthreads();
        if (t < 128 && n0 + t < p.Cdy) {
            float v = 0.f;
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) v += red[r2 * 128 + t];
            atomicAdd(p.colsum + n0 + t, v / sy);
        }
    }
    // C/D layout: col = lane&31 (co), row = (r&3) + 8*(r>>2) + 4*half (k)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = n0 + wn * 64 + 32 * j + l31;
        if (co >= p.Cdy) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = q0c * 8 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (k < p.kchunks * 4) atomicAdd(p.dw + (long long)k * p.Cdy + co, acc[i][j][r] * post);
            }
    }
}

__global__ __launch_bounds__(256, 2) void wgrad_dma(const WgradArgs p) { wgrad_dma_t<false>(p); }
__global__ __launch_bounds__(256, 2) void wgrad_dma_h1(const WgradArgs p) { wgrad_dma_t<true>(p); }

// fp32 [n][C] -> fp16 high / low cells with the scale shadow_scale_for predicts from `prev` (layer-level entry point only: inside the
// training step the producers' epilogues write the copies)
__global__ __launch_bounds__(256) void f32_to_shadow_kernel(const float4* __restrict__ x, long long n8, const float* __restrict__ prev,
                                                            uint4* __restrict__ out) {
    const float s = shadow_scale_for(prev, threadIdx.x & 63);
    if (!(s > 0.f)) return;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n8; g += (long long)gridDim.x * 256) {
        const float4 a = x[2 * g], b = x[2 * g + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint4 hi, lo;
        h2_pack8(v, s, hi, lo);
        out[2 * g] = hi; out[2 * g + 1] = lo;
    }
}

// H2 tensors with predicted scales (ConvArgs::*_scale_dev): after the pass, every such tensor's measured range must lie inside the
// window its predicted scale covers, else flag |= 1 and the host repeats the step on fp32 tensors.  One wave per listed slot.
struct H2CheckList { int n; short idx[254]; };
__global__ __launch_bounds__(64) void h2_pred_check_kernel(const H2CheckList list, const float* __restrict__ pool,
                                                           const float* __restrict__ prev, int* __restrict__ flag) {
    const int k = blockIdx.x;
    if (k >= list.n) return;
    const int lane = threadIdx.x;
    const float* pv = prev + (size_t)list.idx[k] * ABSMAX_SLOTS;
    const float* cu = pool + (size_t)list.idx[k] * ABSMAX_SLOTS;
    // (bits 8..: slot + 1 of a failed tensor -- diagnostics; any non-zero value means "repeat the step")
    if (!shadow_usable(shadow_scale_for(pv, lane), cu, lane) && lane == 0) atomicOr(flag, 1 | ((list.idx[k] + 1) << 8));
}

// H2 cells with a predicted scale -> fp32 (the heads' weight gradient and gate read the block4 features as fp32)
__global__ __launch_bounds__(256) void h2_to_f32_pred_kernel(const uint4* __restrict__ x, long long n8, const float* __restrict__ prev,
                                                             float4* __restrict__ out) {
    const float s = shadow_scale_for(prev, threadIdx.x & 63);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n8; g += (long long)gridDim.x * 256) {
        float v[8];
        h2_unpack8(x[2 * g], x[2 * g + 1], inv, v);
        out[2 * g] = make_float4(v[0], v[1], v[2], v[3]);
        out[2 * g + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// 16-bit tier: H1 cells with a predicted scale -> fp32 (the block4 features for the heads' backward; the data gradient that leaves the
// H1 units for block1).  gate != nullptr: an H1 tensor of the same shape, the output is zeroed where its half is not positive.
__global__ __launch_bounds__(256) void h1_to_f32_pred_kernel(const uint4* __restrict__ x, long long n8, const float* __restrict__ prev,
                                                             float4* __restrict__ out, const uint4* __restrict__ gate) {
    const float s = shadow_scale_for(prev, threadIdx.x & 63);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n8; g += (long long)gridDim.x * 256) {
        float v[8];
        h1_unpack8(x[g], inv, v);
        if (gate) {
            const half8 q = __builtin_bit_cast(half8, gate[g]);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (float)q[k] > 0.f ? v[k] : 0.f;
        }
        out[2 * g] = make_float4(v[0], v[1], v[2], v[3]);
        out[2 * g + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}
// fp32 -> H1 cells with the scale predicted from `prev` (the heads' data gradient enters the H1 units)
__global__ __launch_bounds__(256) void f32_to_h1_pred_kernel(const float4* __restrict__ x, long long n8, const float* __restrict__ prev,
                                                             uint4* __restrict__ out) {
    const float s = shadow_scale_for(prev, threadIdx.x & 63);
    if (!(s > 0.f)) return;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < n8; g += (long long)gridDim.x * 256) {
        const float4 a = x[2 * g], b = x[2 * g + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        out[g] = h1_pack8(v, s);
    }
}
// zero-stuffed copy of an H1 gradient tensor [N, Ho, Wo, C] -> [N, 2 Ho, 2 Wo, C] (data at the even positions; `out` zeroed by the caller):
// the data gradient of a stride-2 conv then is a plain stride-1 conv on the cell kernels (the gather form `up = 2` has no LDS-DMA loader)
__global__ __launch_bounds__(256) void h1_zero_stuff_kernel(const uint4* __restrict__ x, int N, int Ho, int Wo, int C8, uint4* __restrict__ out) {
    const long long total = (long long)N * Ho * Wo * C8;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int c = (int)(g % C8);
        const long long px = g / C8;
        const int wo = (int)(px % Wo), ho = (int)((px / Wo) % Ho), n = (int)(px / ((long long)Wo * Ho));
        out[(((long long)n * 2 * Ho + 2 * ho) * (2 * Wo) + 2 * wo) * C8 + c] = x[g];
    }
}

// dW = scale * dWraw (HWIO, real Cin) and dot[co] += sum_k W[k][co] * dWraw[k][co]   (grid: k-chunks x co-tiles of 64)
__global__ __launch_bounds__(256) void scale_dw_dot_kernel(const float* __restrict__ dwraw, const float* __restrict__ w,
                                                           const float* __restrict__ scale, int taps, int cin, int cin_real,
                                                           int cout, int rows_per_block, float* __restrict__ dW,
                                                           float* __restrict__ dot) {
    __shared__ float part[4][64];
    const int co = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int krows = taps * cin_real;
    const int k0 = blockIdx.y * rows_per_block, k1 = min(krows, k0 + rows_per_block);
    float acc = 0.f;
    if (co < cout) {
        const float s = scale[co];
        for (int k = k0 + rl; k < k1; k += 4) {
            const int tp = k / cin_real, ci = k - tp * cin_real;
            const float g = dwraw[((long long)tp * cin + ci) * cout + co];
            const long long o = (long long)k * cout + co;
            acc += w[o] * g;
            dW[o] = s * g;
        }
    }
    part[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && co < cout)
        atomicAdd(dot + co, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// dgamma = r * (dot - mean * dbeta), dbeta = colsum           (r = rsqrt(var + eps))
__global__ void bn_param_grads_kernel(const float* __restrict__ dot, const float* __restrict__ colsum,
                                      const float* __restrict__ mean, const float* __restrict__ var, float eps, int cout,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= cout) return;
    const float r = 1.f / sqrtf(var[co] + eps);
    const float db = colsum[co];
    dbeta[co] = db;
    dgamma[co] = r * (dot[co] - mean[co] * db);
}

// The two kernels above for ALL non-head layers in one launch each (blockIdx.z = layer; offsets are relative to the workspace,
// which the caller may move between steps): every layer keeps its own dWraw / colsum region, so nothing has to be finalised
// between two layers' weight-gradient GEMMs.
struct FinDesc {
    long long dw_off, cs_off;              // bytes from the workspace base
    long long w_off, g_off, b_off, mean_off, var_off;
    int taps, cin, cin_real, cout;
    const float* d_scale;
};
__global__ __launch_bounds__(256) void scale_dw_dot_all_kernel(const FinDesc* __restrict__ table, const char* __restrict__ ws,
                                                               const float* __restrict__ params, float* __restrict__ grads,
                                                               int rows_per_block) {
    // a thread owns 4 adjacent output channels (16-byte accesses; every layer but the heads has Cout % 4 == 0 and 16-byte aligned
    // offsets) of every 16th row of the block's row range
    __shared__ float4 part[16][16];
    const FinDesc d = table[blockIdx.z];
    const int krows = d.taps * d.cin_real;
    if ((int)blockIdx.x * 64 >= d.cout || (int)blockIdx.y * rows_per_block >= krows) return;
    const float* dwraw = reinterpret_cast<const float*>(ws + d.dw_off);
    float* dot = reinterpret_cast<float*>(const_cast<char*>(ws) + d.cs_off) + d.cout;
    const float* w = params + d.w_off;
    float* dW = grads + d.w_off;
    const int c4 = threadIdx.x & 15, co = blockIdx.x * 64 + 4 * c4, rl = threadIdx.x >> 4;
    const int k0 = blockIdx.y * rows_per_block, k1 = min(krows, k0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (co < d.cout) {
        const float4 sc = *reinterpret_cast<const float4*>(d.d_scale + co);
#pragma unroll 4
        for (int k = k0 + rl; k < k1; k += 16) {
            const int tp = k / d.cin_real, ci = k - tp * d.cin_real;
            const float4 g = *reinterpret_cast<const float4*>(dwraw + ((long long)tp * d.cin + ci) * d.cout + co);
            const long long o = (long long)k * d.cout + co;
            const float4 wv = *reinterpret_cast<const float4*>(w + o);
            acc.x += wv.x * g.x; acc.y += wv.y * g.y; acc.z += wv.z * g.z; acc.w += wv.w * g.w;
            *reinterpret_cast<float4*>(dW + o) = make_float4(sc.x * g.x, sc.y * g.y, sc.z * g.z, sc.w * g.w);
        }
    }
    part[rl][c4] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < (unsigned)d.cout) {
        const int q = threadIdx.x >> 2, e = threadIdx.x & 3;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += reinterpret_cast<const float*>(&part[r][q])[e];
        atomicAdd(dot + blockIdx.x * 64 + threadIdx.x, sum);
    }
}
__global__ void bn_param_grads_all_kernel(const FinDesc* __restrict__ table, const char* __restrict__ ws, const float* __restrict__ stats,
                                          float eps, float* __restrict__ grads) {
    const FinDesc d = table[blockIdx.y];
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= d.cout) return;
    const float* colsum = reinterpret_cast<const float*>(ws + d.cs_off);
    const float r = 1.f / sqrtf(stats[d.var_off + co] + eps);
    const float db = colsum[co];
    grads[d.b_off + co] = db;
    grads[d.g_off + co] = r * (colsum[d.cout + co] - stats[d.mean_off + co] * db);
}

// head: dw[ka][kb][c][ci] = dW'raw[(khp,kwp)][ci][(a,b),c] (each w element appears once), db[c] = sum_phases colsum
// (cpad: columns per row of dwraw; col0: first column of this head -- the 16-bit tier's merged launch holds both heads side by side)
__global__ void finalize_head_grads(const float* __restrict__ dwraw, const float* __restrict__ colsum, int njt, int cin,
                                    int cpad, int col0, float* __restrict__ dw, float* __restrict__ db) {
    const long long total = 9ll * njt * cin;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(g % cin);
        long long r = g / cin;
        const int c = (int)(r % njt);
        r /= njt;
        const int kb = (int)(r % 3), ka = (int)(r / 3);
        // ka = a + 2 - 2 khp  ->  (ka=0: a=0,khp=1) (ka=1: a=1,khp=1) (ka=2: a=0,khp=0)
        const int a = ka == 1 ? 1 : 0, khp = ka == 2 ? 0 : 1;
        const int b = kb == 1 ? 1 : 0, kwp = kb == 2 ? 0 : 1;
        dw[g] = dwraw[(((long long)(khp * 2 + kwp)) * cin + ci) * cpad + col0 + (a * 2 + b) * njt + c];
    }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < njt) db[c] = colsum[col0 + c] + colsum[col0 + njt + c] + colsum[col0 + 2 * njt + c] + colsum[col0 + 3 * njt + c];
}

// d scoremap [N,2h,2w,njt] -> phase-major rows [N*h*w, cpad] (inverse of the forward scatter), zero padded
// (rng: range slots that take max |value| -- one array for BOTH heads' gathers: it predicts the scale of the merged H1 tensor that
// head_gather_h1_kernel writes in the next 16-bit pass)
__global__ void head_gather_kernel(const float* __restrict__ dsc, int N, int h, int w, int njt, int cpad,
                                   float* __restrict__ out, float* __restrict__ rng) {
    const long long total = (long long)N * h * w * cpad;
    float mx = 0.f;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(g % cpad);
        long long m = g / cpad;
        float v = 0.f;
        if (co < 4 * njt) {
            const int ph = co / njt, c = co - ph * njt;
            const int j = (int)(m % w);
            m /= w;
            const int i = (int)(m % h);
            const int n = (int)(m / h);
            v = dsc[(((long long)n * 2 * h + 2 * i + (ph >> 1)) * 2 * w + 2 * j + (ph & 1)) * njt + c];
        }
        out[g] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    pack_track(rng, mx);
}

// 16-bit tier: both heads' loss gradients -> ONE phase-major H1 tensor [N*h*w, CT].  Columns [0, cpad0): the part head's phases as
// head_gather_kernel lays them out, [cpad0, cpad0 + cpad1): the locref head's, zero up to CT (a multiple of 64: one K-step of the H1
// kernels).  Scale: predicted from the range the same values had one step ago (prev); this step's range goes to rng.  One data-gradient
// launch and one weight-gradient launch then serve both heads.
__global__ __launch_bounds__(256) void head_gather_h1_kernel(const float* __restrict__ dsc0, const float* __restrict__ dsc1, int N, int h, int w,
                                                             int njt0, int cpad0, int njt1, int cpad1, int CT, const float* __restrict__ prev,
                                                             uint4* __restrict__ out, float* __restrict__ rng) {
    const float scale = shadow_scale_for(prev, threadIdx.x & 63);
    const int G = CT >> 3;
    const long long total = (long long)N * h * w * G;
    float mx = 0.f;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int cg = (int)(g % G);
        long long m = g / G;
        const int j = (int)(m % w);
        m /= w;
        const int i = (int)(m % h);
        const int n = (int)(m / h);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int co = cg * 8 + k;
            const float* dsc = dsc0;
            int njt = njt0;
            if (co >= cpad0) { co -= cpad0; dsc = dsc1; njt = njt1; if (co >= cpad1) njt = 0; }
            v[k] = 0.f;
            if (co < 4 * njt) {
                const int ph = co / njt, c = co - ph * njt;
                v[k] = dsc[(((long long)n * 2 * h + 2 * i + (ph >> 1)) * 2 * w + 2 * j + (ph & 1)) * njt + c];
            }
            mx = fmaxf(mx, fabsf(v[k]));
        }
        out[g] = h1_pack8(v, scale);
    }
    pack_track(rng, mx);
}

// max-pool 3x3/2 SAME backward fused with the stem's ReLU gate: dC1 = (C1 > 0) * sum over windows whose first
// maximum is this element of dPool.
// Training forward pool: the same maxima as maxpool3x3s2_same plus, per window and channel, the position (a * 3 + b) of its FIRST
// maximum in row-major order -- the element TF's MaxPoolGrad (and the re-scan of maxpool_bwd_kernel below) routes the gradient to.
__global__ __launch_bounds__(256) void maxpool_fwd_idx_kernel(const float* __restrict__ x, int N, int H, int W, int C4, int Ho, int Wo,
                                                              int pt, int pl, float* __restrict__ y, uchar4* __restrict__ idx,
                                                              uint2* __restrict__ y_h1 = nullptr, const float* __restrict__ h1_prev = nullptr) {
    // y_h1 (16-bit tier): an H1 copy of the pool output with the scale predicted from the previous step's range of conv1's output
    const float sh1 = y_h1 ? shadow_scale_for(h1_prev, threadIdx.x & 63) : 0.f;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(g % C4);
        long long pix = g / C4;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 k = make_uchar4(255, 255, 255, 255);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int hi = ho * 2 - pt + a;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int wi = wo * 2 - pl + b;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((((long long)n * H + hi) * W + wi) * C4 + c4) * 4);
                const unsigned char p = (unsigned char)(a * 3 + b);
                if (v.x > m.x || k.x == 255) { m.x = v.x; k.x = p; }        // strictly greater: the first maximum stays
                if (v.y > m.y || k.y == 255) { m.y = v.y; k.y = p; }
                if (v.z > m.z || k.z == 255) { m.z = v.z; k.z = p; }
                if (v.w > m.w || k.w == 255) { m.w = v.w; k.w = p; }
            }
        }
        *reinterpret_cast<float4*>(y + g * 4) = m;
        idx[g] = k;
        if (sh1 > 0.f) {
            uint2 hh_, ll_;
            split2_f16(m, sh1, hh_, ll_);
            y_h1[g] = hh_;
        }
    }
}

// Backward of that pool (+ the stem's ReLU gate): a pixel collects the gradient of every window (<= 4) whose recorded first maximum it
// is, in the same window order as maxpool_bwd_kernel below -- bitwise the same result with ~5 loads per pixel instead of ~33.
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const float* __restrict__ c1, const float* __restrict__ dpool,
                                                              const uchar4* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo,
                                                              int pt, int pl, float* __restrict__ dc1) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(g % C4);
        long long r = g / C4;
        const int wi = (int)(r % W);
        r /= W;
        const int hi = (int)(r % H);
        const int n = (int)(r / H);
        // c1 == nullptr (16-bit tier, fused root block): no conv1 map; dpool is already zero wherever the window's maximum is not positive
        const float4 v = c1 ? *reinterpret_cast<const float4*>(c1 + g * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v.x > 0.f || v.y > 0.f || v.z > 0.f || v.w > 0.f) {
            for (int ho = max(0, (hi + pt - 2 + 1) / 2); ho <= min(Ho - 1, (hi + pt) / 2); ++ho)
                for (int wo = max(0, (wi + pl - 2 + 1) / 2); wo <= min(Wo - 1, (wi + pl) / 2); ++wo) {
                    const long long o = (((long long)n * Ho + ho) * Wo + wo) * C4 + c4;
                    const uchar4 k = idx[o];
                    const unsigned char me = (unsigned char)((hi - (ho * 2 - pt)) * 3 + (wi - (wo * 2 - pl)));
                    if (k.x != me && k.y != me && k.z != me && k.w != me) continue;
                    const float4 d = *reinterpret_cast<const float4*>(dpool + o * 4);
                    if (k.x == me) acc.x += d.x;
                    if (k.y == me) acc.y += d.y;
                    if (k.z == me) acc.z += d.z;
                    if (k.w == me) acc.w += d.w;
                }
            if (!(v.x > 0.f)) acc.x = 0.f;        // ReLU gate of the stem
            if (!(v.y > 0.f)) acc.y = 0.f;
            if (!(v.z > 0.f)) acc.z = 0.f;
            if (!(v.w > 0.f)) acc.w = 0.f;
        }
        *reinterpret_cast<float4*>(dc1 + g * 4) = acc;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ c1, const float* __restrict__ dpool, int N, int H,
                                                          int W, int C, int Ho, int Wo, int pt, int pl, float* __restrict__ dc1) {
    // one thread per pixel and 4 channels (16-byte loads); <= 4 windows contain a pixel, each re-scanned for its first maximum
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(g % C4);
        long long r = g / C4;
        const int wi = (int)(r % W);
        r /= W;
        const int hi = (int)(r % H);
        const int n = (int)(r / H);
        const float4 v = *reinterpret_cast<const float4*>(c1 + g * 4);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v.x > 0.f || v.y > 0.f || v.z > 0.f || v.w > 0.f) {
            for (int ho = max(0, (hi + pt - 2 + 1) / 2); ho <= min(Ho - 1, (hi + pt) / 2); ++ho)
                for (int wo = max(0, (wi + pl - 2 + 1) / 2); wo <= min(Wo - 1, (wi + pl) / 2); ++wo) {
                    // is this pixel the FIRST maximum (row-major) of window (ho, wo), per channel?  It is iff no earlier
                    // element is >= it and no later element is > it.
                    bool fx = true, fy = true, fz = true, fw = true;
                    for (int a = 0; a < 3; ++a) {
                        const int hh = ho * 2 - pt + a;
                        if ((unsigned)hh >= (unsigned)H) continue;
                        for (int b = 0; b < 3; ++b) {
                            const int ww = wo * 2 - pl + b;
                            if ((unsigned)ww >= (unsigned)W || (hh == hi && ww == wi)) continue;
                            const float4 u = *reinterpret_cast<const float4*>(c1 + ((((long long)n * H + hh) * W + ww) * C4 + c4) * 4);
                            const bool before = hh < hi || (hh == hi && ww < wi);
                            fx = fx && (before ? u.x < v.x : u.x <= v.x);
                            fy = fy && (before ? u.y < v.y : u.y <= v.y);
                            fz = fz && (before ? u.z < v.z : u.z <= v.z);
                            fw = fw && (before ? u.w < v.w : u.w <= v.w);
                        }
                    }
                    const float4 d = *reinterpret_cast<const float4*>(dpool + ((((long long)n * Ho + ho) * Wo + wo) * C4 + c4) * 4);
                    if (fx) acc.x += d.x;
                    if (fy) acc.y += d.y;
                    if (fz) acc.z += d.z;
                    if (fw) acc.w += d.w;
                }
            if (!(v.x > 0.f)) acc.x = 0.f;        // ReLU gate of the stem
            if (!(v.y > 0.f)) acc.y = 0.f;
            if (!(v.z > 0.f)) acc.z = 0.f;
            if (!(v.w > 0.f)) acc.w = 0.f;
        }
        *reinterpret_cast<float4*>(dc1 + g * 4) = acc;
    }
}

// Stem panel [49 taps][CoutP][4] -> the row panel of the fused root kernel [7 kernel rows x 8 pixels][CoutP][4]: pixel 7 is zero, and channel
// slot 3 carries sum_c (round(mean_c) - mean_c) w_c, the weight of the kernel's constant fourth input channel (dgp_net_load_weights builds
// the same panel on the host, dgp_net.hip).
__global__ __launch_bounds__(256) void stem_rows_kernel(const float4* __restrict__ panel, int coutP, float d0, float d1, float d2,
                                                        float4* __restrict__ rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 56 * coutP) return;
    const int r = i / coutP, co = i - r * coutP;
    const int kh = r >> 3, kw = r & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kw < 7) {
        v = panel[(size_t)(kh * 7 + kw) * coutP + co];
        v.w = d0 * v.x + d1 * v.y + d2 * v.z;
    }
    rows[i] = v;
}

// ------------------------------------------------------------------------------------------------
// optimiser: global-norm clip + momentum SGD over the flat trainable buffer
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ out) {
    __shared__ double part[4];
    double s = 0.0;
    const long long n4 = n >> 2;                 // 16-byte loads (hipMalloc'ed buffer), the tail one by one
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        s += (double)v.x * (double)v.x + (double)v.y * (double)v.y + (double)v.z * (double)v.z + (double)v.w * (double)v.w;
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        s += (double)g[i] * (double)g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

__global__ void momentum_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v, long long n,
                                float lr, float mom, float clip, const double* __restrict__ sumsq, double* __restrict__ sumsq_next,
                                float* __restrict__ gnorm_out, const int* __restrict__ skip) {
    // skip: the range flag of a 16-bit / fast pass -- set: that pass's gradients are invalid, nothing is updated (the host repeats the step)
    // sumsq_next: the accumulator the NEXT step's sumsq_kernel adds into (nobody reads it now): zeroed here instead of by a fill launch
    const float gn = (float)sqrt(*sumsq);
    if (blockIdx.x == 0 && threadIdx.x == 0) *sumsq_next = 0.0;
    if (skip && *skip) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm_out) *gnorm_out = gn;
        return;
    }
    const float scale = clip > 0.f ? clip / fmaxf(gn, clip) : 1.f;      // tf.clip_by_global_norm; clip <= 0: none
    if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm_out) *gnorm_out = gn;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float a = mom * v[i] + g[i] * scale;            // accum = momentum * accum + grad
        v[i] = a;
        w[i] -= lr * a;                                       // var -= lr * accum
    }
}

__global__ void step_status_kernel(const float* __restrict__ losses, int n_losses, const float* __restrict__ gnorm, const int* __restrict__ flag,
                                   float* __restrict__ host) {
    const int t = threadIdx.x;
    if (t < n_losses) host[t] = losses[t];
    if (t == 8) host[8] = *gnorm;
    if (t == 9) host[9] = flag ? (float)*flag : 0.f;
    __threadfence_system();
}

// ------------------------------------------------------------------------------------------------
// 16-bit tier: the stem's weight gradient with the max-pool's backward fused into its loader (round 5).
//   dW[(kh, kw, c)][co] = sum over conv1 pixels (n, y, x) of  in[n, 2y + kh - 3, 2x + kw - 3, c] * dC1[n, y, x, co]
//   dC1[n, y, x, co]    = sum over the <= 4 pool windows q that hold (y, x) of dPool[n, q, co] * [first maximum of (q, co) is (y, x)]
// The layer-by-layer form wrote dC1 as fp32 (216 MB at 11 frames), read it FOUR times (one per 64-row k-tile of the 64 x 64 fp32 tile) and
// ran on the fp32 matrix pipe: 0.30 ms, the last launch of the pass, + 0.13 ms for the pool's backward and an fp32 copy of dPool.  Here a
// workgroup walks items of 4 conv rows x 32 columns: the centred frame patch (13 x 69 pixels) goes to LDS as fp16, dPool (H1 cells, in
// place) and the first-maximum record of the (up to 4 x 18) pool windows that touch the item go to LDS, dC1 of the item is formed there as fp16
// [co][pixel], and the four waves (one 16-channel column block each) run v_mfma_f32_16x16x32_f16 over all 13 k-blocks (196 rows padded to
// 208) with the reduction over the item's 128 pixels -- dC1 never exists in memory, dPool and the record are read once.  The A operand
// (a k row x 8 pixels of one conv row: stride-2 input columns) is gathered with eight 2-byte LDS reads; a reduction group is 4 rows x 8
// columns so that the four k-groups of a wave read rows 2 * 74 pixels apart (8 banks: conflict-free).  Partial sums stay in registers over
// all items of the (persistent) workgroup and are added to dW with float atomics at the end, scaled by 1 / (dPool's predicted scale).
// ------------------------------------------------------------------------------------------------
constexpr int STEM_SLAB_ROW = 196 * 64 + 64;
struct StemWgradArgs {
    const float* x;              // centred frames [B, H, W, 4] fp32 (preprocess_u8)
    const uint4* g;              // dPool: H1 cells [B, HP, WP, 64]
    const unsigned char* idx;    // first-maximum positions [B, HP, WP, 64]
    const float* g_prev;         // range slots that predict dPool's scale
    float* slab;                 // [gridDim.x][196 x 64 + 64] fp32: every workgroup's partial dW and column sums (plain stores; 512 workgroups
                                 // adding to the SAME 50 KB with float atomics took 2.2 ms: 512 serialized updates per address)
    float* dw;                   // stem_wgrad_reduce_kernel: [49 taps x 4][64] fp32
    float* colsum;               // [64]
    int B, H, W, H1, W1, HP, WP, pt, pl, bands, chunks, nitems;
};

__global__ __launch_bounds__(256) void stem_wgrad_h1_kernel(const StemWgradArgs p) {
    constexpr int IN_R = 13, IN_C = 74, PR = 4, PC = 18, LDY = 136;      // (4 x 18 pool windows: 3 x 17 when the pool's SAME padding starts at 0, one more when it starts at 1)
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) uint2 sIn[IN_R * IN_C];                 // 4 halves per input pixel
    __shared__ __attribute__((aligned(16))) uint4 sG[PR * PC * 8];                  // dPool cells of the item's pool windows
    __shared__ __attribute__((aligned(16))) uint4 sI[PR * PC * 4];                  // their first-maximum bytes
    __shared__ __attribute__((aligned(16))) unsigned short sDy[64 * LDY];           // dC1 of the item, [co][pixel = row * 32 + column]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    floatx4 c[13];
#pragma unroll
    for (int kb = 0; kb < 13; ++kb) c[kb] = floatx4{0.f, 0.f, 0.f, 0.f};
    int kbase[13];                       // half index of this lane's k row inside the patch (-1: padding row)
#pragma unroll
    for (int kb = 0; kb < 13; ++kb) {
        const int kk = 16 * kb + l15, tap = kk >> 2, cc = kk & 3, kh = tap / 7, kw = tap - 7 * kh;
        kbase[kb] = kk < 196 ? (kh * IN_C + kw) * 4 + cc : -1;
    }
    const int o8 = t & 15, co4 = t >> 4;             // dC1 unit of this thread: pixels 8 o8 .. 8 o8 + 7 (one conv row), channels 4 co4 .. + 3
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* sInH = reinterpret_cast<const unsigned short*>(sIn);
    const unsigned short* sGH = reinterpret_cast<const unsigned short*>(sG);
    const unsigned char* sIB = reinterpret_cast<const unsigned char*>(sI);
    for (int item = blockIdx.x; item < p.nitems; item += gridDim.x) {
        const int n = item / (p.bands * p.chunks), rem = item - n * (p.bands * p.chunks);
        const int y0 = (rem / p.chunks) * 4, x0 = (rem % p.chunks) * 32;
        const int qy0 = (y0 + p.pt) / 2 - 1, qx0 = (x0 + p.pl) / 2 - 1;            // first pool window that can hold (y0, x0)
        // ---- centred frame patch -> fp16
        for (int i = t; i < IN_R * IN_C; i += 256) {
            const int r = i / IN_C, cidx = i - r * IN_C;
            const int gr = 2 * y0 - 3 + r, gc = 2 * x0 - 3 + cidx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gr < (unsigned)p.H && (unsigned)gc < (unsigned)p.W)
                v = *reinterpret_cast<const float4*>(p.x + (((size_t)n * p.H + gr) * p.W + gc) * 4);
            const half2v a = {(_Float16)v.x, (_Float16)v.y}, b = {(_Float16)v.z, (_Float16)v.w};
            sIn[i] = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
        }
        // ---- dPool cells and first-maximum record of the 3 x 17 windows
        for (int i = t; i < PR * PC * 8; i += 256) {
            const int lp = i >> 3, cell = i & 7, a = lp / PC, b = lp - a * PC;
            const int qy = qy0 + a, qx = qx0 + b;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)qy < (unsigned)p.HP && (unsigned)qx < (unsigned)p.WP) v = p.g[(((size_t)n * p.HP + qy) * p.WP + qx) * 8 + cell];
            sG[i] = v;
        }
        for (int i = t; i < PR * PC * 4; i += 256) {
            const int lp = i >> 2, part = i & 3, a = lp / PC, b = lp - a * PC;
            const int qy = qy0 + a, qx = qx0 + b;
            uint4 v = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
            if ((unsigned)qy < (unsigned)p.HP && (unsigned)qx < (unsigned)p.WP)
                v = *reinterpret_cast<const uint4*>(p.idx + ((((size_t)n * p.HP + qy) * p.WP + qx) * 64 + part * 16));
            sI[i] = v;
        }
        __syncthreads();
        // ---- dC1 of the item: this thread's 8 pixels x 4 channels
        {
            float dy[8][4];
            const int yy = o8 >> 2, xb = (o8 & 3) * 8;
            const int y = y0 + yy;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int x = x0 + xb + e;
                dy[e][0] = dy[e][1] = dy[e][2] = dy[e][3] = 0.f;
                if (y < p.H1 && x < p.W1) {
                    const int qya = max(0, (y + p.pt - 1) / 2), qyb = min(p.HP - 1, (y + p.pt) / 2);
                    const int qxa = max(0, (x + p.pl - 1) / 2), qxb = min(p.WP - 1, (x + p.pl) / 2);
                    for (int qy = qya; qy <= qyb; ++qy)
                        for (int qx = qxa; qx <= qxb; ++qx) {
                            const unsigned me = (unsigned)((y - (2 * qy - p.pt)) * 3 + (x - (2 * qx - p.pl)));
                            const int lp = (qy - qy0) * PC + (qx - qx0);
                            const unsigned k4 = *reinterpret_cast<const unsigned*>(sIB + lp * 64 + 4 * co4);
                            const uint2 g4 = *reinterpret_cast<const uint2*>(sGH + lp * 64 + 4 * co4);
                            const half2v g01 = __builtin_bit_cast(half2v, g4.x), g23 = __builtin_bit_cast(half2v, g4.y);
                            if ((k4 & 0xffu) == me) dy[e][0] += (float)g01[0];
                            if (((k4 >> 8) & 0xffu) == me) dy[e][1] += (float)g01[1];
                            if (((k4 >> 16) & 0xffu) == me) dy[e][2] += (float)g23[0];
                            if ((k4 >> 24) == me) dy[e][3] += (float)g23[1];
                        }
                }
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                unsigned w[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const half2v h = {(_Float16)dy[2 * e2][cc], (_Float16)dy[2 * e2 + 1][cc]};
                    w[e2] = __builtin_bit_cast(unsigned, h);
                    cs[cc] += (float)h[0] + (float)h[1];
                }
                *reinterpret_cast<uint4*>(sDy + (4 * co4 + cc) * LDY + 8 * o8) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        __syncthreads();
        // ---- MFMAs: reduction groups of 4 conv rows (k-group g) x 8 columns
#pragma unroll 1
        for (int blk = 0; blk < 4; ++blk) {
            const uint4 bfrag = *reinterpret_cast<const uint4*>(sDy + (16 * wave + l15) * LDY + 32 * g + 8 * blk);
            const int pbase = ((2 * g) * IN_C + 2 * (8 * blk)) * 4;
#pragma unroll
            for (int kb = 0; kb < 13; ++kb) {
                unsigned w[4] = {0u, 0u, 0u, 0u};
                if (kbase[kb] >= 0) {
                    const unsigned short* q = sInH + kbase[kb] + pbase;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) w[e2] = (unsigned)q[16 * e2] | ((unsigned)q[16 * e2 + 8] << 16);
                }
                c[kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, make_uint4(w[0], w[1], w[2], w[3])),
                                                               __builtin_bit_cast(half8, bfrag), c[kb], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- partial sums -> this workgroup's slab row (column sums: reduced over the 16 threads of a channel quad through LDS first)
    float* row = p.slab + (size_t)blockIdx.x * STEM_SLAB_ROW;
#pragma unroll
    for (int kb = 0; kb < 13; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = 16 * kb + 4 * g + r;
            if (kk < 196) row[kk * 64 + 16 * wave + l15] = c[kb][r];
        }
    float* red = reinterpret_cast<float*>(sDy);          // (free: the last item's MFMAs are behind a barrier)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) red[(4 * co4 + cc) * 16 + o8] = cs[cc];
    __syncthreads();
    if (t < 64) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[t * 16 + k];
        row[196 * 64 + t] = s;
    }
}

// sum of the workgroups' slab rows x 1 / (dPool's predicted scale) -> dW and the column sums (both zeroed by the pass): blockIdx.y sums
// one group of 32 rows with independent loads, the groups meet with float atomics (16-32 per address)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const StemWgradArgs p, int nrows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float sg = shadow_scale_for(p.g_prev, threadIdx.x & 63);
    if (i >= STEM_SLAB_ROW) return;
    const float inv = sg > 0.f ? 1.f / sg : 0.f;
    const int r0 = blockIdx.y * 32, r1 = min(nrows, r0 + 32);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        s0 += p.slab[(size_t)r * STEM_SLAB_ROW + i]; s1 += p.slab[(size_t)(r + 1) * STEM_SLAB_ROW + i];
        s2 += p.slab[(size_t)(r + 2) * STEM_SLAB_ROW + i]; s3 += p.slab[(size_t)(r + 3) * STEM_SLAB_ROW + i];
    }
    for (; r < r1; ++r) s0 += p.slab[(size_t)r * STEM_SLAB_ROW + i];
    const float s = ((s0 + s1) + (s2 + s3)) * inv;
    if (i < 196 * 64) atomicAdd(p.dw + i, s);
    else if (p.colsum) atomicAdd(p.colsum + (i - 196 * 64), s);
}

bool pool_idx_on() { static const bool v = (dgp_tune("DGP_POOL_IDX", 1) != 0); return v; }
int grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

// ================================================================================================
// trainer
// ================================================================================================
struct TLayer {
    long long w_off = -1, g_off = -1, b_off = -1;     // offsets into the flat trainable buffer (floats)
    long long mean_off = -1, var_off = -1;            // offsets into the frozen-statistics buffer
    int cin_real = 0;
    float* d_wT = nullptr;                            // data-gradient panels
    float* d_wTh3 = nullptr;                          // ... pre-split into fp16 high/low cells (launch_pack_h3), per sync
    void* d_wTh1 = nullptr;                           // ... as high-only H1 cells (tier 1)
    int nkT = 0, cinP = 0, cpad = 0;                  // cpad: padded phase-major channel count (heads)
};

// Per-trainer launch context (was process-global state: two trainers, or a trainer created after another was destroyed, must not
// see each other's range maps, fp16 cell tables or workspace pointers).  Bound to the calling thread for the duration of an entry point.
struct TPlanFwd;
struct RangeCtx {
    float* pool = nullptr;            // RANGE_POOL slot arrays of ABSMAX_SLOTS floats
    float* prev = nullptr;            // the pool as the previous pass of the same kind (forward / backward) left it: scales of the fp16 copies
    int next = 0, limit = 0;
    std::map<const void*, const float*> of;     // tensor / panel pointer -> its slot array
    // second forward chain (frames [n1, B)): its own slot arrays, CHAIN2_OFF arrays above the first chain's and taken in the same
    // order, so that each chain's scales depend on its own frames only (deterministic) and one elementwise max after the join
    // gives the backward pass the ranges of the whole tensors
    std::map<const void*, const float*> of2;
    int next2 = 0;
    bool chain2 = false;
    bool on = false;
};
struct TrainCtx {
    float* tail_slab = nullptr;                  // K-split slab of the running forward / backward pass (a region of the caller's workspace)
    RangeCtx rng;
    std::unordered_map<const float*, const float*> cells;      // weight panel -> the same panel pre-split into fp16 cells
    std::unordered_map<const float*, const void*> cells1;      // ... -> its high-only H1 cells (16-bit tier, launch_pack_h1)
    const void* defer_plan = nullptr;            // TPlan of the running backward pass when weight-gradient finalisation is deferred
    char* defer_ws = nullptr;
    // Weight gradients on a second stream (EXPERIMENTS.md section 6a (8)): the data-gradient chain is the critical path of the backward pass
    // and its 11-frame grids leave CUs idle; a layer's weight gradient only needs (x, dY) and is not needed before the finalisation
    // launches, so it runs on `s2` behind an event and the chain goes on.  readers: events recorded on s2 behind the launches that
    // READ a gradient buffer -- the chain waits for them before it overwrites that buffer.
    hipStream_t s2 = nullptr;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_next = 0;
    std::multimap<const void*, hipEvent_t> readers;
    bool overlap = false;                        // set for the duration of a backward pass
    // fp16 high / low copies of retained activations and gradient buffers (ConvArgs::shadow), operands of wgrad_dma:
    // tensor base -> copy base (from the plan, per pass); tensor base -> previous-step range slots of the launch that WROTE the copy
    // in this pass (absent: no copy of the current contents exists)
    std::unordered_map<const void*, float*> shadow_base;
    std::unordered_map<const void*, const float*> shadow_prev;
    std::vector<int> h2_slots;                   // range slots of the tensors written as H2 with predicted scales in this pass
    ~TrainCtx() {
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        if (s2) (void)hipStreamDestroy(s2);
    }
    hipEvent_t take_event() {
        if (ev_next == ev_pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            ev_pool.push_back(e);
        }
        return ev_pool[ev_next++];
    }
};

struct dgp_trainer {
    TrainCtx ctx;
    dgp_net* net = nullptr;
    std::vector<TLayer> tl;
    std::vector<std::string> names;
    std::vector<long long> offs, sizes;
    std::vector<int> is_stat;
    long long n_train = 0, n_stat = 0;
    float *params = nullptr, *grads = nullptr, *mom = nullptr, *stats = nullptr;
    double* d_sumsq = nullptr;        // TWO accumulators used in turn: the optimiser's kernel zeroes the other one for the next step (no fill launch)
    int sumsq_k = 0;
    float* d_gnorm = nullptr;
    float* h_status = nullptr;        // pinned, device-visible: dgp_trainer_step_status' kernel writes [8 losses | gnorm | flag] straight into it
    float* d_rng_pool = nullptr;      // activation / gradient range slots (RANGE_POOL arrays), zeroed per pass
    float* d_rng_prev = nullptr;      // ... as the previous step left them (copied before the zeroing)
    int* d_fast_flag = nullptr;       // != 0: an H2 tensor of the last fast pass left its predicted range (the step must be repeated)
    bool fast_next = false;           // dgp_trainer_fast_mode: the next forward pass keeps blocks 2-4 as H2 tensors with predicted scales
    bool fwd_fast = false;            // what the last forward pass did (the backward pass reads its tensors accordingly)
    // precision tier (dgp_trainer_set_tier): 0 = parity (fp32-class arithmetic, fp32 retained activations); 1 = 16-bit: once a pass of
    // the same shapes has left its ranges behind, blocks 2-4 keep their retained activations AND their gradient tensors as H1 cells
    // (2 bytes per channel, scales predicted from the previous step's ranges, no fp32 twins), forward / data-gradient convs run one MFMA
    // per product on H1 operands and the weight gradients read both operands in place by LDS-DMA (wgrad_dma_h1).  fp32 master
    // weights, momentum, gradient accumulation (float atomics into fp32 dW), loss, block1 / stem / heads.  A step whose tensors left
    // their predicted ranges raises d_fast_flag and the host repeats it on the parity path.
    int tier = 0;
    int fwd_fmt = 0;                  // cell format of the last fast forward pass: 1 H2 (tuning builds' fast pass), 2 H1 (tier 1)
    void* d_h1_table = nullptr;       // PackH3Desc of every panel whose H1 cells are rebuilt per sync (tier 1)
    float* d_stem_rows = nullptr;     // tier 1: the stem's row panel [7 x 8 pixels][64][4] and its cells for stem_pool_fused_kernel, rebuilt per sync
    void* d_stem_cells = nullptr;
    bool fwd_stem_fused = false;      // the last forward pass ran the fused root block (no conv1 map, no fp32 pool output)
    // tier 1: both heads' data-gradient panels merged (pack_heads_dgrad_kernel), its range slots and its H1 cells; CT input channels
    float* d_hmT = nullptr;
    float* d_hm_rng = nullptr;
    void* d_hmT_h1 = nullptr;
    int hm_ct = 0, hm_nk = 0;
    bool fwd_feat32 = true;           // the last fast forward pass left an fp32 copy of the features (not when the heads' backward reads H1)
    // tier 1: what only a parity pass reads -- the H2 cells of every panel, the heads' 2x2-conv and data-gradient panels -- is not rebuilt by
    // every dgp_trainer_sync_weights but when a plain pass is about to run (refresh_parity_panels)
    bool parity_stale = false;
    int n_h1 = 0;
    float* d_wrng = nullptr;          // weight-panel range slots: [2 * n_layers] (forward panels, data-gradient panels), per sync
    void* d_pack_table = nullptr;     // PackDesc of every non-head layer (pack_all_kernel), built at the first sync
    int n_pack = 0;
    void* d_h3_table = nullptr;       // PackH3Desc of every panel whose cells are rebuilt per sync (DGP_TRAIN_CELLS)
    int n_h3 = 0;
    void* d_fin_table = nullptr;      // FinDesc of every non-head layer (deferred weight-gradient finalisation), per batch size
    int n_fin = 0, fin_B = -1, fin_h = -1, fin_w = -1;
    // Gradient groups (data-parallel training: the all-reduce of a group starts while the backward pass is still running).  The flat
    // gradient buffer is laid out in layer order and the pass finishes the layers back to front, so a group is a run of bottleneck
    // units: group 0 = the heads + the last units, ..., the LAST group = whatever is left + the stem.  Per group: the unit after whose
    // backward it is complete (-1: the end of the pass), its rows of the finalisation table, its float range in the flat buffer and
    // the event recorded behind its finalisation.
    struct GradGroup { int cut_ui = -1, fin_first = 0, fin_count = 0; long long lo = 0, hi = 0; hipEvent_t ev = nullptr; };
    std::vector<GradGroup> groups;
    std::vector<int> fin_of_layer;
    ~dgp_trainer() {
        for (void* q : {(void*)d_rng_pool, (void*)d_rng_prev, (void*)d_fast_flag, (void*)d_wrng, d_pack_table, d_fin_table, d_h3_table, d_h1_table, (void*)d_stem_rows, d_stem_cells, (void*)d_hmT, (void*)d_hm_rng, d_hmT_h1}) if (q) (void)hipFree(q);
        for (auto& g : groups) if (g.ev) (void)hipEventDestroy(g.ev);
        for (auto& t : tl) { if (t.d_wT) (void)hipFree(t.d_wT); if (t.d_wTh3) (void)hipFree(t.d_wTh3); if (t.d_wTh1) (void)hipFree(t.d_wTh1); }
        for (void* p : {(void*)params, (void*)grads, (void*)mom, (void*)stats, (void*)d_sumsq, (void*)d_gnorm})
            if (p) (void)hipFree(p);
        if (h_status) (void)hipHostFree(h_status);
    }
};

namespace {

int next_pow2(int x) { int p = 4; while (p < x) p <<= 1; return p; }

struct TPlan {
    // retained activations
    size_t p0, c1, pool, pidx;
    std::vector<size_t> sc, r1, r2, xo;         // per unit
    size_t scmap, locref;
    // gradients
    size_t g0, g1, dxa, dr1, dr2, dxa_b, dr1_b, dr2_b, dc1, dph0, dph1, dwraw, colsum, tail;      // (_b: second copies, units alternate)
    // per-layer weight-gradient scratch (non-head layers): [colsum | dot][dWraw], one contiguous region zeroed once per backward pass
    std::vector<size_t> cs_l, dw_l;
    size_t dwall = 0, dwall_bytes = 0;
    // fp16 high / low copies (ConvArgs::shadow) of the tensors the 128 x 128 weight-gradient tiles read; 0: none
    std::vector<size_t> sh_r1, sh_r2, sh_xo;
    size_t sh_g0 = 0, sh_g1 = 0, sh_dr1 = 0, sh_dr2 = 0, sh_dr1_b = 0, sh_dr2_b = 0;
    size_t feat32 = 0;           // fast pass: fp32 copy of the block4 features for the heads' backward
    size_t dphh = 0;             // 16-bit tier: both heads' gathered loss gradients as one H1 tensor [B*h*w, heads_ct(nj)]
    size_t sh_pool = 0;          // 16-bit tier: H1 copy of the pool output (input of the first unit)
    size_t total;
};

// A/B switch (DGP_WGRAD_DMA=0: no copies, wgrad_h3p as before)
static const bool g_wgrad_dma = (dgp_tune("DGP_WGRAD_DMA", 1) != 0);

size_t al(size_t x) { return (x + 255) / 256 * 256; }
// channels of the merged heads tensor: both heads' padded phase columns, rounded up to the H1 kernels' K-step
int heads_ct(int nj) { return (next_pow2(4 * nj) + next_pow2(8 * nj) + 63) / 64 * 64; }

TPlan make_tplan(const dgp_trainer* tr, int B) {
    const dgp_net* net = tr->net;
    const dgp_net_desc& d = net->desc;
    TPlan p;
    size_t o = 0;
    auto take = [&](size_t nfl) { size_t r = o; o += al(nfl * sizeof(float)); return r; };
    p.p0 = take((size_t)B * d.in_h * d.in_w * 4);
    p.c1 = take((size_t)B * net->h1 * net->w1 * 64);
    p.pool = take((size_t)B * net->hp * net->wp * 64);
    p.pidx = take((size_t)B * net->hp * net->wp * 64 / 4);      // uchar4 per window and 4 channels: first-maximum positions of the pool
    int h = net->hp, w = net->wp;
    size_t xmax = (size_t)B * h * w * 64, r1max = 0, r2max = 0;
    for (const Unit& u : net->units) {
        const int ho = (h + u.stride - 1) / u.stride, wo = (w + u.stride - 1) / u.stride;
        p.sc.push_back(u.sc >= 0 ? take((size_t)B * ho * wo * u.depth) : 0);
        p.r1.push_back(take((size_t)B * h * w * u.depth_bn));
        p.r2.push_back(take((size_t)B * ho * wo * u.depth_bn));
        p.xo.push_back(take((size_t)B * ho * wo * u.depth));
        xmax = std::max(xmax, (size_t)B * h * w * u.depth_in);
        xmax = std::max(xmax, (size_t)B * ho * wo * u.depth);
        r1max = std::max(r1max, (size_t)B * h * w * u.depth_bn);
        r2max = std::max(r2max, (size_t)B * ho * wo * u.depth_bn);
        h = ho; w = wo;
    }
    const int nj = d.num_joints;
    p.scmap = take((size_t)B * 4 * h * w * nj);
    p.locref = take((size_t)B * 4 * h * w * 2 * nj);
    p.g0 = take(xmax); p.g1 = take(xmax); p.dxa = take(xmax);
    p.dr1 = take(r1max); p.dr2 = take(r2max);
    p.dxa_b = take(xmax); p.dr1_b = take(r1max); p.dr2_b = take(r2max);
    p.dc1 = take((size_t)B * net->h1 * net->w1 * 64);
    p.dph0 = take((size_t)B * h * w * next_pow2(4 * nj));
    p.dph1 = take((size_t)B * h * w * next_pow2(8 * nj));
    p.dphh = take((size_t)B * h * w * heads_ct(nj) / 2);
    size_t wmax = 0;
    for (size_t li = 0; li < net->layers.size(); ++li) {
        const ConvLayer& l = net->layers[li];
        const int cdy = tr->tl[li].cpad ? tr->tl[li].cpad : l.Cout;
        wmax = std::max(wmax, (size_t)l.KH * l.KW * l.Cin * cdy);
    }
    p.colsum = take(2 * 4096);      // directly in front of dwraw: wgrad_launch zeroes both with one memset
    p.dwraw = take(wmax);
    p.tail = take(TAIL_SLAB_FLOATS);
    p.dwall = o;
    p.cs_l.assign(net->layers.size(), 0); p.dw_l.assign(net->layers.size(), 0);
    for (size_t li = 0; li < net->layers.size(); ++li) {
        if ((int)li == net->head_part || (int)li == net->head_locref) continue;
        const ConvLayer& l = net->layers[li];
        p.cs_l[li] = take((size_t)2 * l.Cout);
        p.dw_l[li] = take((size_t)l.KH * l.KW * l.Cin * l.Cout);
    }
    p.dwall_bytes = o - p.dwall;
    p.sh_r1.assign(net->units.size(), 0); p.sh_r2.assign(net->units.size(), 0); p.sh_xo.assign(net->units.size(), 0);
    if (g_wgrad_dma) {
        int hh = net->hp, ww = net->wp;
        for (size_t ui = 0; ui < net->units.size(); ++ui) {
            const Unit& u = net->units[ui];
            const int ho = (hh + u.stride - 1) / u.stride, wo = (ww + u.stride - 1) / u.stride;
            // a copy pays where a 128 x 128 tile reads it: r1 feeds conv2 (K = 9 C1, Cdy = C1), r2 feeds conv3, xo the next unit's conv1 / shortcut
            if (u.depth_bn >= 128) p.sh_r1[ui] = take((size_t)B * hh * ww * u.depth_bn);
            if (u.depth_bn >= 128) p.sh_r2[ui] = take((size_t)B * ho * wo * u.depth_bn);
            if (ui + 1 < net->units.size() && net->units[ui + 1].depth_bn >= 128) p.sh_xo[ui] = take((size_t)B * ho * wo * u.depth);
            hh = ho; ww = wo;
        }
        p.sh_g0 = take(xmax); p.sh_g1 = take(xmax);
        p.sh_dr1 = take(r1max); p.sh_dr2 = take(r2max); p.sh_dr1_b = take(r1max); p.sh_dr2_b = take(r2max);
        p.feat32 = take((size_t)B * h * w * net->units.back().depth);
        p.sh_pool = take((size_t)B * net->hp * net->wp * 64 / 2);
    }
    p.total = o;
    return p;
}

// K-split slab of the running forward / backward pass (a region of the caller's workspace)
static thread_local TrainCtx* g_ctx = nullptr;      // the trainer whose entry point is running on this thread

// Operand ranges for the fp16-split conv kernels inside the training step.  Every conv launched through conv_launch takes a
// fresh slot array from a pool for max |out| and records it under its output pointer; a later conv whose input pointer (and
// weight panel) has a recorded range runs the fp16 kernels, anything else (tensors written by other kernels: pooling, loss
// gradients, head gathers) falls back to the range-free bf16 split.  The pool is zeroed at the start of each pass.
constexpr int RANGE_POOL = 768;             // forward chain 1 [0, 256) | forward chain 2 [256, 512) | backward [512, 768)
constexpr int RANGE_FWD = 256;

__global__ __launch_bounds__(256) void range_merge_kernel(float* __restrict__ a, const float* __restrict__ b, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float x = a[i], y = b[i]; a[i] = (y > x || y != y) ? y : x; }      // non-negative maxima; a NaN stays
}

// start of a pass: prev <- pool over [n_copy) floats (what the same kind of pass measured one step ago), pool <- 0 over [n_zero) floats,
// optionally *flag <- 0 -- one launch instead of a copy and two fills (blit launches cost ~15 us of queue gaps each at the step's boundary)
__global__ __launch_bounds__(256) void range_roll_kernel(float* __restrict__ pool, float* __restrict__ prev, int n_copy, int n_zero, int* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_copy) prev[i] = pool[i];
    if (i < n_zero) pool[i] = 0.f;
    if (i == 0 && flag) *flag = 0;
}

static float* range_take() {
    if (!g_ctx || !g_ctx->rng.on || !g_ctx->rng.pool) return nullptr;
    if (g_ctx->rng.chain2) {
        if (g_ctx->rng.next2 >= RANGE_FWD) return nullptr;
        return g_ctx->rng.pool + (size_t)(RANGE_FWD + g_ctx->rng.next2++) * ABSMAX_SLOTS;
    }
    if (g_ctx->rng.next >= g_ctx->rng.limit) return nullptr;
    return g_ctx->rng.pool + (size_t)(g_ctx->rng.next++) * ABSMAX_SLOTS;
}
static const float* range_of(const void* p) {
    if (!g_ctx || !g_ctx->rng.on) return nullptr;
    if (g_ctx->rng.chain2) {                     // the chain's own tensors first; weight panels are shared
        auto it2 = g_ctx->rng.of2.find(p);
        if (it2 != g_ctx->rng.of2.end()) return it2->second;
    }
    auto it = g_ctx->rng.of.find(p);
    return it == g_ctx->rng.of.end() ? nullptr : it->second;
}
static void range_set(const void* p, const float* slot) {
    auto& m = g_ctx->rng.chain2 ? g_ctx->rng.of2 : g_ctx->rng.of;
    if (slot) m[p] = slot; else m.erase(p);
}

// Start of a forward or a backward pass.  Forward: everything fresh (first half of the pool).  Backward: the forward tensors'
// ranges stay (weight gradients read the retained activations), gradient tensors take slots from the second half.
static void range_pass_begin(dgp_trainer* tr, hipStream_t s, bool backward, int* flag_to_clear = nullptr) {
    static const bool enabled = (dgp_tune("DGP_TRAIN_F16", 1) != 0) &&
                                !(getenv("DGP_CONV_MODE") && strcmp(getenv("DGP_CONV_MODE"), "f16x3") != 0);
    g_ctx->rng.on = enabled && tr->d_rng_pool && tr->d_wrng;
    g_ctx->rng.pool = tr->d_rng_pool;
    const size_t fwd_bytes = (size_t)(2 * RANGE_FWD) * ABSMAX_SLOTS * sizeof(float);
    const size_t bwd_bytes = (size_t)(RANGE_POOL - 2 * RANGE_FWD) * ABSMAX_SLOTS * sizeof(float);
    g_ctx->rng.chain2 = false;
    if (!backward) { g_ctx->rng.of.clear(); g_ctx->rng.of2.clear(); g_ctx->rng.next = 0; g_ctx->rng.next2 = 0; g_ctx->rng.limit = RANGE_FWD; }
    else { g_ctx->rng.next = 2 * RANGE_FWD; g_ctx->rng.limit = RANGE_POOL; }
    g_ctx->rng.prev = tr->d_rng_prev;
    if (!backward) g_ctx->shadow_prev.clear();
    if (!g_ctx->rng.on) return;
    // what this kind of pass measured one step ago predicts the scales of this pass's fp16 copies (chain 2's own slots are not needed:
    // after the merge chain 1's hold the maxima of the whole tensors)
    {
        const size_t off = (backward ? fwd_bytes : 0) / sizeof(float);
        const int n_copy = (g_wgrad_dma && tr->d_rng_prev) ? (int)((backward ? bwd_bytes : fwd_bytes / 2) / sizeof(float)) : 0;
        const int n_zero = (int)((backward ? bwd_bytes : fwd_bytes) / sizeof(float));
        hipLaunchKernelGGL(range_roll_kernel, dim3((n_zero + 255) / 256), dim3(256), 0, s, tr->d_rng_pool + off,
                           tr->d_rng_prev ? tr->d_rng_prev + off : nullptr, n_copy, n_zero, flag_to_clear);
    }
    if (backward) return;
    const size_t nl = tr->net->layers.size();
    for (size_t li = 0; li < nl; ++li) {
        if (tr->net->layers[li].d_w) g_ctx->rng.of[tr->net->layers[li].d_w] = tr->d_wrng + li * ABSMAX_SLOTS;
        if (tr->tl[li].d_wT) g_ctx->rng.of[tr->tl[li].d_wT] = tr->d_wrng + (nl + li) * ABSMAX_SLOTS;
    }
    if (tr->d_hmT && tr->d_hm_rng) g_ctx->rng.of[tr->d_hmT] = tr->d_hm_rng;
}


// previous-step slots of the tensor that takes `slot` in this pass (the passes take their slots in the same order every step)
static const float* range_prev_of(const float* slot) {
    if (!slot || !g_ctx || !g_ctx->rng.pool || !g_ctx->rng.prev) return nullptr;
    long long idx = (slot - g_ctx->rng.pool) / ABSMAX_SLOTS;
    if (idx < 0 || idx >= RANGE_POOL) return nullptr;
    if (idx >= RANGE_FWD && idx < 2 * RANGE_FWD) idx -= RANGE_FWD;      // chain 2 writes its frames of the same copy with the same scale
    return g_ctx->rng.prev + idx * ABSMAX_SLOTS;
}
static thread_local bool g_shadow_want = true;      // backward pass: only gradients that a 128 x 128 weight-gradient tile will read
// fast pass: formats of the NEXT conv_launch (consumed by it).  An H2 tensor's scale is the one predicted from the previous step's slots
// of the tensor named by the launch's in_key / out_key / res_key.
struct H2Launch { int in_fmt = 0, out_fmt = 0, res_fmt = 0, mask_fmt = 0; const void* res_key = nullptr; };
static thread_local H2Launch g_h2;
static thread_local int g_shadow_fmt = 1;            // format of the fp16 copies written in this pass: 1 H2 (high / low), 2 H1 (tier 1: high only)

// weight panel -> the same panel pre-split into fp16 cells (filled by dgp_trainer_sync_weights): with the cells and both ranges the
// conv runs on the compute-side-split / LDS-DMA kernels of the inference engine
// Default on (DGP_TRAIN_CELLS=0: the trainer's convs split their weights in the loaders): the cells of all panels are rebuilt by ONE
// launch per sync (pack_h3_all_kernel), after which forward and data-gradient convs run the engine's compute-side-split / LDS-DMA /
// 16x16x32 kernels: 17.0 -> 16.2 ms per step.  (With one pack launch per layer and panel the packing cost what the kernels saved.)
static const bool g_train_cells = (dgp_tune("DGP_TRAIN_CELLS", 1) != 0);

hipError_t conv_launch(const ConvLayer& l, const float* wpk, int nk, int coutP, const float* in, int N, int H, int W,
                       int Cin, int pad_t, int pad_l, int Ho, int Wo, int Cout, int stride, int up, const float* scale,
                       const float* bias, const float* res, int res_s, int res_H, int res_W, const float* mask,
                       bool relu, int out_mode, int dc_nj, float* out, hipStream_t s, const void* in_key = nullptr,
                       const void* out_key = nullptr) {
    // in_key / out_key: the tensors under whose names this launch looks up / registers range slots when `in` / `out` are frame ranges
    // inside them (the forward pass as two chains of frames; each chain keeps its own slots, merged after the join)
    ConvArgs a{};
    a.in = in; a.wpk = wpk; a.scale = scale; a.bias = bias; a.res = res; a.mask = mask; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.log2cin4 = ilog2(Cin / 4);
    a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.CoutP = coutP;
    a.KH = l.KH; a.KW = l.KW; a.stride = stride; a.dil = l.rate; a.pad_t = pad_t; a.pad_l = pad_l;
    a.ntaps = l.KH * l.KW; a.nk = nk; a.M = N * Ho * Wo;
    a.res_s = res ? res_s : 0; a.res_H = res_H; a.res_W = res_W; a.up = up;
    a.relu = relu ? 1 : 0; a.out_mode = out_mode; a.dc_nj = dc_nj;
    if ((double)N * H * W * Cin * 4 > 4294967000.0 || (double)a.M * Cout * 4 > 4294967000.0 ||
        (res && (double)N * res_H * res_W * Cout * 4 > 4294967000.0))
        return hipErrorInvalidValue;              // 32-bit buffer descriptors: every tensor stays below 4 GiB (lower the batch)
    a.in_bytes = (unsigned)((size_t)N * H * W * Cin * 4);
    a.out_bytes = (unsigned)((size_t)a.M * Cout * 4);
    a.res_bytes = res ? (unsigned)((size_t)N * res_H * res_W * Cout * 4) : 0u;
    a.w_bytes = (unsigned)((size_t)nk * 8 * coutP * 16);
    a.slab = g_ctx->tail_slab; a.slab_bytes = g_ctx->tail_slab ? (unsigned)(TAIL_SLAB_FLOATS * sizeof(float)) : 0u;
    const H2Launch h2 = g_h2;
    g_h2 = H2Launch();
    const float* rin = range_of(in_key ? in_key : in);
    const float* rw = range_of(wpk);
    a.mask_fmt = h2.mask_fmt;
    if (h2.in_fmt || h2.out_fmt || h2.res_fmt) {
        if (!(rin && rw && out_mode == 0 && h2.in_fmt && h2.out_fmt == h2.in_fmt && (!h2.res_fmt || h2.res_fmt == h2.in_fmt))) return hipErrorInvalidValue;
        a.in_fmt = h2.in_fmt; a.out_fmt = h2.out_fmt; a.res_fmt = h2.res_fmt;
        a.in_scale_dev = range_prev_of(rin);
        if (h2.res_fmt) a.res_scale_dev = range_prev_of(range_of(h2.res_key));
        if (!a.in_scale_dev || (h2.res_fmt && !a.res_scale_dev)) return hipErrorInvalidValue;
    }
    if (rin && rw && out_mode == 0) {
        a.in_absmax = rin; a.w_absmax = rw;
        const auto c = g_ctx->cells.find(wpk);
        if (g_train_cells && c != g_ctx->cells.end()) { a.wh3 = c->second; a.wh3_bytes = a.w_bytes; }
        if (a.in_fmt == 2) {                       // H1 operands: the panel's high-only cells (none: launch_conv refuses)
            const auto c1 = g_ctx->cells1.find(wpk);
            a.wh3 = c1 != g_ctx->cells1.end() ? c1->second : nullptr;
            a.wh3_bytes = a.w_bytes;               // (fp32-panel bytes: launch_conv halves both for H1 cells)
        }
    }
    if (out_mode == 0) {
        a.out_absmax = range_take();
        range_set(out_key ? out_key : out, a.out_absmax);
    }
    const int tile = pick_tile(a.M, coutP, nk * BK, a.in_absmax && a.w_absmax);
    if (a.out_fmt) {
        // the output IS the fp16 high / low tensor: later weight-gradient launches read it in place
        const void* key = out_key ? out_key : (const void*)out;
        a.out_scale_dev = range_prev_of(a.out_absmax);
        if (!a.out_scale_dev || !a.wh3) return hipErrorInvalidValue;
        g_ctx->shadow_base[key] = const_cast<float*>(static_cast<const float*>(key));
        g_ctx->shadow_prev[key] = a.out_scale_dev;
        const int idx = (int)((a.out_scale_dev - g_ctx->rng.prev) / ABSMAX_SLOTS);
        if (std::find(g_ctx->h2_slots.begin(), g_ctx->h2_slots.end(), idx) == g_ctx->h2_slots.end()) g_ctx->h2_slots.push_back(idx);
    } else if (g_ctx && out_mode == 0) {
        // fp16 copy of the output for wgrad_dma: kernels that end in ls_epilogue / tail_fixup_kernel write it
        const void* key = out_key ? out_key : (const void*)out;
        const auto sb = g_ctx->shadow_base.find(key);
        const float* prev = range_prev_of(a.out_absmax);
        if (sb != g_ctx->shadow_base.end() && g_shadow_want && prev && (tile == TILE_128x128_H3K32 || tile == TILE_128x64_H3) &&
            a.in_absmax && a.w_absmax && Cin >= 32 && Cout % 8 == 0) {
            a.shadow_fmt = g_shadow_fmt;
            a.shadow = g_shadow_fmt == 2 ? reinterpret_cast<float*>(reinterpret_cast<char*>(sb->second) + (out - static_cast<const float*>(key)) * 2)
                                         : sb->second + (out - static_cast<const float*>(key));
            a.shadow_prev = prev;
            g_ctx->shadow_prev[key] = prev;
        } else {
            g_ctx->shadow_prev.erase(key);
        }
    }
    return launch_conv(a, tile, s);
}

hipError_t wgrad_launch(const float* x, int N, int H, int W, int Cin, const float* dy, int Ho, int Wo, int Cdy, int KH,
                        int KW, int stride, int dil, int pad_t, int pad_l, float* dwraw, float* colsum, hipStream_t s,
                        bool zeroed = false, const float* rx_given = nullptr, const float* rdy_given = nullptr,
                        const void* xs = nullptr, const float* x_prev = nullptr, const void* dys = nullptr, const float* dy_prev = nullptr,
                        int* x_h2_only_flag = nullptr, bool h1 = false) {
    // x_h2_only_flag: x exists only as H2 cells (xs; fast pass) -- the LDS-DMA tile is the only kernel that can read it, and copies that
    // left their predicted range raise the flag instead of falling back
    WgradArgs a{};
    a.x = x; a.dy = dy; a.dw = dwraw; a.colsum = colsum; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.log2cin4 = ilog2(Cin / 4);
    a.Ho = Ho; a.Wo = Wo; a.Cdy = Cdy; a.KW = KW; a.stride = stride; a.dil = dil; a.pad_t = pad_t; a.pad_l = pad_l;
    a.ntaps = KH * KW; a.kchunks = KH * KW * (Cin / 4); a.M = N * Ho * Wo;
    if ((double)N * H * W * Cin * 4 > 4294967000.0 || (double)a.M * Cdy * 4 > 4294967000.0) return hipErrorInvalidValue;
    a.x_bytes = (unsigned)((size_t)N * H * W * Cin * 4);
    a.dy_bytes = (unsigned)((size_t)a.M * Cdy * 4);
    hipError_t e = hipSuccess;
    if (zeroed) {
        // per-layer scratch, zeroed once for the whole pass (dgp_train_backward)
    } else if (colsum && colsum + 2 * 4096 == dwraw) {      // the plan's layout: [colsum | dot][dwraw] -> one fill
        e = hipMemsetAsync(colsum, 0, ((size_t)2 * 4096 + (size_t)a.kchunks * 4 * Cdy) * sizeof(float), s);
        if (e != hipSuccess) return e;
    } else {
        e = hipMemsetAsync(dwraw, 0, (size_t)a.kchunks * 4 * Cdy * sizeof(float), s);
        if (e != hipSuccess) return e;
        if (colsum) {       // [0, Cdy): column sums, [Cdy, 2 Cdy): dot products of finalize (zeroed together)
            e = hipMemsetAsync(colsum, 0, (size_t)2 * Cdy * sizeof(float), s);
            if (e != hipSuccess) return e;
        }
    }
    const bool big = h1 || (a.kchunks * 4 >= 128 && Cdy >= 128);       // (H1 operands: the LDS-DMA tile, partly filled for the 64-channel layers)
    const int BR = big ? 128 : 64;
    const int kt = (a.kchunks * 4 + BR - 1) / BR, nt = (Cdy + BR - 1) / BR;
    // workgroups per launch: ONE round of resident workgroups (2 per CU).  Every workgroup adds its whole 128 x 128 tile to dW with
    // float atomics, so more pixel slices mean more atomic traffic, fewer leave CUs idle: 1536 -> 18.4 ms per step, 1024 -> 17.8,
    // 640 -> 18.2, 512 -> 16.7, 384 -> 17.2, 256 -> 18.1 (DGP_WGRAD_WGS)
    static const int wgs_target = dgp_tune("DGP_WGRAD_WGS", 512);
    static const int wgs_small = dgp_tune("DGP_WGRAD_WGS_SMALL", 1024);      // 64 x 64 tiles: 1024 (16.7 vs 16.9 ms at 512)
    // (wgrad_dma_h1: its loop is shorter, so the 64-KB-per-workgroup atomic epilogue weighs more: 512 -> 7.99, 384 -> 7.80, 256 -> 8.02 ms per step)
    static const int wgs_h1 = dgp_tune("DGP_WGRAD_WGS_H1", 384);
    int split = std::max(1, (h1 ? wgs_h1 : big ? wgs_target : wgs_small) / (kt * nt));
    int mpb = ((a.M + split - 1) / split + 63) / 64 * 64;      // (64: wgrad_dma walks 16-pixel steps unrolled by four)
    if (mpb < 256) mpb = 256;
    split = (a.M + mpb - 1) / mpb;
    a.m_per_block = mpb;
    static bool attr_dev[16][4] = {};
    auto& attr = attr_dev[dgp_device_slot()];
    static const bool h3_env = (dgp_tune("DGP_WGRAD_F16", 1) != 0);       // A/B switch
    const float* rx = rx_given ? rx_given : range_of(x);
    const float* rdy = rdy_given ? rdy_given : range_of(dy);
    // both operands also exist as fp16 high / low copies written by their producers: LDS-DMA tile (falls back per workgroup to the
    // fp32-MFMA tile when this step's ranges left the copies' predicted scales)
    // (guards: the kernel's offset walkers run up to 15 (row0) + 63 (sub-steps rounded up to four) + 64 (four prefetched steps) = 142
    //  pixel rows past the tensor's end, and its channel masks need Cin / 8 to be a power of two)
    if (h1) {       // 16-bit tier: both operands are H1 tensors (xs, dys: in place or the H1 copy of a boundary tensor); no other kernel reads them
        if (!(big && g_wgrad_dma && rx && rdy && xs && dys && x_prev && dy_prev && x_h2_only_flag && Cin % 16 == 0 && ((Cin / 8) & (Cin / 8 - 1)) == 0 &&
              Cdy % 8 == 0 && (double)a.x_bytes + 160.0 * Cin * 4 < 4294967000.0 && (double)a.dy_bytes + 160.0 * Cdy * 4 < 4294967000.0))
            return hipErrorInvalidValue;
        a.xs = xs; a.dys = dys; a.x_prev = x_prev; a.x_cur = rx; a.dy_prev = dy_prev; a.dy_cur = rdy;
        a.x = nullptr; a.dy = nullptr; a.fail_flag = x_h2_only_flag;
        hipLaunchKernelGGL(wgrad_dma_h1, dim3(kt, nt, split), dim3(256), 32 * 1024, s, a);
        return hipGetLastError();
    }
    if (big && h3_env && g_wgrad_dma && rx && rdy && xs && dys && x_prev && dy_prev && Cin % 16 == 0 && ((Cin / 8) & (Cin / 8 - 1)) == 0 && Cdy % 8 == 0 &&
        (double)a.x_bytes + 160.0 * Cin * 4 < 4294967000.0 && (double)a.dy_bytes + 160.0 * Cdy * 4 < 4294967000.0) {
        if (!attr[3]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_dma), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            if (e != hipSuccess) return e;
            attr[3] = true;
        }
        a.xs = xs; a.dys = dys; a.x_prev = x_prev; a.x_cur = rx; a.dy_prev = dy_prev; a.dy_cur = rdy;
        if (x_h2_only_flag) { a.x = nullptr; a.fail_flag = x_h2_only_flag; }
        hipLaunchKernelGGL(wgrad_dma, dim3(kt, nt, split), dim3(256), 64 * 1024, s, a);
        return hipGetLastError();
    }
    if (x_h2_only_flag) return hipErrorInvalidValue;       // (no other kernel reads H2 cells)
    if (big && h3_env && rx && rdy && Cin % 4 == 0) {      // both operand ranges known: 16-bit matrix pipe
        if (!attr[2]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_h3), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            if (e != hipSuccess) return e;
            attr[2] = true;
        }
        WgradRanges rg{rx, rdy};
        static const bool pipe_env = (dgp_tune("DGP_WGRAD_PIPE", 1) != 0);      // A/B switch
        if (pipe_env) {
            if (!attr[0]) {
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_h3p), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                if (e != hipSuccess) return e;
                attr[0] = true;
            }
            hipLaunchKernelGGL(wgrad_h3p, dim3(kt, nt, split), dim3(256), 2 * 4 * 32 * 256, s, a, rg);
            return hipGetLastError();
        }
#ifdef DGP_DIAG
        unsigned long long hz[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_diag), hz, sizeof(hz));
#endif
        hipLaunchKernelGGL(wgrad_h3, dim3(kt, nt, split), dim3(256), 2 * 4 * 32 * 256, s, a, rg);
#ifdef DGP_DIAG
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(hz, HIP_SYMBOL(g_wgrad_diag), sizeof(hz));
        const double ns = (double)hz[7], nw = (double)hz[10];
        printf("[diag wgrad_h3] K %d Cdy %d M %d grid %dx%dx%d steps/wg %.1f | per step (wave 0, 100 MHz ticks): gload-issue %.1f tr-read0 %.1f "
               "mfma0 %.1f tr-read1 %.1f mfma1 %.1f split+store %.1f barrier %.1f | per wg: loop %.0f epilogue %.0f\n", a.kchunks * 4, Cdy, a.M,
               kt, nt, split, ns / nw, hz[0] / ns, hz[1] / ns, hz[2] / ns, hz[3] / ns, hz[4] / ns, hz[5] / ns, hz[6] / ns, hz[8] / nw, hz[9] / nw);
#endif
        return hipGetLastError();
    }
    if (big) {
        if (!attr[1]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f32<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            if (e != hipSuccess) return e;
            attr[1] = true;
        }
        hipLaunchKernelGGL(wgrad_f32<2>, dim3(kt, nt, split), dim3(256), 2 * 2 * 32 * 128 * 4, s, a);
    } else {
        hipLaunchKernelGGL(wgrad_f32<1>, dim3(kt, nt, split), dim3(256), 2 * 2 * 32 * 64 * 4, s, a);
    }
    return hipGetLastError();
}

#define TRY_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return fail(DGP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

}  // namespace

static int trainer_owner_sync(void* owner, void* stream);

extern "C" {

int dgp_trainer_create(dgp_net* net, dgp_trainer** out) {
    if (!net || !out) return fail(DGP_ERR_INVALID, "dgp_trainer_create: null argument");
    if (net->head_locref < 0) return fail(DGP_ERR_INVALID, "dgp_trainer_create: the net must be built with_locref (dgp_loss trains both heads)");
    dgp_trainer* tr = new dgp_trainer();
    tr->net = net;
    net->owner = tr; net->owner_sync = trainer_owner_sync;
    tr->tl.resize(net->layers.size());
    auto add = [&](const std::string& name, long long size, bool stat) {
        long long& ctr = stat ? tr->n_stat : tr->n_train;
        const long long off = ctr;
        ctr += (size + 3) / 4 * 4;                 // keep every tensor 16-byte aligned
        tr->names.push_back(name); tr->offs.push_back(off); tr->sizes.push_back(size); tr->is_stat.push_back(stat ? 1 : 0);
        return off;
    };
    for (size_t li = 0; li < net->layers.size(); ++li) {
        const ConvLayer& l = net->layers[li];
        TLayer& t = tr->tl[li];
        const bool head = ((int)li == net->head_part || (int)li == net->head_locref);
        if (head) {
            const int njt = l.Cout / 4;
            t.cin_real = l.Cin;
            t.w_off = add(l.scope + "/weights", 9ll * njt * l.Cin, false);
            t.b_off = add(l.scope + "/biases", njt, false);
            t.cpad = next_pow2(l.Cout);
            t.cinP = coutp_for(l.Cin);
            t.nkT = nk_for(2, 2, t.cpad);
        } else {
            t.cin_real = ((int)li == net->conv1) ? 3 : l.Cin;
            t.w_off = add(l.scope + "/weights", (long long)l.KH * l.KW * t.cin_real * l.Cout, false);
            t.g_off = add(l.scope + "/BatchNorm/gamma", l.Cout, false);
            t.b_off = add(l.scope + "/BatchNorm/beta", l.Cout, false);
            t.mean_off = add(l.scope + "/BatchNorm/moving_mean", l.Cout, true);
            t.var_off = add(l.scope + "/BatchNorm/moving_variance", l.Cout, true);
            if ((int)li != net->conv1) {
                t.cinP = coutp_for(l.Cin);
                t.nkT = nk_for(l.KH, l.KW, l.Cout);
            }
        }
        if (t.nkT > 0) {
            if (hipMalloc(&t.d_wT, (size_t)t.nkT * 8 * t.cinP * 16) != hipSuccess) {
                delete tr;
                return fail(DGP_ERR_HIP, "dgp_trainer_create: hipMalloc (dgrad panels) failed");
            }
        }
    }
    const size_t nb = (size_t)tr->n_train * sizeof(float);
    if (hipMalloc(&tr->params, nb) != hipSuccess || hipMalloc(&tr->grads, nb) != hipSuccess ||
        hipMalloc(&tr->mom, nb) != hipSuccess || hipMalloc(&tr->stats, (size_t)tr->n_stat * sizeof(float)) != hipSuccess ||
        hipMalloc(&tr->d_sumsq, 2 * sizeof(double)) != hipSuccess || hipMalloc(&tr->d_gnorm, sizeof(float)) != hipSuccess ||
        hipMemset(tr->d_sumsq, 0, 2 * sizeof(double)) != hipSuccess || hipMemset(tr->d_gnorm, 0, sizeof(float)) != hipSuccess ||
        hipHostMalloc(&tr->h_status, 16 * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        delete tr;
        return fail(DGP_ERR_HIP, "dgp_trainer_create: hipMalloc failed");
    }
    (void)hipMemset(tr->params, 0, nb); (void)hipMemset(tr->grads, 0, nb); (void)hipMemset(tr->mom, 0, nb);
    if (hipMalloc(&tr->d_rng_pool, (size_t)RANGE_POOL * ABSMAX_SLOTS * sizeof(float)) != hipSuccess ||
        hipMalloc(&tr->d_rng_prev, (size_t)RANGE_POOL * ABSMAX_SLOTS * sizeof(float)) != hipSuccess ||
        hipMalloc(&tr->d_fast_flag, sizeof(int)) != hipSuccess || hipMemset(tr->d_fast_flag, 0, sizeof(int)) != hipSuccess ||
        hipMemset(tr->d_rng_pool, 0, (size_t)RANGE_POOL * ABSMAX_SLOTS * sizeof(float)) != hipSuccess ||
        hipMemset(tr->d_rng_prev, 0, (size_t)RANGE_POOL * ABSMAX_SLOTS * sizeof(float)) != hipSuccess ||
        hipMalloc(&tr->d_wrng, (size_t)2 * net->layers.size() * ABSMAX_SLOTS * sizeof(float)) != hipSuccess) {
        delete tr;
        return fail(DGP_ERR_HIP, "dgp_trainer_create: hipMalloc failed");
    }
    *out = tr;
    return DGP_OK;
}

void dgp_trainer_destroy(dgp_trainer* tr) {
    if (tr && tr->net && tr->net->owner == tr) { tr->net->owner = nullptr; tr->net->owner_sync = nullptr; }      // (the net outlives its trainer: train.py)
    delete tr;
}

int dgp_trainer_num_tensors(const dgp_trainer* tr, int32_t* n_tensors, int64_t* n_trainable_floats, int64_t* n_stat_floats) {
    if (!tr) return fail(DGP_ERR_INVALID, "dgp_trainer_num_tensors: null");
    if (n_tensors) *n_tensors = (int32_t)tr->names.size();
    if (n_trainable_floats) *n_trainable_floats = tr->n_train;
    if (n_stat_floats) *n_stat_floats = tr->n_stat;
    return DGP_OK;
}

int dgp_trainer_tensor_info(const dgp_trainer* tr, int32_t i, char* name, int32_t cap, int64_t* offset, int64_t* size,
                            int32_t* is_stat) {
    if (!tr || i < 0 || i >= (int)tr->names.size()) return fail(DGP_ERR_INVALID, "dgp_trainer_tensor_info: bad index");
    if (name && cap > 0) { strncpy(name, tr->names[i].c_str(), cap - 1); name[cap - 1] = 0; }
    if (offset) *offset = tr->offs[i];
    if (size) *size = tr->sizes[i];
    if (is_stat) *is_stat = tr->is_stat[i];
    return DGP_OK;
}

/* which: 0 params, 1 grads, 2 momentum, 3 frozen statistics */
float* dgp_trainer_buffer(dgp_trainer* tr, int32_t which) {
    if (!tr) return nullptr;
    switch (which) {
        case 0: return tr->params; case 1: return tr->grads; case 2: return tr->mom; case 3: return tr->stats;
        case 4: return reinterpret_cast<float*>(tr->d_fast_flag);      // ONE int32: != 0 when a 16-bit pass left its predicted ranges (data-parallel: all-reduce MAX)
    }
    return nullptr;
}

int dgp_trainer_workspace_bytes(const dgp_trainer* tr, int32_t nt, size_t* out_bytes) {
    if (!tr || !out_bytes || nt < 1) return fail(DGP_ERR_INVALID, "dgp_trainer_workspace_bytes: bad argument");
    *out_bytes = make_tplan(tr, nt).total;
    return DGP_OK;
}

}  // extern "C"

// the heads' panels that only the parity path reads + the H2 cells of all panels, from the current master parameters / fp32 panels
static int refresh_parity_panels(dgp_trainer* tr, hipStream_t s) {
    dgp_net* net = tr->net;
    for (int hd : {net->head_part, net->head_locref}) {
        ConvLayer& l = net->layers[hd];
        TLayer& t = tr->tl[hd];
        const float* w = tr->params + t.w_off;
        const int njt = l.Cout / 4;
        hipLaunchKernelGGL(pack_head_fwd_kernel, dim3(grid_for((long long)l.nk * 8 * l.CoutP)), dim3(256), 0, s, w, njt, l.Cin, l.CoutP, l.nk * 8, l.d_w);
        hipLaunchKernelGGL(pack_head_dgrad_kernel, dim3(grid_for((long long)t.nkT * 8 * t.cinP)), dim3(256), 0, s, w, njt, l.Cin, t.cpad, t.cinP,
                           t.nkT * 8, t.d_wT);
    }
    if (tr->d_h3_table) TRY_HIP(launch_pack_h3_all(reinterpret_cast<const PackH3Desc*>(tr->d_h3_table), tr->n_h3, s));
    TRY_HIP(hipGetLastError());
    tr->parity_stale = false;
    return DGP_OK;
}

// dgp_net::owner_sync: a forward on the trainer's net (Trainer.net.infer between steps) reads the heads' 2x2-conv panels and the H2 cells,
// which a 16-bit trainer leaves stale between plain passes
static int trainer_owner_sync(void* owner, void* stream) {
    dgp_trainer* tr = static_cast<dgp_trainer*>(owner);
    if (!tr || !tr->parity_stale) return DGP_OK;
    g_ctx = &tr->ctx;
    return refresh_parity_panels(tr, (hipStream_t)stream);
}

extern "C" {

// master parameters -> forward panels / folded BN / data-gradient panels of the engine
int dgp_trainer_sync_weights(dgp_trainer* tr, void* stream) {
    if (tr) g_ctx = &tr->ctx;
    if (!tr) return fail(DGP_ERR_INVALID, "dgp_trainer_sync_weights: null");
    dgp_net* net = tr->net;
    hipStream_t s = (hipStream_t)stream;
    const float eps = net->desc.bn_eps;
    const size_t nl_all = net->layers.size();
    if (tr->d_wrng) TRY_HIP(hipMemsetAsync(tr->d_wrng, 0, 2 * nl_all * ABSMAX_SLOTS * sizeof(float), s));
    // one launch for the panels / folded BN of all non-head layers (DGP_PACK_MERGED=0: one launch per layer and job, as before)
    static const bool merged_env = (dgp_tune("DGP_PACK_MERGED", 1) != 0);
    const bool merged = merged_env;
    // tier 1, every sync after the first: the parity-only panels wait for a plain pass (A/B switch DGP_TRAIN_LAZY_PARITY=0)
    static const bool lazy_env = (dgp_tune("DGP_TRAIN_LAZY_PARITY", 1) != 0);
    const bool lazy = lazy_env && merged && g_train_cells && tr->tier == 1 && tr->d_h3_table && tr->d_h1_table && tr->d_wrng;
    tr->parity_stale = lazy;
    for (size_t li = 0; li < net->layers.size(); ++li) {
        ConvLayer& l = net->layers[li];
        TLayer& t = tr->tl[li];
        float* rng_f = tr->d_wrng ? tr->d_wrng + li * ABSMAX_SLOTS : nullptr;          // ranges of the panels packed below
        float* rng_b = tr->d_wrng ? tr->d_wrng + (nl_all + li) * ABSMAX_SLOTS : nullptr;
        const bool head = ((int)li == net->head_part || (int)li == net->head_locref);
        const size_t nfl = (size_t)l.nk * 8 * l.CoutP * 4;
        if (!l.d_w) TRY_HIP(hipMalloc(&l.d_w, nfl * sizeof(float)));
        if (!l.d_scale) TRY_HIP(hipMalloc(&l.d_scale, l.Cout * sizeof(float)));
        if (!l.d_bias) TRY_HIP(hipMalloc(&l.d_bias, l.Cout * sizeof(float)));
        const float* w = tr->params + t.w_off;
        const long long tot = (long long)l.nk * 8 * l.CoutP;
        if (!head && merged) continue;        // packed by pack_all_kernel below
        if (head) {
            const int njt = l.Cout / 4;
            if (!lazy) hipLaunchKernelGGL(pack_head_fwd_kernel, dim3(grid_for(tot)), dim3(256), 0, s, w, njt, l.Cin, l.CoutP, l.nk * 8, l.d_w);
            if (g_train_cells && merged && rng_f) {      // pointwise form of the same head for the forward pass (cells built below)
                l.coutp_pw = head_pw_coutp(16 * njt);      // same width as dgp_net_load_weights gives the shared buffers
                const size_t npw = (size_t)nk_for(1, 1, l.Cin) * 8 * l.coutp_pw * 4;
                if (!l.d_w_pw) TRY_HIP(hipMalloc(&l.d_w_pw, npw * sizeof(float)));
                if (!l.d_wh3_pw) TRY_HIP(hipMalloc(&l.d_wh3_pw, npw * sizeof(float)));
                hipLaunchKernelGGL(pack_head_pw_kernel, dim3(grid_for((long long)(l.Cin >> 2) * l.coutp_pw)), dim3(256), 0, s, w, njt,
                                   l.Cin, l.coutp_pw, l.d_w_pw, rng_f);
            }
            hipLaunchKernelGGL(head_bias_kernel, dim3(1), dim3(256), 0, s, tr->params + t.b_off, njt, l.d_bias);
            const long long totT = (long long)t.nkT * 8 * t.cinP;
            if (!lazy) hipLaunchKernelGGL(pack_head_dgrad_kernel, dim3(grid_for(totT)), dim3(256), 0, s, w, njt, l.Cin, t.cpad, t.cinP,
                                          t.nkT * 8, t.d_wT);
        } else {
            hipLaunchKernelGGL(pack_fwd_kernel, dim3(grid_for(tot)), dim3(256), 0, s, w, l.KH * l.KW, t.cin_real, l.Cin, l.Cout,
                               l.CoutP, l.nk * 8, l.d_w, rng_f);
            if (g_train_cells && rng_f && l.Cin >= 32) {          // cells of the forward panel, scaled by the range tracked just above
                if (!l.d_wh3) TRY_HIP(hipMalloc(&l.d_wh3, nfl * sizeof(float)));
                TRY_HIP(launch_pack_h3(l.d_w, l.nk, l.CoutP, rng_f, l.d_wh3, s));
                g_ctx->cells[l.d_w] = reinterpret_cast<const float*>(l.d_wh3);
            }
            hipLaunchKernelGGL(fold_bn_kernel, dim3((l.Cout + 255) / 256), dim3(256), 0, s, tr->params + t.g_off,
                               tr->params + t.b_off, tr->stats + t.mean_off, tr->stats + t.var_off, eps, l.Cout, l.d_scale,
                               l.d_bias);
            if (t.d_wT) {
                const long long totT = (long long)t.nkT * 8 * t.cinP;
                hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid_for(totT)), dim3(256), 0, s, w, l.d_scale, l.KH * l.KW, l.Cin,
                                   l.Cout, t.cinP, t.nkT * 8, t.d_wT, rng_b);
                if (g_train_cells && rng_b && l.Cout >= 32) {
                    if (!t.d_wTh3) TRY_HIP(hipMalloc(&t.d_wTh3, (size_t)t.nkT * 8 * t.cinP * 16));
                    TRY_HIP(launch_pack_h3(t.d_wT, t.nkT, t.cinP, rng_b, t.d_wTh3, s));
                    g_ctx->cells[t.d_wT] = t.d_wTh3;
                }
            }
        }
    }
    if (merged) {
        if (!tr->d_pack_table) {
            std::vector<PackDesc> tab;
            for (size_t li = 0; li < net->layers.size(); ++li) {
                if ((int)li == net->head_part || (int)li == net->head_locref) continue;
                const ConvLayer& l = net->layers[li];
                const TLayer& t = tr->tl[li];
                PackDesc d{};
                d.w_off = t.w_off; d.g_off = t.g_off; d.b_off = t.b_off; d.mean_off = t.mean_off; d.var_off = t.var_off;
                d.taps = l.KH * l.KW; d.cin_real = t.cin_real; d.cin = l.Cin; d.cout = l.Cout; d.coutP = l.CoutP;
                d.nchunks_f = l.nk * 8; d.cinP = t.cinP; d.nchunks_b = t.nkT * 8;
                d.d_w = l.d_w; d.rng_f = tr->d_wrng ? tr->d_wrng + li * ABSMAX_SLOTS : nullptr;
                d.d_wT = t.d_wT; d.rng_b = tr->d_wrng ? tr->d_wrng + (nl_all + li) * ABSMAX_SLOTS : nullptr;
                d.d_scale = l.d_scale; d.d_bias = l.d_bias;
                tab.push_back(d);
            }
            tr->n_pack = (int)tab.size();
            TRY_HIP(hipMalloc(&tr->d_pack_table, tab.size() * sizeof(PackDesc)));
            TRY_HIP(hipMemcpy(tr->d_pack_table, tab.data(), tab.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(pack_all_kernel, dim3(256, (unsigned)tr->n_pack, 3), dim3(256), 0, s,
                           reinterpret_cast<const PackDesc*>(tr->d_pack_table), tr->params, tr->stats, eps);
        if (g_train_cells && tr->d_wrng) {        // the panels' fp16 cells, split with the ranges the launch above has just tracked
            if (!tr->d_h3_table) {
                std::vector<PackH3Desc> tab;
                for (size_t li = 0; li < net->layers.size(); ++li) {
                    if ((int)li == net->head_part || (int)li == net->head_locref) continue;
                    ConvLayer& l = net->layers[li];
                    TLayer& t = tr->tl[li];
                    if (l.Cin >= 32) {
                        if (!l.d_wh3) TRY_HIP(hipMalloc(&l.d_wh3, (size_t)l.nk * 8 * l.CoutP * 16));
                        tab.push_back(PackH3Desc{l.d_w, l.nk * 4, l.CoutP, tr->d_wrng + li * ABSMAX_SLOTS, l.d_wh3});
                        g_ctx->cells[l.d_w] = reinterpret_cast<const float*>(l.d_wh3);
                    }
                    if (t.d_wT && l.Cout >= 32) {
                        if (!t.d_wTh3) TRY_HIP(hipMalloc(&t.d_wTh3, (size_t)t.nkT * 8 * t.cinP * 16));
                        tab.push_back(PackH3Desc{t.d_wT, t.nkT * 4, t.cinP, tr->d_wrng + (nl_all + li) * ABSMAX_SLOTS, t.d_wTh3});
                        g_ctx->cells[t.d_wT] = t.d_wTh3;
                    }
                }
                for (int hd : {net->head_part, net->head_locref}) {         // the heads' pointwise panels (range slot of the layer)
                    ConvLayer& l = net->layers[hd];
                    if (l.d_w_pw && l.d_wh3_pw)
                        tab.push_back(PackH3Desc{l.d_w_pw, nk_for(1, 1, l.Cin) * 4, l.coutp_pw, tr->d_wrng + (size_t)hd * ABSMAX_SLOTS, l.d_wh3_pw});
                }
                tr->n_h3 = (int)tab.size();
                TRY_HIP(hipMalloc(&tr->d_h3_table, tab.size() * sizeof(PackH3Desc)));
                TRY_HIP(hipMemcpy(tr->d_h3_table, tab.data(), tab.size() * sizeof(PackH3Desc), hipMemcpyHostToDevice));
            }
            if (!lazy) TRY_HIP(launch_pack_h3_all(reinterpret_cast<const PackH3Desc*>(tr->d_h3_table), tr->n_h3, s));
            if (tr->tier == 1) {              // 16-bit tier: the same panels as high-only H1 cells (K-steps of 64 channels: K % 64 == 0)
                if (!tr->d_h1_table) {
                    std::vector<PackH3Desc> tab;
                    for (size_t li = 0; li < net->layers.size(); ++li) {
                        if ((int)li == net->head_part || (int)li == net->head_locref) continue;
                        ConvLayer& l = net->layers[li];
                        TLayer& t = tr->tl[li];
                        if (l.Cin >= 64 && (l.Cin % 64) == 0 && (l.nk % 2) == 0 && l.CoutP % 64 == 0) {
                            if (!l.d_wh1) TRY_HIP(hipMalloc(&l.d_wh1, (size_t)l.nk * 8 * l.CoutP * 8));
                            tab.push_back(PackH3Desc{l.d_w, l.nk * 4, l.CoutP, tr->d_wrng + li * ABSMAX_SLOTS, l.d_wh1});
                            g_ctx->cells1[l.d_w] = l.d_wh1;
                        }
                        if (t.d_wT && l.Cout >= 64 && (l.Cout % 64) == 0 && (t.nkT % 2) == 0 && t.cinP % 64 == 0) {
                            if (!t.d_wTh1) TRY_HIP(hipMalloc(&t.d_wTh1, (size_t)t.nkT * 8 * t.cinP * 8));
                            tab.push_back(PackH3Desc{t.d_wT, t.nkT * 4, t.cinP, tr->d_wrng + (nl_all + li) * ABSMAX_SLOTS, t.d_wTh1});
                            g_ctx->cells1[t.d_wT] = t.d_wTh1;
                        }
                    }
                    for (int hd : {net->head_part, net->head_locref}) {
                        ConvLayer& l = net->layers[hd];
                        if (l.d_w_pw) {
                            if (!l.d_wh1_pw) TRY_HIP(hipMalloc(&l.d_wh1_pw, (size_t)nk_for(1, 1, l.Cin) * 8 * l.coutp_pw * 8));
                            tab.push_back(PackH3Desc{l.d_w_pw, nk_for(1, 1, l.Cin) * 4, l.coutp_pw, tr->d_wrng + (size_t)hd * ABSMAX_SLOTS, l.d_wh1_pw});
                        }
                    }
                    tr->n_h1 = (int)tab.size();
                    TRY_HIP(hipMalloc(&tr->d_h1_table, tab.size() * sizeof(PackH3Desc)));
                    TRY_HIP(hipMemcpy(tr->d_h1_table, tab.data(), tab.size() * sizeof(PackH3Desc), hipMemcpyHostToDevice));
                }
                TRY_HIP(launch_pack_h1_all(reinterpret_cast<const PackH3Desc*>(tr->d_h1_table), tr->n_h1, s));
                // both heads' data-gradient panels as one panel + its H1 cells (one launch gives d features of both heads)
                static const bool heads_h1_env = (dgp_env("DGP_TRAIN_HEADS_H1", 1) != 0);      // A/B switch
                {
                    const ConvLayer &l0 = net->layers[net->head_part], &l1 = net->layers[net->head_locref];
                    const TLayer &t0 = tr->tl[net->head_part], &t1 = tr->tl[net->head_locref];
                    const int CT = heads_ct(net->desc.num_joints), nkT = nk_for(2, 2, CT);
                    if (heads_h1_env && l0.Cin == l1.Cin && t0.cinP == t1.cinP && t0.cpad + t1.cpad <= CT && (nkT % 2) == 0 && t0.cinP % 64 == 0) {
                        const size_t pbytes = (size_t)nkT * 8 * t0.cinP * 16;
                        if (!tr->d_hmT) TRY_HIP(hipMalloc(&tr->d_hmT, pbytes));
                        if (!tr->d_hmT_h1) TRY_HIP(hipMalloc(&tr->d_hmT_h1, pbytes / 2));
                        if (!tr->d_hm_rng) TRY_HIP(hipMalloc(&tr->d_hm_rng, ABSMAX_SLOTS * sizeof(float)));
                        tr->hm_ct = CT; tr->hm_nk = nkT;
                        TRY_HIP(hipMemsetAsync(tr->d_hm_rng, 0, ABSMAX_SLOTS * sizeof(float), s));
                        const long long totT = (long long)nkT * 8 * t0.cinP;
                        hipLaunchKernelGGL(pack_heads_dgrad_kernel, dim3(grid_for(totT)), dim3(256), 0, s, tr->params + t0.w_off, l0.Cout / 4, t0.cpad,
                                           tr->params + t1.w_off, l1.Cout / 4, t1.cpad, l0.Cin, CT, t0.cinP, nkT * 8, tr->d_hmT, tr->d_hm_rng);
                        TRY_HIP(launch_pack_h1(tr->d_hmT, nkT, t0.cinP, tr->d_hm_rng, tr->d_hmT_h1, s));
                        g_ctx->cells1[tr->d_hmT] = tr->d_hmT_h1;
                    }
                }
                // the fused root block's weight cells (stem_pool_fused_kernel): row panel of conv1 from the panel pack_all_kernel just wrote
                static const bool stem_fused_env = (dgp_tune("DGP_TRAIN_STEM_FUSED", 1) != 0);      // A/B switch
                ConvLayer& lc = net->layers[net->conv1];
                if (stem_fused_env && lc.CoutP == 64 && lc.Cin == 4 && lc.KH == 7 && lc.KW == 7) {
                    const size_t nrow = (size_t)56 * lc.CoutP * 4;
                    if (!tr->d_stem_rows) TRY_HIP(hipMalloc(&tr->d_stem_rows, nrow * sizeof(float)));
                    if (!tr->d_stem_cells) TRY_HIP(hipMalloc(&tr->d_stem_cells, nrow * sizeof(float)));
                    const float* mp = net->desc.mean_pixel;
                    hipLaunchKernelGGL(stem_rows_kernel, dim3((56 * lc.CoutP + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float4*>(lc.d_w),
                                       lc.CoutP, roundf(mp[0]) - mp[0], roundf(mp[1]) - mp[1], roundf(mp[2]) - mp[2],
                                       reinterpret_cast<float4*>(tr->d_stem_rows));
                    TRY_HIP(launch_pack_h3(tr->d_stem_rows, 7, lc.CoutP, tr->d_wrng + (size_t)net->conv1 * ABSMAX_SLOTS, tr->d_stem_cells, s));
                }
            }
        }
    }
    TRY_HIP(hipGetLastError());
    net->wmax_valid = false;      // panels changed: weight ranges of the fp16-split kernels are stale
    net->loaded = true;
    return DGP_OK;
}

int dgp_train_forward(dgp_trainer* tr, const uint8_t* frames, int32_t nt, void* workspace, size_t workspace_bytes,
                      float** scmap, float** locref, void* stream) {
    if (tr) g_ctx = &tr->ctx;
    if (!tr || !frames || !workspace) return fail(DGP_ERR_INVALID, "dgp_train_forward: null argument");
    dgp_net* net = tr->net;
    if (!net->loaded) return fail(DGP_ERR_STATE, "dgp_train_forward: call dgp_trainer_sync_weights first");
    const TPlan pl = make_tplan(tr, nt);
    if (workspace_bytes < pl.total) return fail(DGP_ERR_INVALID, "dgp_train_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    g_ctx->tail_slab = F(pl.tail);
    range_pass_begin(tr, s, false, tr->d_fast_flag);      // (also clears the range flag of a fast pass: fast needs the ranges on)
    g_ctx->shadow_base.clear();
    g_ctx->h2_slots.clear();
    g_shadow_want = true;
    g_h2 = H2Launch();
    // Fast pass (dgp_trainer_fast_mode; the host asks for it once a pass of the same shapes has left its ranges behind): from the first
    // unit with a 128-channel bottleneck on (block2) every retained activation is an H2 tensor -- fp16 high / low cells with a scale
    // predicted from the previous step's range, no fp32 twin -- so these convs run the inference engine's cell kernels (no split in the
    // K loop, no second copy written) and the weight gradients read them in place.  ub: first such unit; its input is the fp16 copy
    // (ConvArgs::shadow) of block1's output.
    size_t ub = net->units.size();
    for (size_t ui = 1; ui < net->units.size(); ++ui)
        if (net->units[ui].depth_bn >= 128 && pl.sh_xo[ui - 1]) { ub = ui; break; }
    // (16-bit tier: EVERY unit -- block1 included -- keeps H1 tensors; the first one reads the H1 copy of the pool output)
    if (tr->tier == 1 && pl.sh_pool && pool_idx_on()) ub = 0;
    const bool fast = tr->fast_next && g_wgrad_dma && g_train_cells && g_ctx->rng.on && ub < net->units.size() && pl.feat32 &&
                      (tr->tier != 1 || tr->d_h1_table);
    const int FMT = tr->tier == 1 ? 2 : 1;       // cell format of this pass's H2 / H1 tensors
    tr->fwd_fast = fast;
    tr->fwd_fmt = fast ? FMT : 0;
    if (!fast && tr->parity_stale) {             // a plain pass after lazy syncs: its panels first
        const int rcp = refresh_parity_panels(tr, s);
        if (rcp) return rcp;
    }
    g_shadow_fmt = (fast && FMT == 2) ? 2 : 1;   // (the one copy a tier-1 forward pass writes: block1's output, read by the first H1 unit)
    for (size_t ui = 0; ui < net->units.size(); ++ui) {
        if (fast && ui >= ub) continue;              // (H2 tensors register themselves as they are written)
        if (pl.sh_r1[ui]) g_ctx->shadow_base[F(pl.r1[ui])] = F(pl.sh_r1[ui]);
        if (pl.sh_r2[ui]) g_ctx->shadow_base[F(pl.r2[ui])] = F(pl.sh_r2[ui]);
        if (pl.sh_xo[ui]) g_ctx->shadow_base[F(pl.xo[ui])] = F(pl.sh_xo[ui]);
    }
    const dgp_net_desc& d = net->desc;
    const int B = nt;
    // The forward pass as TWO chains of frames (DGP_FWD_CHAINS=1: one): frames [0, n1) on the caller's stream, [n1, B) on the trainer's
    // second stream.  At 11 frames the grids of block3 / block4 cover half of the chip, and two independent chains drift apart so that
    // one chain's small or tail-heavy layers run under the other's.  Both chains write frame ranges of the SAME activation tensors
    // (the backward pass sees one batch) and max into the same range slots.
    static const bool side_env = (dgp_tune("DGP_WGRAD_OVERLAP", 1) != 0);
    static const int chains_env = dgp_tune("DGP_FWD_CHAINS", 2);
    if (side_env && !g_ctx->s2) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        TRY_HIP(hipStreamCreateWithPriority(&g_ctx->s2, hipStreamNonBlocking, lo));
    }
    g_ctx->ev_next = 0;                          // (the previous backward pass has joined: its events are free again)
    const bool two = side_env && g_ctx->s2 && chains_env >= 2 && B >= 4;
    const int n1 = two ? (B + 1) / 2 : B;
    const ConvLayer& c1 = net->layers[net->conv1];
    static const bool pool_idx = (dgp_tune("DGP_POOL_IDX", 1) != 0);       // A/B switch (0: re-scan in backward)
    const bool stem_fused = fast && FMT == 2 && ub == 0 && pool_idx && tr->d_stem_cells && tr->d_wrng && c1.d_scale && c1.d_bias;
    tr->fwd_stem_fused = stem_fused;
    if (two) {
        hipEvent_t ready = g_ctx->take_event();
        if (!ready) return fail(DGP_ERR_HIP, "forward chains: hipEventCreate failed");
        TRY_HIP(hipEventRecord(ready, s));       // frames, weights and the zeroed range slots are in place
        TRY_HIP(hipStreamWaitEvent(g_ctx->s2, ready, 0));
    }
    int h = net->hp, w = net->wp;
    const float* xin = F(pl.pool);
    // host enqueue order: stage by stage, chain 0 then chain 1 (a chain enqueued whole before the other would start a millisecond late)
    struct Chain { int n0, nB; hipStream_t cs; bool second; int hh, ww; size_t x_off; int x_c; };
    Chain chains[2] = {{0, n1, s, false, net->hp, net->wp, pl.pool, 64}, {n1, B - n1, g_ctx->s2, true, net->hp, net->wp, pl.pool, 64}};
    const int nchains = two ? 2 : 1;
    float* const slab_keep = g_ctx->tail_slab;
    struct Restore { float*& ref; float* v; ~Restore() { ref = v; } } restore{g_ctx->tail_slab, slab_keep};
    auto root = [&](Chain& c) -> int {
        g_ctx->tail_slab = c.second ? nullptr : slab_keep;      // the K-split slab belongs to the first chain's launches
        g_ctx->rng.chain2 = c.second;
        const int n0 = c.n0, nB = c.nB;
        hipStream_t cs = c.cs;
        auto at = [&](size_t off, size_t per_frame) { return F(off) + (size_t)n0 * per_frame; };
        const size_t px_in = (size_t)d.in_h * d.in_w, px1 = (size_t)net->h1 * net->w1, pxp = (size_t)net->hp * net->wp;
        TRY_HIP(launch_preprocess(frames + (size_t)n0 * px_in * 3, (long long)nB * d.in_h * d.in_w, d.mean_pixel[0], d.mean_pixel[1],
                                  d.mean_pixel[2], at(pl.p0, px_in * 4), cs));
        if (stem_fused) {
            // 16-bit tier: the root block as the inference engine's ONE kernel (uint8 frame -> conv1 + BN + ReLU -> max-pool -> H1 cells of the
            // pool output, scale predicted from conv1's range one step ago) that also records each window's first maximum.  No conv1 map, no
            // fp32 pool output: the first unit reads the H1 tensor, the pool's backward needs only the record.  (The centred frame above
            // stays: the stem's weight gradient reads it.)  The launch takes conv1's range slot, in conv1's place in the order of slots.
            float* slot = range_take();
            const float* yprev = range_prev_of(slot);
            if (!slot || !yprev) return fail(DGP_ERR_STATE, "16-bit tier: the root block's output has no predicted range");
            range_set(F(pl.c1), slot);
            range_set(F(pl.pool), slot);
            g_ctx->shadow_base[F(pl.pool)] = F(pl.sh_pool);
            g_ctx->shadow_prev[F(pl.pool)] = yprev;
            TRY_HIP(launch_stem_pool_fused(frames + (size_t)n0 * px_in * 3, nB, d.in_h, d.in_w, tr->d_stem_cells,
                                           tr->d_wrng + (size_t)net->conv1 * ABSMAX_SLOTS, c1.d_scale, c1.d_bias, d.mean_pixel[0], d.mean_pixel[1],
                                           d.mean_pixel[2], 0.f, reinterpret_cast<float*>(reinterpret_cast<char*>(F(pl.sh_pool)) + (size_t)n0 * pxp * 64 * 2),
                                           slot, cs, 1, yprev, reinterpret_cast<unsigned char*>(ws + pl.pidx) + (size_t)n0 * pxp * 64));
            return DGP_OK;
        }
        TRY_HIP(conv_launch(c1, c1.d_w, c1.nk, c1.CoutP, at(pl.p0, px_in * 4), nB, d.in_h, d.in_w, 4, 3, 3, net->h1, net->w1, 64, 2, 0,
                            c1.d_scale, c1.d_bias, nullptr, 0, 0, 0, nullptr, true, 0, 0, at(pl.c1, px1 * 64), cs, F(pl.p0), F(pl.c1)));
        if (pool_idx) {
            int pth = (net->hp - 1) * 2 + 3 - net->h1; if (pth < 0) pth = 0;
            int ptw = (net->wp - 1) * 2 + 3 - net->w1; if (ptw < 0) ptw = 0;
            const long long totp = (long long)nB * net->hp * net->wp * 16;
            uint2* yh1 = nullptr;
            const float* yprev = nullptr;
            if (fast && FMT == 2 && ub == 0) {        // H1 copy of the pool output for the first unit (scale: conv1's range one step ago)
                yprev = range_prev_of(range_of(F(pl.c1)));
                if (!yprev) return fail(DGP_ERR_STATE, "16-bit tier: the root block's output has no predicted range");
                yh1 = reinterpret_cast<uint2*>(reinterpret_cast<char*>(F(pl.sh_pool)) + (size_t)n0 * pxp * 64 * 2);
                g_ctx->shadow_base[F(pl.pool)] = F(pl.sh_pool);
                g_ctx->shadow_prev[F(pl.pool)] = yprev;
            }
            hipLaunchKernelGGL(maxpool_fwd_idx_kernel, dim3(grid_for(totp)), dim3(256), 0, cs, at(pl.c1, px1 * 64), nB, net->h1, net->w1, 16,
                               net->hp, net->wp, pth / 2, ptw / 2, at(pl.pool, pxp * 64),
                               reinterpret_cast<uchar4*>(ws + pl.pidx) + (size_t)n0 * pxp * 16, yh1, yprev);
        } else {
            TRY_HIP(launch_maxpool(at(pl.c1, px1 * 64), nB, net->h1, net->w1, 64, at(pl.pool, pxp * 64), cs));
        }
        range_set(F(pl.pool), range_of(F(pl.c1)));      // max-pooling cannot raise the maximum
        return DGP_OK;
    };
    auto unit = [&](Chain& c, size_t ui) -> int {
        g_ctx->tail_slab = c.second ? nullptr : slab_keep;
        g_ctx->rng.chain2 = c.second;
        const int n0 = c.n0, nB = c.nB, hh = c.hh, ww = c.ww;
        hipStream_t cs = c.cs;
        auto at = [&](size_t off, size_t per_frame) { return F(off) + (size_t)n0 * per_frame; };
        const Unit& u = net->units[ui];
        const int ho = (hh + u.stride - 1) / u.stride, wo = (ww + u.stride - 1) / u.stride;
        const size_t pin = (size_t)hh * ww, pout = (size_t)ho * wo;
        const bool h2u = fast && ui >= ub;             // this unit's tensors are H2
        // (the first H2 unit reads the fp16 copy of its fp32 input; ranges and scales go by the tensor's own name, F(c.x_off))
        // (frame offsets inside H1 tensors are half the fp32 ones: atf)
        auto atf = [&](size_t off, size_t per_frame, bool cells1) {
            return cells1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(F(off)) + (size_t)n0 * per_frame * 2) : F(off) + (size_t)n0 * per_frame;
        };
        const bool h1u = h2u && FMT == 2;
        const float* x = (h2u && ui == ub) ? atf(ui == 0 ? pl.sh_pool : pl.sh_xo[ui - 1], pin * c.x_c, h1u) : atf(c.x_off, pin * c.x_c, h1u && ui > ub);
        const float* res = x;
        const void* res_key = F(c.x_off);
        int res_s = u.stride, res_H = hh, res_W = ww;
        auto h2_next = [&](bool with_res) {
            if (!h2u) return;
            g_h2.in_fmt = FMT; g_h2.out_fmt = FMT;
            if (with_res) { g_h2.res_fmt = FMT; g_h2.res_key = res_key; }
        };
        if (u.sc >= 0) {
            const ConvLayer& l = net->layers[u.sc];
            h2_next(false);
            TRY_HIP(conv_launch(l, l.d_w, l.nk, l.CoutP, x, nB, hh, ww, l.Cin, 0, 0, ho, wo, l.Cout, u.stride, 0, l.d_scale,
                                l.d_bias, nullptr, 0, 0, 0, nullptr, false, 0, 0, atf(pl.sc[ui], pout * u.depth, h1u), cs, F(c.x_off),
                                F(pl.sc[ui])));
            res = atf(pl.sc[ui], pout * u.depth, h1u); res_s = 1; res_H = ho; res_W = wo;
            res_key = F(pl.sc[ui]);
        }
        const ConvLayer& l1 = net->layers[u.c1];
        h2_next(false);
        TRY_HIP(conv_launch(l1, l1.d_w, l1.nk, l1.CoutP, x, nB, hh, ww, l1.Cin, 0, 0, hh, ww, l1.Cout, 1, 0, l1.d_scale,
                            l1.d_bias, nullptr, 0, 0, 0, nullptr, true, 0, 0, atf(pl.r1[ui], pin * u.depth_bn, h1u), cs, F(c.x_off),
                            F(pl.r1[ui])));
        const ConvLayer& l2 = net->layers[u.c2];
        const int pb_h = pad_before_for(hh, 3, u.stride, u.rate, true), pb_w = pad_before_for(ww, 3, u.stride, u.rate, true);
        h2_next(false);
        TRY_HIP(conv_launch(l2, l2.d_w, l2.nk, l2.CoutP, atf(pl.r1[ui], pin * u.depth_bn, h1u), nB, hh, ww, l2.Cin, pb_h, pb_w, ho, wo,
                            l2.Cout, u.stride, 0, l2.d_scale, l2.d_bias, nullptr, 0, 0, 0, nullptr, true, 0, 0,
                            atf(pl.r2[ui], pout * u.depth_bn, h1u), cs, F(pl.r1[ui]), F(pl.r2[ui])));
        const ConvLayer& l3 = net->layers[u.c3];
        h2_next(true);
        TRY_HIP(conv_launch(l3, l3.d_w, l3.nk, l3.CoutP, atf(pl.r2[ui], pout * u.depth_bn, h1u), nB, ho, wo, l3.Cin, 0, 0, ho, wo, l3.Cout,
                            1, 0, l3.d_scale, l3.d_bias, res, res_s, res_H, res_W, nullptr, true, 0, 0,
                            atf(pl.xo[ui], pout * u.depth, h1u), cs, F(pl.r2[ui]), F(pl.xo[ui])));
        c.x_off = pl.xo[ui]; c.x_c = u.depth; c.hh = ho; c.ww = wo;
        return DGP_OK;
    };
    {
        int rc0;
        for (int ci = 0; ci < nchains; ++ci) if ((rc0 = root(chains[ci]))) return rc0;
        for (size_t ui = 0; ui < net->units.size(); ++ui)
            for (int ci = 0; ci < nchains; ++ci) if ((rc0 = unit(chains[ci], ui))) return rc0;
        g_ctx->tail_slab = slab_keep;
        g_ctx->rng.chain2 = false;
        if (two) {
            hipEvent_t done = g_ctx->take_event();
            if (!done) return fail(DGP_ERR_HIP, "forward chains: hipEventCreate failed");
            TRY_HIP(hipEventRecord(done, g_ctx->s2));
            TRY_HIP(hipStreamWaitEvent(s, done, 0));
            if (g_ctx->rng.on && g_ctx->rng.next2 > 0) {      // ranges of the whole tensors = max of the chains' (same slot order)
                if (g_ctx->rng.next2 != g_ctx->rng.next) return fail(DGP_ERR_STATE, "forward chains took different numbers of range slots");
                const int nfl = g_ctx->rng.next2 * ABSMAX_SLOTS;
                hipLaunchKernelGGL(range_merge_kernel, dim3((nfl + 255) / 256), dim3(256), 0, s, g_ctx->rng.pool,
                                   g_ctx->rng.pool + (size_t)RANGE_FWD * ABSMAX_SLOTS, nfl);
            }
        }
    }
    for (size_t ui = 0; ui < net->units.size(); ++ui) {
        const Unit& u = net->units[ui];
        h = (h + u.stride - 1) / u.stride; w = (w + u.stride - 1) / u.stride;
        xin = F(pl.xo[ui]);
    }
    // heads: pointwise GEMM on the cell kernels + gather of the four taps (as the inference engine) when the feature map's range
    // and the pointwise cells exist, else the 2x2-conv form on the fp32 kernel
    static const bool head_pw = (dgp_tune("DGP_HEAD_PW", 1) != 0);
    auto head_forward = [&](const ConvLayer& hd, int li, int njt, float* out) -> hipError_t {
        const float* rin = range_of(xin);
        const float* rw = tr->d_wrng ? tr->d_wrng + (size_t)li * ABSMAX_SLOTS : nullptr;
        if (fast && !(head_pw && rin && rw && hd.d_wh3_pw && tr->d_h3_table && (FMT == 1 || hd.d_wh1_pw))) return hipErrorInvalidValue;       // (H2 / H1 features: cell kernels only)
        if (!(head_pw && g_ctx->rng.on && rin && rw && hd.d_wh3_pw && tr->d_h3_table))
            return conv_launch(hd, hd.d_w, hd.nk, hd.CoutP, xin, B, h, w, hd.Cin, 1, 1, h, w, hd.Cout, 1, 0, nullptr, hd.d_bias,
                               nullptr, 0, 0, 0, nullptr, false, 1, njt, out, s);
        float* T = F(pl.g0);                       // gradient scratch: free during the forward pass
        ConvArgs a{};
        a.in = xin; a.wpk = hd.d_w_pw; a.wh3 = hd.d_wh3_pw; a.out = T; a.in_absmax = rin; a.w_absmax = rw;
        if (fast) {
            a.in_fmt = FMT; a.in_scale_dev = range_prev_of(rin);
            if (!a.in_scale_dev) return hipErrorInvalidValue;
            if (FMT == 2) a.wh3 = hd.d_wh1_pw;
        }
        a.slab = g_ctx->tail_slab; a.slab_bytes = g_ctx->tail_slab ? (unsigned)(TAIL_SLAB_FLOATS * sizeof(float)) : 0u;
        a.N = B; a.H = h; a.W = w; a.Cin = hd.Cin; a.log2cin4 = ilog2(hd.Cin / 4);
        a.Ho = h; a.Wo = w; a.Cout = hd.coutp_pw; a.CoutP = hd.coutp_pw;
        a.KH = 1; a.KW = 1; a.stride = 1; a.dil = 1; a.ntaps = 1; a.nk = nk_for(1, 1, hd.Cin); a.M = B * h * w;
        a.in_bytes = (unsigned)((size_t)a.M * hd.Cin * 4); a.out_bytes = (unsigned)((size_t)a.M * hd.coutp_pw * 4);
        a.w_bytes = (unsigned)((size_t)a.nk * 8 * hd.coutp_pw * 16); a.wh3_bytes = a.w_bytes;
        hipError_t e = launch_conv(a, pick_tile(a.M, a.CoutP, a.nk * BK, true), s);
        if (e != hipSuccess) return e;
        return launch_head_gather(T, hd.d_bias, B, h, w, njt, hd.coutp_pw, out, s);
    };
    TRY_HIP(head_forward(net->layers[net->head_part], net->head_part, d.num_joints, F(pl.scmap)));
    TRY_HIP(head_forward(net->layers[net->head_locref], net->head_locref, 2 * d.num_joints, F(pl.locref)));
    if (fast) {
        // every H2 tensor of this pass against its predicted scale (the scoremaps are garbage when one failed: the host repeats the step)
        H2CheckList cl{};
        if (g_ctx->h2_slots.size() > sizeof(cl.idx) / sizeof(cl.idx[0])) return fail(DGP_ERR_STATE, "fast pass: too many H2 tensors");
        {   // ... and the fp16 copy of block1's output that the first H2 unit read
            const float* pv = range_prev_of(range_of(ub == 0 ? F(pl.pool) : F(pl.xo[ub - 1])));
            if (!pv) return fail(DGP_ERR_STATE, "fast pass: the first H2 unit's input has no range");
            g_ctx->h2_slots.push_back((int)((pv - g_ctx->rng.prev) / ABSMAX_SLOTS));
        }
        cl.n = (int)g_ctx->h2_slots.size();
        for (int k = 0; k < cl.n; ++k) cl.idx[k] = (short)g_ctx->h2_slots[k];
        hipLaunchKernelGGL(h2_pred_check_kernel, dim3(cl.n), dim3(64), 0, s, cl, g_ctx->rng.pool, g_ctx->rng.prev, tr->d_fast_flag);
        // fp32 copy of the features for the heads' backward (not when it reads the H1 features in place: merged heads of the 16-bit tier)
        const float* rfeat = range_of(xin);
        const long long n8 = (long long)B * h * w * net->units.back().depth / 8;
        tr->fwd_feat32 = !(FMT == 2 && tr->d_hmT && tr->d_hmT_h1);
        if (!tr->fwd_feat32) {
        } else if (FMT == 2)
            hipLaunchKernelGGL(h1_to_f32_pred_kernel, dim3(grid_for(n8)), dim3(256), 0, s, reinterpret_cast<const uint4*>(xin), n8,
                               range_prev_of(rfeat), reinterpret_cast<float4*>(F(pl.feat32)), (const uint4*)nullptr);
        else
        hipLaunchKernelGGL(h2_to_f32_pred_kernel, dim3(grid_for(n8)), dim3(256), 0, s, reinterpret_cast<const uint4*>(xin), n8,
                           range_prev_of(rfeat), reinterpret_cast<float4*>(F(pl.feat32)));
        TRY_HIP(hipGetLastError());
    }
    if (scmap) *scmap = F(pl.scmap);
    if (locref) *locref = F(pl.locref);
    return DGP_OK;
}

// One conv layer's parameter gradients: dWraw = A^T dY, d beta = colsum(dY), then the BN-affine algebra.
// Deferred finalisation (default; DGP_WGRAD_DEFER=0: per layer as before): every non-head layer accumulates dWraw / colsum in its
// own region of the workspace (zeroed once per pass) and two launches at the end of dgp_train_backward turn them into dW, d gamma,
// d beta for all layers -- ~150 dispatches fewer per step than fill + fill + wgrad + scale + bn per layer.

static int layer_param_grads(dgp_trainer* tr, size_t li, const float* x, int N, int H, int W, const float* dy, int Ho,
                             int Wo, int stride, int pad_t, int pad_l, float* dwraw, float* colsum, hipStream_t s) {
    dgp_net* net = tr->net;
    const ConvLayer& l = net->layers[li];
    const TLayer& t = tr->tl[li];
    if (g_ctx->defer_plan) {
        hipStream_t ws_ = s;
        hipEvent_t done = nullptr;
        if (g_ctx->overlap) {                    // behind everything enqueued on the chain's stream so far (dY's producer included)
            hipEvent_t ready = g_ctx->take_event();
            done = g_ctx->take_event();
            if (!ready || !done) return fail(DGP_ERR_HIP, "weight-gradient stream: hipEventCreate failed");
            TRY_HIP(hipEventRecord(ready, s));
            TRY_HIP(hipStreamWaitEvent(g_ctx->s2, ready, 0));
            ws_ = g_ctx->s2;
        }
        const void *xs = nullptr, *dys = nullptr;
        const float *xp = nullptr, *dyp = nullptr;
        {
            const auto px = g_ctx->shadow_prev.find(x), py = g_ctx->shadow_prev.find(dy);
            const auto bx = g_ctx->shadow_base.find(x), by = g_ctx->shadow_base.find(dy);
            if (px != g_ctx->shadow_prev.end() && py != g_ctx->shadow_prev.end() && bx != g_ctx->shadow_base.end() && by != g_ctx->shadow_base.end()) {
                xs = bx->second; xp = px->second; dys = by->second; dyp = py->second;
            }
        }
        int* h2_only = (xs && xs == (const void*)x) ? tr->d_fast_flag : nullptr;       // fast pass: the activation has no fp32 twin
        const bool h1 = tr->fwd_fast && tr->fwd_fmt == 2 && xs && dys;                 // 16-bit tier: both operands are H1 tensors
        if (h1) h2_only = tr->d_fast_flag;
        if (!h2_only && tr->fwd_fast && g_ctx->shadow_base.count(x) && g_ctx->shadow_base[x] == x)
            return fail(DGP_ERR_STATE, "weight gradient of an H2-only activation without a usable gradient copy");
        TRY_HIP(wgrad_launch(x, N, H, W, l.Cin, dy, Ho, Wo, l.Cout, l.KH, l.KW, stride, l.rate, pad_t, pad_l,
                             reinterpret_cast<float*>(g_ctx->defer_ws + ((const TPlan*)g_ctx->defer_plan)->dw_l[li]),
                             reinterpret_cast<float*>(g_ctx->defer_ws + ((const TPlan*)g_ctx->defer_plan)->cs_l[li]), ws_, true,
                             nullptr, nullptr, xs, xp, dys, dyp, h2_only, h1));
        if (done) {
            TRY_HIP(hipEventRecord(done, g_ctx->s2));
            g_ctx->readers.emplace((const void*)dy, done);
        }
        return DGP_OK;
    }
    TRY_HIP(wgrad_launch(x, N, H, W, l.Cin, dy, Ho, Wo, l.Cout, l.KH, l.KW, stride, l.rate, pad_t, pad_l, dwraw, colsum, s));
    float* dot = colsum + l.Cout;
    const int krows = l.KH * l.KW * t.cin_real;
    const int rpb = 64;
    hipLaunchKernelGGL(scale_dw_dot_kernel, dim3((l.Cout + 63) / 64, (krows + rpb - 1) / rpb), dim3(256), 0, s, dwraw,
                       tr->params + t.w_off, l.d_scale, l.KH * l.KW, l.Cin, t.cin_real, l.Cout, rpb, tr->grads + t.w_off, dot);
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3((l.Cout + 127) / 128), dim3(128), 0, s, dot, colsum, tr->stats + t.mean_off,
                       tr->stats + t.var_off, net->desc.bn_eps, l.Cout, tr->grads + t.g_off, tr->grads + t.b_off);
    TRY_HIP(hipGetLastError());
    return DGP_OK;
}

int dgp_train_backward(dgp_trainer* tr, int32_t nt, void* workspace, size_t workspace_bytes, const float* dscmap,
                       const float* dlocref, void* stream) {
    if (tr) g_ctx = &tr->ctx;
    if (!tr || !workspace || !dscmap || !dlocref) return fail(DGP_ERR_INVALID, "dgp_train_backward: null argument");
    dgp_net* net = tr->net;
    const TPlan pl = make_tplan(tr, nt);
    if (workspace_bytes < pl.total) return fail(DGP_ERR_INVALID, "dgp_train_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    g_ctx->tail_slab = F(pl.tail);
    range_pass_begin(tr, s, true);
    if (pl.sh_g0) {
        const size_t gb[6][2] = {{pl.g0, pl.sh_g0}, {pl.g1, pl.sh_g1}, {pl.dr1, pl.sh_dr1}, {pl.dr2, pl.sh_dr2}, {pl.dr1_b, pl.sh_dr1_b}, {pl.dr2_b, pl.sh_dr2_b}};
        for (const auto& q : gb) { g_ctx->shadow_base[F(q[0])] = F(q[1]); g_ctx->shadow_prev.erase(F(q[0])); }
    }
    struct WantReset { ~WantReset() { g_shadow_want = true; } } want_reset;
    const dgp_net_desc& d = net->desc;
    const int B = nt, nj = d.num_joints;
    const int nu = (int)net->units.size();
    static const bool defer_env = (dgp_tune("DGP_WGRAD_DEFER", 1) != 0);
    g_ctx->defer_plan = defer_env ? &pl : nullptr;
    g_ctx->defer_ws = ws;
    if (defer_env) TRY_HIP(hipMemsetAsync(ws + pl.dwall, 0, pl.dwall_bytes, s));
    // weight gradients on their own stream beside the data-gradient chain (DGP_WGRAD_OVERLAP=0: one stream, A/B)
    static const bool overlap_env = (dgp_tune("DGP_WGRAD_OVERLAP", 1) != 0);
    TrainCtx* const ctx = g_ctx;
    if (defer_env && overlap_env) {
        if (!ctx->s2) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);         // lo: numerically greatest = least urgent
            static const int prio_mode = dgp_tune("DGP_WGRAD_PRIO", 0);      // 0 least urgent, 1 default, 2 most urgent
            TRY_HIP(hipStreamCreateWithPriority(&ctx->s2, hipStreamNonBlocking, prio_mode == 0 ? lo : prio_mode == 2 ? hi : 0));
        }
        ctx->overlap = true;
        ctx->ev_next = 0;
        ctx->readers.clear();
    }
    // join: the chain's stream waits for every weight gradient (before the finalisation launches, and on every exit path)
    struct Join {
        TrainCtx* c; hipStream_t s;
        void operator()() {
            if (!c->overlap) return;
            c->overlap = false;
            c->readers.clear();
            hipEvent_t e = c->take_event();
            if (e && hipEventRecord(e, c->s2) == hipSuccess) (void)hipStreamWaitEvent(s, e, 0);
            else (void)hipStreamSynchronize(c->s2);
        }
        ~Join() { (*this)(); }
    } join{ctx, s};
    // the chain is about to overwrite `buf`: wait for the weight-gradient launches that still read it
    auto before_write = [&](const void* buf) -> hipError_t {
        if (!ctx->overlap) return hipSuccess;
        auto range = ctx->readers.equal_range(buf);
        for (auto it = range.first; it != range.second; ++it) {
            hipError_t e2 = hipStreamWaitEvent(s, it->second, 0);
            if (e2 != hipSuccess) return e2;
        }
        ctx->readers.erase(range.first, range.second);
        return hipSuccess;
    };
    // geometry per unit input
    std::vector<int> hs(nu + 1), wsz(nu + 1);
    hs[0] = net->hp; wsz[0] = net->wp;
    for (int ui = 0; ui < nu; ++ui) {
        const Unit& u = net->units[ui];
        hs[ui + 1] = (hs[ui] + u.stride - 1) / u.stride;
        wsz[ui + 1] = (wsz[ui] + u.stride - 1) / u.stride;
    }
    const int fh = hs[nu], fw = wsz[nu];
    float* dwraw = F(pl.dwraw);
    float* colsum = F(pl.colsum);
    float* G[2] = {F(pl.g0), F(pl.g1)};
    int cur = 0;
    int rc;

    // fast pass (dgp_train_forward): the activations of units >= ub are H2 tensors -- gates read them as such, weight gradients read
    // them in place, and the heads use the fp32 copy of the features the forward pass left in feat32
    const bool fast = tr->fwd_fast;
    const bool h1p = fast && tr->fwd_fmt == 2;          // 16-bit tier: the gradient tensors of units >= ub are H1-only as well
    if (h1p && !defer_env) return fail(DGP_ERR_STATE, "the 16-bit tier needs the deferred weight-gradient finalisation (DGP_WGRAD_DEFER)");
    int ub = nu;
    for (int ui = 1; ui < nu; ++ui)
        if (net->units[ui].depth_bn >= 128 && pl.sh_xo[ui - 1]) { ub = ui; break; }
    if (h1p && pl.sh_pool && pool_idx_on()) ub = 0;      // (as dgp_train_forward decided)
    g_h2 = H2Launch();
    g_shadow_fmt = h1p ? 2 : 1;
    if (h1p) g_ctx->h2_slots.clear();                   // this pass's H1 gradient tensors, checked against their predicted scales at its end
    float* GH[2] = {F(pl.sh_g0), F(pl.sh_g1)};           // tier 1: G of the H1 units (the fp16-copy regions of the parity pass hold the tensors themselves)
    // H1 tensor `t` takes the range slot / predicted scale of the launch that wrote `src` (a converted copy of it)
    auto adopt_h1 = [&](const void* t, const void* src) -> int {
        const float* slot = range_of(src);
        const float* pv = range_prev_of(slot);
        if (!slot || !pv) return fail(DGP_ERR_STATE, "16-bit tier: a gradient tensor has no predicted range");
        range_set(t, slot);
        g_ctx->shadow_base[t] = const_cast<float*>(static_cast<const float*>(t));
        g_ctx->shadow_prev[t] = pv;
        const int idx = (int)((pv - g_ctx->rng.prev) / ABSMAX_SLOTS);
        if (std::find(g_ctx->h2_slots.begin(), g_ctx->h2_slots.end(), idx) == g_ctx->h2_slots.end()) g_ctx->h2_slots.push_back(idx);
        return DGP_OK;
    };
    // ---- heads: gather phases, parameter grads, data grad into G[cur] (gated by the last unit's ReLU)
    const float* feat = fast ? F(pl.feat32) : F(pl.xo[nu - 1]);
    // range of the gathered loss gradients (both heads): the first slot of every backward pass -- it predicts the scale of the merged
    // H1 tensor of the next 16-bit pass
    float* const slot_dph = range_take();
    const bool heads_h1 = h1p && tr->d_hmT && tr->d_hmT_h1 && slot_dph && range_prev_of(slot_dph) && !tr->fwd_feat32;
    if (fast && !heads_h1 && !tr->fwd_feat32) return fail(DGP_ERR_STATE, "backward: the forward pass left no fp32 features for the heads");
    if (heads_h1) {
        // ---- 16-bit tier: both heads at once.  Their loss gradients are gathered into ONE H1 tensor (scale predicted from its range one
        // step ago); ONE weight-gradient launch reads it and the H1 features in place (second stream), ONE H1 -> H1 data-gradient launch
        // with the merged panel writes d features, gated by the features' ReLU, as the H1 tensor the last unit's backward reads.
        const ConvLayer &l0 = net->layers[net->head_part], &l1 = net->layers[net->head_locref];
        const TLayer &t0 = tr->tl[net->head_part], &t1 = tr->tl[net->head_locref];
        const int CT = tr->hm_ct, njt0 = l0.Cout / 4, njt1 = l1.Cout / 4;
        float* const DPH = F(pl.dphh);
        const float* dprev = range_prev_of(slot_dph);
        const float* featH = F(pl.xo[nu - 1]);
        const auto fp = g_ctx->shadow_prev.find(featH);
        if (fp == g_ctx->shadow_prev.end()) return fail(DGP_ERR_STATE, "16-bit tier: the features have no predicted range");
        hipLaunchKernelGGL(head_gather_h1_kernel, dim3(grid_for((long long)B * fh * fw * (CT / 8))), dim3(256), 0, s, dscmap, dlocref, B, fh, fw,
                           njt0, t0.cpad, njt1, t1.cpad, CT, dprev, reinterpret_cast<uint4*>(DPH), slot_dph);
        range_set(DPH, slot_dph);
        g_ctx->shadow_base[DPH] = DPH;
        g_ctx->shadow_prev[DPH] = dprev;
        g_ctx->h2_slots.push_back((int)((dprev - g_ctx->rng.prev) / ABSMAX_SLOTS));
        hipStream_t hs_ = s;
        if (ctx->overlap) {
            hipEvent_t ready = ctx->take_event();
            if (!ready) return fail(DGP_ERR_HIP, "weight-gradient stream: hipEventCreate failed");
            TRY_HIP(hipEventRecord(ready, s));
            TRY_HIP(hipStreamWaitEvent(ctx->s2, ready, 0));
            hs_ = ctx->s2;
        }
        TRY_HIP(wgrad_launch(featH, B, fh, fw, l0.Cin, DPH, fh, fw, CT, 2, 2, 1, 1, 1, 1, dwraw, colsum, hs_, false, nullptr, nullptr, featH,
                             fp->second, DPH, dprev, tr->d_fast_flag, true));
        hipLaunchKernelGGL(finalize_head_grads, dim3(grid_for(9ll * njt0 * l0.Cin)), dim3(256), 0, hs_, dwraw, colsum, njt0, l0.Cin, CT, 0,
                           tr->grads + t0.w_off, tr->grads + t0.b_off);
        hipLaunchKernelGGL(finalize_head_grads, dim3(grid_for(9ll * njt1 * l1.Cin)), dim3(256), 0, hs_, dwraw, colsum, njt1, l1.Cin, CT, t0.cpad,
                           tr->grads + t1.w_off, tr->grads + t1.b_off);
        (void)range_take();                      // (the slot the first head's launch takes in a plain pass: every pass takes its slots in one order)
        ConvLayer lt = l0;
        lt.KH = lt.KW = 2; lt.rate = 1;
        g_h2 = H2Launch();
        g_h2.in_fmt = 2; g_h2.out_fmt = 2; g_h2.mask_fmt = 2;
        TRY_HIP(conv_launch(lt, tr->d_hmT, tr->hm_nk, t0.cinP, DPH, B, fh, fw, CT, 0, 0, fh, fw, l0.Cin, 1, 0, nullptr, nullptr, nullptr, 0, 0, 0,
                            featH, false, 0, 0, GH[cur], s));
    } else {
        // the per-head data-gradient panels (t.d_wT) are parity-only panels: a 16-bit pass that cannot merge its heads (DGP_TRAIN_HEADS_H1=0,
        // head widths the merged panel does not take) lands here after lazy syncs and must not read last step's weights
        if (tr->parity_stale && (rc = refresh_parity_panels(tr, s))) return rc;
        const size_t heads[2] = {(size_t)net->head_part, (size_t)net->head_locref};
        const float* dsrc[2] = {dscmap, dlocref};
        float* dph[2] = {F(pl.dph0), F(pl.dph1)};
        for (int k = 0; k < 2; ++k) {
            const ConvLayer& l = net->layers[heads[k]];
            const TLayer& t = tr->tl[heads[k]];
            const int njt = l.Cout / 4;
            const long long tot = (long long)B * fh * fw * t.cpad;
            hipLaunchKernelGGL(head_gather_kernel, dim3(grid_for(tot)), dim3(256), 0, s, dsrc[k], B, fh, fw, njt, t.cpad, dph[k], slot_dph);
            // the head's parameter gradients (fill + wgrad + finalise through the shared dwraw / colsum scratch, in stream order) go to the
            // second stream when the pass overlaps: the chain only needs dph[k] for the data gradient below
            hipStream_t hs_ = s;
            if (ctx->overlap) {
                hipEvent_t ready = ctx->take_event();
                if (!ready) return fail(DGP_ERR_HIP, "weight-gradient stream: hipEventCreate failed");
                TRY_HIP(hipEventRecord(ready, s));
                TRY_HIP(hipStreamWaitEvent(ctx->s2, ready, 0));
                hs_ = ctx->s2;
            }
            TRY_HIP(wgrad_launch(feat, B, fh, fw, l.Cin, dph[k], fh, fw, t.cpad, 2, 2, 1, 1, 1, 1, dwraw, colsum, hs_));
            hipLaunchKernelGGL(finalize_head_grads, dim3(grid_for(9ll * njt * l.Cin)), dim3(256), 0, hs_, dwraw, colsum, njt,
                               l.Cin, t.cpad, 0, tr->grads + t.w_off, tr->grads + t.b_off);
            // dfeat (+)= convT: 2x2 taps flipped, pad' = 0; second head accumulates onto the first; gate on the last
            ConvLayer lt = l;
            lt.KH = lt.KW = 2; lt.rate = 1;
            TRY_HIP(conv_launch(lt, t.d_wT, t.nkT, t.cinP, dph[k], B, fh, fw, t.cpad, 0, 0, fh, fw, l.Cin, 1, 0, nullptr, nullptr,
                                k == 1 ? G[cur] : nullptr, 1, fh, fw, k == 1 ? feat : nullptr, false, 0, 0, G[cur], s));
        }
    }

    if (heads_h1) {
        // (the merged launch wrote the H1 tensor itself)
    } else if (h1p) {
        // the heads' data gradient (fp32, from the kernels of the parity path) enters the H1 units as an H1 tensor
        if ((rc = adopt_h1(GH[cur], G[cur]))) return rc;
        const long long n8 = (long long)B * fh * fw * net->units[nu - 1].depth / 8;
        hipLaunchKernelGGL(f32_to_h1_pred_kernel, dim3(grid_for(n8)), dim3(256), 0, s, reinterpret_cast<const float4*>(G[cur]), n8,
                           g_ctx->shadow_prev[GH[cur]], reinterpret_cast<uint4*>(GH[cur]));
    } else if (fast) {
        // the heads' data gradient came from a kernel that writes no fp16 copy, and the last unit's conv3 weight gradient can only read
        // its H2 activation through the LDS-DMA tile: make the copy here
        const float* pv = range_prev_of(range_of(G[cur]));
        const auto sb = g_ctx->shadow_base.find(G[cur]);
        if (!pv || sb == g_ctx->shadow_base.end()) return fail(DGP_ERR_STATE, "fast pass: no range for the heads' data gradient");
        const long long n8 = (long long)B * fh * fw * net->units[nu - 1].depth / 8;
        hipLaunchKernelGGL(f32_to_shadow_kernel, dim3(grid_for(n8)), dim3(256), 0, s, reinterpret_cast<const float4*>(G[cur]), n8, pv,
                           reinterpret_cast<uint4*>(sb->second));
        g_ctx->shadow_prev[G[cur]] = pv;
    }
    // ---- bottleneck units, last to first.  G[cur] = d loss / d (unit output), already gated by its ReLU.
    int stop_after = -1;
#ifdef DGP_TUNING
    if (const char* e = getenv("DGP_BWD_STOP")) stop_after = atoi(e);      // debugging aid (tuning builds): leave G of an inner unit in place
#endif
    // 16-bit tier with the fused root block: the stem's weight gradient reads d pool as the H1 tensor unit 0 leaves (stem_wgrad_h1_kernel:
    // pool backward fused, no d conv1 map); A/B switch DGP_TRAIN_STEM_WGRAD_H1=0
    static const bool stem_wgrad_env = (dgp_tune("DGP_TRAIN_STEM_WGRAD_H1", 1) != 0);
    const bool stem_wgrad_h1 = stem_wgrad_env && h1p && ub == 0 && tr->fwd_stem_fused && pool_idx_on() && g_ctx->defer_plan &&
                               net->layers[net->conv1].Cout == 64 && net->layers[net->conv1].KH == 7 && net->layers[net->conv1].Cin == 4;
    const float* stem_g = nullptr;
    const float* stem_g_prev = nullptr;
    // deferred finalisation: table of every non-head layer, conv1 LAST (its weight gradient is the last launch of the pass: the other
    // layers are finalised beside it)
    int max_cout = 0, max_krows = 0;
    if (g_ctx->defer_plan) {
        if (!tr->d_fin_table || tr->fin_B != B || tr->fin_h != d.in_h || tr->fin_w != d.in_w) {      // offsets follow the plan
            std::vector<FinDesc> tab;
            tr->fin_of_layer.assign(net->layers.size(), -1);
            auto entry = [&](size_t li) {
                tr->fin_of_layer[li] = (int)tab.size();
                const ConvLayer& l = net->layers[li];
                const TLayer& t = tr->tl[li];
                FinDesc f{};
                f.dw_off = (long long)pl.dw_l[li]; f.cs_off = (long long)pl.cs_l[li];
                f.w_off = t.w_off; f.g_off = t.g_off; f.b_off = t.b_off; f.mean_off = t.mean_off; f.var_off = t.var_off;
                f.taps = l.KH * l.KW; f.cin = l.Cin; f.cin_real = t.cin_real; f.cout = l.Cout; f.d_scale = l.d_scale;
                tab.push_back(f);
            };
            for (size_t li = 0; li < net->layers.size(); ++li)
                if ((int)li != net->head_part && (int)li != net->head_locref && (int)li != net->conv1) entry(li);
            entry((size_t)net->conv1);
            if (!tr->d_fin_table) TRY_HIP(hipMalloc(&tr->d_fin_table, tab.size() * sizeof(FinDesc)));
            TRY_HIP(hipStreamSynchronize(s));        // (a previous pass may still read the old table)
            TRY_HIP(hipMemcpy(tr->d_fin_table, tab.data(), tab.size() * sizeof(FinDesc), hipMemcpyHostToDevice));
            tr->n_fin = (int)tab.size(); tr->fin_B = B; tr->fin_h = d.in_h; tr->fin_w = d.in_w;
        }
        for (size_t li = 0; li < net->layers.size(); ++li) {
            if ((int)li == net->head_part || (int)li == net->head_locref) continue;
            max_cout = std::max(max_cout, net->layers[li].Cout);
            max_krows = std::max(max_krows, net->layers[li].KH * net->layers[li].KW * tr->tl[li].cin_real);
        }
    }
    const int rpb = 128;
    const FinDesc* fin_tab = reinterpret_cast<const FinDesc*>(tr->d_fin_table);
    auto finalise = [&](int first, int count, hipStream_t st) {
        if (count <= 0) return;
        hipLaunchKernelGGL(scale_dw_dot_all_kernel, dim3((max_cout + 63) / 64, (max_krows + rpb - 1) / rpb, (unsigned)count), dim3(256), 0, st,
                           fin_tab + first, (const char*)ws, tr->params, tr->grads, rpb);
    };
    auto bn_grads = [&](int first, int count, hipStream_t st) {
        if (count <= 0) return;
        hipLaunchKernelGGL(bn_param_grads_all_kernel, dim3((max_cout + 127) / 128, (unsigned)count), dim3(128), 0, st, fin_tab + first,
                           (const char*)ws, tr->stats, d.bn_eps, tr->grads);
    };
    // ---- gradient groups (see dgp_trainer::GradGroup): cut the units, back to front, into runs of >= a quarter of the parameters each
    if (tr->groups.empty()) {
        auto layer_range = [&](int li, long long& lo, long long& hi) {
            if (li < 0) return;
            const ConvLayer& l = net->layers[li];
            const TLayer& t = tr->tl[li];
            const bool head = (li == net->head_part || li == net->head_locref);
            const long long wsz_ = head ? 9ll * (l.Cout / 4) * l.Cin : (long long)l.KH * l.KW * t.cin_real * l.Cout;
            const long long bsz_ = head ? l.Cout / 4 : l.Cout;
            auto acc = [&](long long off, long long n) { lo = std::min(lo, off); hi = std::max(hi, off + (n + 3) / 4 * 4); };
            acc(t.w_off, wsz_);
            if (!head) acc(t.g_off, bsz_);
            acc(t.b_off, bsz_);
        };
        std::vector<dgp_trainer::GradGroup> gs;
        const bool can = g_ctx->defer_plan != nullptr && !tr->fin_of_layer.empty();
        if (can) {
            dgp_trainer::GradGroup cur_g;
            long long lo = tr->n_train, hi = 0;
            int f_lo = tr->n_fin, f_n = 0;
            layer_range(net->head_part, lo, hi);
            layer_range(net->head_locref, lo, hi);
            for (int ui = nu - 1; ui >= 1; --ui) {
                const Unit& u = net->units[ui];
                for (int li : {u.sc, u.c1, u.c2, u.c3}) {
                    if (li < 0) continue;
                    layer_range(li, lo, hi);
                    f_lo = std::min(f_lo, tr->fin_of_layer[li]); ++f_n;
                }
                if (hi - lo >= tr->n_train / 4 && gs.size() < 6) {
                    cur_g.cut_ui = ui; cur_g.fin_first = f_lo; cur_g.fin_count = f_n; cur_g.lo = lo; cur_g.hi = hi;
                    gs.push_back(cur_g);
                    lo = tr->n_train; hi = 0; f_lo = tr->n_fin; f_n = 0;
                }
            }
            // the last group: everything in front of the last cut (its table rows are [0, first cut row) + the stem's row, finalised at the end)
            dgp_trainer::GradGroup last;
            last.cut_ui = -1; last.lo = 0; last.hi = gs.empty() ? tr->n_train : gs.back().lo;
            last.fin_first = 0; last.fin_count = gs.empty() ? tr->n_fin - 1 : gs.back().fin_first;
            bool ok = true;                              // the groups must tile the flat buffer and the table, back to front
            long long expect_hi = tr->n_train;
            int expect_f = tr->n_fin - 1;
            for (const auto& g : gs) {
                ok = ok && g.hi == expect_hi && g.fin_first + g.fin_count == expect_f && g.lo < g.hi;
                expect_hi = g.lo; expect_f = g.fin_first;
            }
            if (!ok) gs.clear(), last.hi = tr->n_train, last.fin_count = tr->n_fin - 1;
            gs.push_back(last);
        } else {
            dgp_trainer::GradGroup all;
            all.cut_ui = -1; all.lo = 0; all.hi = tr->n_train;
            gs.push_back(all);
        }
        for (auto& g : gs)
            if (hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) return fail(DGP_ERR_HIP, "gradient groups: hipEventCreate failed");
        tr->groups = gs;
    }
    size_t grp_next = 0;                                 // next group to complete
    // a group is complete behind the weight gradients of its layers: finalise it on their stream and record its event
    auto close_groups_at = [&](int ui) -> hipError_t {
        while (grp_next + 1 < tr->groups.size() && tr->groups[grp_next].cut_ui == ui) {
            auto& g = tr->groups[grp_next++];
            hipStream_t st = ctx->overlap ? ctx->s2 : s;
            finalise(g.fin_first, g.fin_count, st);
            bn_grads(g.fin_first, g.fin_count, st);
            hipError_t e2 = hipEventRecord(g.ev, st);
            if (e2 != hipSuccess) return e2;
        }
        return hipSuccess;
    };
    for (int ui = nu - 1; ui >= 0; --ui) {
        if (stop_after >= 0 && (nu - 1 - ui) >= stop_after) {
            if (const char* e2 = getenv("DGP_BWD_DUMP")) {
                (void)hipStreamSynchronize(s);
                const size_t n = (size_t)B * hs[ui + 1] * wsz[ui + 1] * net->units[ui].depth;
                std::vector<float> hbuf(n);
                (void)hipMemcpy(hbuf.data(), G[cur], n * sizeof(float), hipMemcpyDeviceToHost);
                FILE* f = fopen(e2, "wb");
                if (f) { fwrite(hbuf.data(), sizeof(float), n, f); fclose(f); }
                (void)hipMemcpy(hbuf.data(), F(pl.xo[ui]), n * sizeof(float), hipMemcpyDeviceToHost);
                f = fopen((std::string(e2) + ".x").c_str(), "wb");
                if (f) { fwrite(hbuf.data(), sizeof(float), n, f); fclose(f); }
            }
            return DGP_OK;
        }
        const Unit& u = net->units[ui];
        const int h = hs[ui], w = wsz[ui], ho = hs[ui + 1], wo = wsz[ui + 1];
        const float* xin = ui == 0 ? F(pl.pool) : F(pl.xo[ui - 1]);
        if (h1p && ui >= ub) {
            // ---- 16-bit tier: every tensor of this unit is an H1 tensor with a predicted scale; data gradients are H1 -> H1 launches of the
            // cell kernels (gate and residual read as H1), weight gradients read both operands in place (wgrad_dma_h1)
            const ConvLayer &l1 = net->layers[u.c1], &l2 = net->layers[u.c2], &l3 = net->layers[u.c3];
            const TLayer &t1 = tr->tl[u.c1], &t2 = tr->tl[u.c2], &t3 = tr->tl[u.c3];
            float* const GoutH = GH[cur];
            float* const GinH = GH[cur ^ 1];
            float* const DR2H = F((ui & 1) ? pl.sh_dr2_b : pl.sh_dr2);
            float* const DR1H = F((ui & 1) ? pl.sh_dr1_b : pl.sh_dr1);
            float* const DXAH = F((ui & 1) ? pl.dxa_b : pl.dxa);
            const float* xinH = ui > ub ? xin : F(ui == 0 ? pl.sh_pool : pl.sh_xo[ui - 1]);       // (unit ub reads the H1 copy of its fp32 input)
            auto fmt = [&](bool gate, const void* res_key) {
                g_h2 = H2Launch();
                g_h2.in_fmt = 2; g_h2.out_fmt = 2; g_h2.mask_fmt = gate ? 2 : 0;
                if (res_key) { g_h2.res_fmt = 2; g_h2.res_key = res_key; }
            };
            rc = layer_param_grads(tr, u.c3, F(pl.r2[ui]), B, ho, wo, GoutH, ho, wo, 1, 0, 0, dwraw, colsum, s);
            if (rc) return rc;
            TRY_HIP(before_write(DR2H));
            fmt(true, nullptr);
            TRY_HIP(conv_launch(l3, t3.d_wT, t3.nkT, t3.cinP, GoutH, B, ho, wo, l3.Cout, 0, 0, ho, wo, l3.Cin, 1, 0, nullptr, nullptr,
                                nullptr, 0, 0, 0, F(pl.r2[ui]), false, 0, 0, DR2H, s));
            const int pb_h = pad_before_for(h, 3, u.stride, u.rate, true), pb_w = pad_before_for(w, 3, u.stride, u.rate, true);
            rc = layer_param_grads(tr, u.c2, F(pl.r1[ui]), B, h, w, DR2H, ho, wo, u.stride, pb_h, pb_w, dwraw, colsum, s);
            if (rc) return rc;
            const int keff = 2 * u.rate + 1;
            TRY_HIP(before_write(DR1H));
            if (u.stride > 1) {
                // stride-2 conv2: its data gradient reads dR2 on the zero-stuffed grid; materialised (a few MB) so that it is a plain stride-1 launch
                if (u.stride != 2) return fail(DGP_ERR_STATE, "16-bit tier: stride > 2");
                float* const ZS = F(pl.dr1);             // (the fp32 dR1 region is free while the H1 units run)
                const size_t zbytes = (size_t)B * 2 * ho * 2 * wo * l2.Cout * 2;
                if (zbytes > (size_t)B * h * w * u.depth_bn * 4 + 0) { if (zbytes > (size_t)4 * ((size_t)B * h * w * u.depth_bn)) return fail(DGP_ERR_STATE, "16-bit tier: zero-stuffed gradient does not fit"); }
                TRY_HIP(before_write(ZS));
                TRY_HIP(hipMemsetAsync(ZS, 0, zbytes, s));
                const long long tot = (long long)B * ho * wo * (l2.Cout / 8);
                hipLaunchKernelGGL(h1_zero_stuff_kernel, dim3(grid_for(tot)), dim3(256), 0, s, reinterpret_cast<const uint4*>(DR2H), B, ho, wo,
                                   l2.Cout / 8, reinterpret_cast<uint4*>(ZS));
                fmt(true, nullptr);
                TRY_HIP(conv_launch(l2, t2.d_wT, t2.nkT, t2.cinP, ZS, B, 2 * ho, 2 * wo, l2.Cout, keff - 1 - pb_h, keff - 1 - pb_w, h, w,
                                    l2.Cin, 1, 0, nullptr, nullptr, nullptr, 0, 0, 0, F(pl.r1[ui]), false, 0, 0, DR1H, s, DR2H));
            } else {
                fmt(true, nullptr);
                TRY_HIP(conv_launch(l2, t2.d_wT, t2.nkT, t2.cinP, DR2H, B, ho, wo, l2.Cout, keff - 1 - pb_h, keff - 1 - pb_w, h, w,
                                    l2.Cin, 1, 0, nullptr, nullptr, nullptr, 0, 0, 0, F(pl.r1[ui]), false, 0, 0, DR1H, s));
            }
            const float* dxa = GoutH;
            int dxa_mode = 1, dxa_h = ho, dxa_w = wo;
            if (u.sc >= 0) {
                const ConvLayer& ls = net->layers[u.sc];
                const TLayer& ts = tr->tl[u.sc];
                if (u.stride != 1) return fail(DGP_ERR_STATE, "16-bit tier: strided shortcut conv");
                rc = layer_param_grads(tr, u.sc, xin, B, h, w, GoutH, ho, wo, 1, 0, 0, dwraw, colsum, s);
                if (rc) return rc;
                TRY_HIP(before_write(DXAH));
                fmt(false, nullptr);
                TRY_HIP(conv_launch(ls, ts.d_wT, ts.nkT, ts.cinP, GoutH, B, ho, wo, ls.Cout, 0, 0, h, w, ls.Cin, 1, 0, nullptr, nullptr,
                                    nullptr, 0, 0, 0, nullptr, false, 0, 0, DXAH, s));
                dxa = DXAH; dxa_h = h; dxa_w = w;
            } else if (u.stride > 1) {
                dxa_mode = -2;
            }
            rc = layer_param_grads(tr, u.c1, xin, B, h, w, DR1H, h, w, 1, 0, 0, dwraw, colsum, s);
            if (rc) return rc;
            TRY_HIP(before_write(GinH));
            fmt(true, dxa);
            TRY_HIP(conv_launch(l1, t1.d_wT, t1.nkT, t1.cinP, DR1H, B, h, w, l1.Cout, 0, 0, h, w, l1.Cin, 1, 0, nullptr, nullptr,
                                dxa, dxa_mode, dxa_h, dxa_w, xinH, false, 0, 0, GinH, s));
            if (ui == ub && ub == 0 && stem_wgrad_h1) {
                // (unit 0's data gradient stays H1: the fused stem weight-gradient kernel below reads it in place)
                stem_g = GinH;
                stem_g_prev = g_ctx->shadow_prev[GinH];
            } else if (ui == ub) {
                // the data gradient leaves the H1 units: fp32 copy for block1's kernels (same range slot: the epilogue tracked max |G| before rounding)
                float* Gf = G[cur ^ 1];
                TRY_HIP(before_write(Gf));
                const long long n8 = (long long)B * h * w * l1.Cin / 8;
                hipLaunchKernelGGL(h1_to_f32_pred_kernel, dim3(grid_for(n8)), dim3(256), 0, s, reinterpret_cast<const uint4*>(GinH), n8,
                                   g_ctx->shadow_prev[GinH], reinterpret_cast<float4*>(Gf), (const uint4*)nullptr);
                range_set(Gf, range_of(GinH));
                g_ctx->shadow_prev.erase(Gf);
            }
            cur ^= 1;
            TRY_HIP(close_groups_at(ui));
            continue;
        }
        float* Gout = G[cur];
        float* Gin = G[cur ^ 1];
        const ConvLayer &l1 = net->layers[u.c1], &l2 = net->layers[u.c2], &l3 = net->layers[u.c3];
        const TLayer &t1 = tr->tl[u.c1], &t2 = tr->tl[u.c2], &t3 = tr->tl[u.c3];
        // gradient buffers of this unit (units alternate between two sets: a weight gradient still reading unit ui + 1's dR1 / dR2 on
        // the second stream does not hold this unit's chain back)
        float* const DR2 = F((ui & 1) ? pl.dr2_b : pl.dr2);
        float* const DR1 = F((ui & 1) ? pl.dr1_b : pl.dr1);
        float* const DXA = F((ui & 1) ? pl.dxa_b : pl.dxa);
        // conv3: params, then dR2 = convT(G) gated by R2 > 0
        rc = layer_param_grads(tr, u.c3, F(pl.r2[ui]), B, ho, wo, Gout, ho, wo, 1, 0, 0, dwraw, colsum, s);
        if (rc) return rc;
        TRY_HIP(before_write(DR2));
        g_shadow_want = u.depth_bn >= 128;       // dR2 -> conv2's weight gradient (9 C1 x C1)
        g_h2.mask_fmt = (fast && ui >= ub) ? 1 : 0;
        TRY_HIP(conv_launch(l3, t3.d_wT, t3.nkT, t3.cinP, Gout, B, ho, wo, l3.Cout, 0, 0, ho, wo, l3.Cin, 1, 0, nullptr, nullptr,
                            nullptr, 0, 0, 0, F(pl.r2[ui]), false, 0, 0, DR2, s));
        // conv2: params, then dR1 = convT(dR2) gated by R1 > 0
        const int pb_h = pad_before_for(h, 3, u.stride, u.rate, true), pb_w = pad_before_for(w, 3, u.stride, u.rate, true);
        rc = layer_param_grads(tr, u.c2, F(pl.r1[ui]), B, h, w, DR2, ho, wo, u.stride, pb_h, pb_w, dwraw, colsum, s);
        if (rc) return rc;
        const int keff = 2 * u.rate + 1;
        TRY_HIP(before_write(DR1));
        g_shadow_want = u.depth_bn >= 128 && u.depth_in >= 128;      // dR1 -> conv1's weight gradient (Cin x C1)
        g_h2.mask_fmt = (fast && ui >= ub) ? 1 : 0;
        TRY_HIP(conv_launch(l2, t2.d_wT, t2.nkT, t2.cinP, DR2, B, ho, wo, l2.Cout, keff - 1 - pb_h, keff - 1 - pb_w, h, w,
                            l2.Cin, 1, u.stride > 1 ? u.stride : 0, nullptr, nullptr, nullptr, 0, 0, 0, F(pl.r1[ui]), false, 0, 0,
                            DR1, s));
        // shortcut branch
        const float* dxa = Gout;
        int dxa_mode = 1;                       // same grid
        int dxa_h = ho, dxa_w = wo;
        if (u.sc >= 0) {
            const ConvLayer& ls = net->layers[u.sc];
            const TLayer& ts = tr->tl[u.sc];
            rc = layer_param_grads(tr, u.sc, xin, B, h, w, Gout, ho, wo, u.stride, 0, 0, dwraw, colsum, s);
            if (rc) return rc;
            TRY_HIP(before_write(DXA));
            g_shadow_want = false;                // (dXa is only added to the next data gradient)
            TRY_HIP(conv_launch(ls, ts.d_wT, ts.nkT, ts.cinP, Gout, B, ho, wo, ls.Cout, 0, 0, h, w, ls.Cin, 1,
                                u.stride > 1 ? u.stride : 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, false, 0, 0, DXA, s));
            dxa = DXA; dxa_h = h; dxa_w = w;
        } else if (u.stride > 1) {
            dxa_mode = -2;                      // subsample shortcut: gradient lives on the coarser grid
        }
        // conv1: params, then dX = (convT(dR1) + dXa) gated by X_in > 0  -> G for the previous unit
        rc = layer_param_grads(tr, u.c1, xin, B, h, w, DR1, h, w, 1, 0, 0, dwraw, colsum, s);
        if (rc) return rc;
        TRY_HIP(before_write(Gin));              // (the G of two units ago: its conv3 / shortcut weight gradients)
        g_shadow_want = ui > 0 && net->units[ui - 1].depth_bn >= 128;      // G of unit ui - 1 -> its conv3 / shortcut weight gradients
        g_h2.mask_fmt = (fast && ui > ub) ? 1 : 0;                           // (unit ub's input is block1's fp32 output)
        TRY_HIP(conv_launch(l1, t1.d_wT, t1.nkT, t1.cinP, DR1, B, h, w, l1.Cout, 0, 0, h, w, l1.Cin, 1, 0, nullptr, nullptr,
                            dxa, dxa_mode, dxa_h, dxa_w, xin, false, 0, 0, Gin, s));
        cur ^= 1;
        TRY_HIP(close_groups_at(ui));
    }
    // ---- root block: max-pool backward (+ stem ReLU gate), stem weight gradient
    static const bool fin_split_env = (dgp_tune("DGP_FIN_SPLIT", 1) != 0);       // A/B switch
    const bool fin_split = fin_split_env && g_ctx->defer_plan && ctx->overlap && tr->n_fin > 1;
    {
        int pth = (net->hp - 1) * 2 + 3 - net->h1; if (pth < 0) pth = 0;
        int ptw = (net->wp - 1) * 2 + 3 - net->w1; if (ptw < 0) ptw = 0;
        const long long tot = (long long)B * net->h1 * net->w1 * 16;
        static const bool pool_idx = (dgp_tune("DGP_POOL_IDX", 1) != 0);
        if (stem_g) {
            if (fin_split) finalise(0, tr->groups.back().fin_count, ctx->s2);      // (what the gradient groups have not finalised yet)
            StemWgradArgs sa{};
            sa.x = F(pl.p0); sa.g = reinterpret_cast<const uint4*>(stem_g); sa.idx = reinterpret_cast<const unsigned char*>(ws + pl.pidx);
            sa.g_prev = stem_g_prev;
            sa.dw = reinterpret_cast<float*>(ws + pl.dw_l[net->conv1]); sa.colsum = reinterpret_cast<float*>(ws + pl.cs_l[net->conv1]);
            sa.B = B; sa.H = d.in_h; sa.W = d.in_w; sa.H1 = net->h1; sa.W1 = net->w1; sa.HP = net->hp; sa.WP = net->wp; sa.pt = pth / 2; sa.pl = ptw / 2;
            sa.bands = (net->h1 + 3) / 4; sa.chunks = (net->w1 + 31) / 32; sa.nitems = B * sa.bands * sa.chunks;
            static int n_cu_s = 0;
            if (!n_cu_s) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu_s, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu_s <= 0) n_cu_s = 256; }
            static const int wgs_per_cu = dgp_tune("DGP_STEM_WGRAD_WGS", 2);      // (1: 6.77, 2: 6.72, 3: 6.72, 4: 6.84 ms per step)
            const int grid = sa.nitems < wgs_per_cu * n_cu_s ? sa.nitems : wgs_per_cu * n_cu_s;
            sa.slab = F(pl.dc1);                  // (the d conv1 map's region: unused on this path; grid x 50 KB)
            hipLaunchKernelGGL(stem_wgrad_h1_kernel, dim3((unsigned)grid), dim3(256), 0, s, sa);
            hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((STEM_SLAB_ROW + 255) / 256, (grid + 31) / 32), dim3(256), 0, s, sa, grid);
        } else if (pool_idx)          // (fused root block in the forward pass: no conv1 map -- unit 0's data gradient is already gated by pool > 0)
            hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(grid_for(tot)), dim3(256), 0, s, (h1p && tr->fwd_stem_fused) ? (const float*)nullptr : F(pl.c1), G[cur],
                               reinterpret_cast<const uchar4*>(ws + pl.pidx), B, net->h1, net->w1, 64, net->hp, net->wp, pth / 2, ptw / 2,
                               F(pl.dc1));
        else
            hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(tot)), dim3(256), 0, s, F(pl.c1), G[cur], B, net->h1, net->w1, 64,
                               net->hp, net->wp, pth / 2, ptw / 2, F(pl.dc1));
        if (stem_g) {
            rc = DGP_OK;
        } else if (fin_split) {
            // the stem's weight gradient (0.3 ms at 11 frames, fp32 MFMA on a 216 MB gradient) is the pass's last launch and nothing else
            // is left to run beside it -- except the finalisation of all OTHER layers: that goes to the second stream behind the last of
            // their weight gradients, the stem's weight gradient to this stream
            finalise(0, tr->groups.back().fin_count, ctx->s2);
            ctx->overlap = false;
            rc = layer_param_grads(tr, net->conv1, F(pl.p0), B, d.in_h, d.in_w, F(pl.dc1), net->h1, net->w1, 2, 3, 3, dwraw, colsum, s);
            ctx->overlap = true;
        } else {
            rc = layer_param_grads(tr, net->conv1, F(pl.p0), B, d.in_h, d.in_w, F(pl.dc1), net->h1, net->w1, 2, 3, 3, dwraw, colsum, s);
        }
        if (rc) return rc;
    }
    (void)nj;
    if (h1p) {            // every H1 gradient tensor of this pass against its predicted scale (wgrad_dma_h1 raised the same flag for its operands)
        H2CheckList cl{};
        if (g_ctx->h2_slots.size() > sizeof(cl.idx) / sizeof(cl.idx[0])) return fail(DGP_ERR_STATE, "16-bit tier: too many H1 gradient tensors");
        cl.n = (int)g_ctx->h2_slots.size();
        for (int k = 0; k < cl.n; ++k) cl.idx[k] = (short)g_ctx->h2_slots[k];
        if (cl.n) hipLaunchKernelGGL(h2_pred_check_kernel, dim3(cl.n), dim3(64), 0, s, cl, g_ctx->rng.pool, g_ctx->rng.prev, tr->d_fast_flag);
    }
    join();                                      // every weight gradient has landed before the finalisation reads them
    if (g_ctx->defer_plan) {
        g_ctx->defer_plan = nullptr;
        const int rest = tr->groups.back().fin_count;      // rows [0, rest) + the stem's row: the last gradient group
        if (!fin_split) finalise(0, rest, s);
        finalise(tr->n_fin - 1, 1, s);
        bn_grads(0, rest, s);
        bn_grads(tr->n_fin - 1, 1, s);
    }
    if (grp_next + 1 != tr->groups.size()) return fail(DGP_ERR_STATE, "backward: a gradient group was not closed");
    TRY_HIP(hipEventRecord(tr->groups.back().ev, s));
    TRY_HIP(hipGetLastError());
    return DGP_OK;
}

/* Fast pass of the training step.  enable != 0: the NEXT dgp_train_forward keeps the retained activations of blocks 2-4 as H2 tensors
 * whose scales are predicted from the ranges the previous pass left behind (call it only after a pass of the same frame count and
 * size; the first pass after dgp_trainer_create, a weight upload or a shape change must be a plain one).  dgp_train_backward follows
 * what the forward did.  dgp_trainer_fast_status: call after the step's work has completed on the stream (it synchronises the
 * device); *failed != 0: a tensor left its predicted range, scoremaps and gradients of that step are NOT valid -- run the step again
 * with enable = 0 before using either. */
int dgp_trainer_fast_mode(dgp_trainer* tr, int32_t enable) {
    if (!tr) return fail(DGP_ERR_INVALID, "dgp_trainer_fast_mode: null");
#ifndef DGP_TUNING
    // (the H2 form measured no faster than the plain pass, EXPERIMENTS.md section 4: an opt-in of tuning builds only; the 16-bit tier's passes use it)
    if (enable && tr->tier != 1) return fail(DGP_ERR_INVALID, "dgp_trainer_fast_mode: needs dgp_trainer_set_tier(tr, 1) (the H2 fast pass is enabled in -DDGP_TUNING builds only)");
#endif
    tr->fast_next = enable != 0;
    return DGP_OK;
}
/* Gradient groups of the LAST dgp_train_backward (data-parallel training): group k = floats [lo[k], hi[k]) of the flat gradient buffer
 * (dgp_trainer_buffer(tr, 1)), in the order the pass completes them -- the heads and the last bottleneck units first, the stem last; the
 * groups tile the buffer.  *n_groups = 0 before the first backward pass. */
int dgp_trainer_grad_groups(dgp_trainer* tr, int32_t max_groups, int32_t* n_groups, int64_t* lo, int64_t* hi) {
    if (!tr || !n_groups) return fail(DGP_ERR_INVALID, "dgp_trainer_grad_groups: null");
    *n_groups = (int32_t)tr->groups.size();
    if ((int)tr->groups.size() > max_groups) return fail(DGP_ERR_INVALID, "dgp_trainer_grad_groups: more groups than max_groups");
    for (size_t k = 0; k < tr->groups.size(); ++k) {
        if (lo) lo[k] = tr->groups[k].lo;
        if (hi) hi[k] = tr->groups[k].hi;
    }
    return DGP_OK;
}
/* Makes `stream` wait until group k of the last dgp_train_backward is final in the gradient buffer (hipStreamWaitEvent: nothing blocks
 * on the host).  A communication stream that waits for group k can all-reduce it while the pass is still computing the later groups. */
int dgp_trainer_grad_group_wait(dgp_trainer* tr, int32_t k, void* stream) {
    if (!tr || k < 0 || k >= (int)tr->groups.size() || !tr->groups[k].ev) return fail(DGP_ERR_INVALID, "dgp_trainer_grad_group_wait: no such group");
    g_ctx = &tr->ctx;
    TRY_HIP(hipStreamWaitEvent((hipStream_t)stream, tr->groups[k].ev, 0));
    return DGP_OK;
}
int dgp_trainer_set_tier(dgp_trainer* tr, int32_t tier) {
    if (!tr || (tier != 0 && tier != 1)) return fail(DGP_ERR_INVALID, "dgp_trainer_set_tier: tier must be 0 (parity) or 1 (16-bit)");
    tr->tier = tier;
    if (!tier) tr->fast_next = false;
    return DGP_OK;
}
int dgp_trainer_get_tier(const dgp_trainer* tr) { return tr ? tr->tier : 0; }

/* One synchronisation for a whole step enqueued without read-backs (forward, loss, backward, dgp_sgd_momentum_clip with a NULL
 * gnorm pointer, dgp_trainer_sync_weights): waits for the device, then *gnorm = the global gradient norm the optimiser saw and the
 * fast / 16-bit pass status as dgp_trainer_fast_status reports it.  When *failed != 0 the optimiser skipped its update on the device
 * (the flag is read by the momentum kernel), so the step can simply be run again on the parity path. */
int dgp_trainer_step_status(dgp_trainer* tr, const float* d_losses, int32_t n_losses, float* losses, float* gnorm, int32_t* was_fast,
                            int32_t* failed, void* stream) {
    if (!tr) return fail(DGP_ERR_INVALID, "dgp_trainer_step_status: null");
    if (n_losses < 0 || n_losses > 8 || (n_losses && (!d_losses || !losses))) return fail(DGP_ERR_INVALID, "dgp_trainer_step_status: 0..8 losses");
    hipStream_t s = (hipStream_t)stream;
    // one tiny kernel at the end of the stream writes everything the host wants into pinned memory; one wait.  (Every launch of the step
    // is on this stream or joined into it: the second stream's forward chain and weight gradients, the optimiser, the re-packing.)
    hipLaunchKernelGGL(step_status_kernel, dim3(1), dim3(64), 0, s, d_losses, n_losses, tr->d_gnorm, tr->fwd_fast ? tr->d_fast_flag : (const int*)nullptr,
                       tr->h_status);
    TRY_HIP(hipGetLastError());
    TRY_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < n_losses; ++i) losses[i] = tr->h_status[i];
    if (gnorm) *gnorm = tr->h_status[8];
    if (was_fast) *was_fast = tr->fwd_fast ? 1 : 0;
    if (failed) *failed = (int32_t)tr->h_status[9];
    return DGP_OK;
}

int dgp_trainer_fast_status(dgp_trainer* tr, int32_t* was_fast, int32_t* failed) {
    if (!tr || !failed) return fail(DGP_ERR_INVALID, "dgp_trainer_fast_status: null");
    int f = 0;
    if (tr->fwd_fast) {
        TRY_HIP(hipDeviceSynchronize());
        TRY_HIP(hipMemcpy(&f, tr->d_fast_flag, sizeof(int), hipMemcpyDeviceToHost));
    }
    if (was_fast) *was_fast = tr->fwd_fast ? 1 : 0;
    *failed = f;
    return DGP_OK;
}

/* ---- single-layer backward entry points (layer-level parity tests at the real shapes; the trainer calls the same launchers) ---- */

/* dWraw[(tap, ci)][co] = sum_m x[m + tap][ci] * dy[m][co] (HWIO order, Cin rows per tap), colsum[co] = sum_m dy[m][co].
 * With both ranges (DGP_ABSMAX_SLOTS floats each) the fp16-split kernel wgrad_h3 runs where the tile is 128 x 128, else wgrad_f32. */
int dgp_conv2d_wgrad(const dgp_conv_desc* d, const float* x, const float* dy, const float* x_absmax, const float* dy_absmax,
                     float* dw_raw, float* colsum, void* stream) {
    g_ctx = nullptr;              // layer-level call: explicit ranges only
    if (!d || !x || !dy || !dw_raw) return fail(DGP_ERR_INVALID, "dgp_conv2d_wgrad: null argument");
    if (d->Cin < 4 || (d->Cin & 3) || ((d->Cin / 4) & (d->Cin / 4 - 1)) || (d->Cout & 3))
        return fail(DGP_ERR_INVALID, "dgp_conv2d_wgrad: Cin must be 4 * 2^k, Cout a multiple of 4");
    if ((double)d->N * d->H * d->W * d->Cin * 4 > 4294967000.0 || (double)d->N * d->Ho * d->Wo * d->Cout * 4 > 4294967000.0)
        return fail(DGP_ERR_INVALID, "dgp_conv2d_wgrad: tensor exceeds 4 GiB");
    hipError_t e = wgrad_launch(x, d->N, d->H, d->W, d->Cin, dy, d->Ho, d->Wo, d->Cout, d->KH, d->KW, d->stride, d->rate, d->pad_t,
                                d->pad_l, dw_raw, colsum, (hipStream_t)stream, false, x_absmax, dy_absmax);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_conv2d_wgrad: ") + hipGetErrorString(e));
    return DGP_OK;
}

/* dgp_conv2d_wgrad through the LDS-DMA tile (wgrad_dma): x and dy are first copied into fp16 high / low cells with the scales that
 * x_prev / dy_prev (range slots "of the previous step") predict, as the producers' epilogues do inside the training step; the kernel
 * checks x_absmax / dy_absmax (this step's ranges) against them and runs the fp32-MFMA tile on x / dy where the copies are unusable.
 * scratch: 4 * (numel(x) + numel(dy)) device bytes. */
int dgp_conv2d_wgrad_shadow(const dgp_conv_desc* d, const float* x, const float* dy, const float* x_absmax, const float* dy_absmax,
                            const float* x_prev, const float* dy_prev, void* scratch, float* dw_raw, float* colsum, void* stream) {
    g_ctx = nullptr;
    if (!d || !x || !dy || !dw_raw || !scratch || !x_absmax || !dy_absmax || !x_prev || !dy_prev)
        return fail(DGP_ERR_INVALID, "dgp_conv2d_wgrad_shadow: null argument");
    if (d->Cin < 16 || (d->Cin & 15) || ((d->Cin / 4) & (d->Cin / 4 - 1)) || (d->Cout & 7) || d->Cout < 128 || d->KH * d->KW * d->Cin < 128)
        return fail(DGP_ERR_INVALID, "dgp_conv2d_wgrad_shadow: Cin must be 16 * 2^k, Cout a multiple of 8, and the tile 128 x 128 (K, Cout >= 128)");
    if ((double)d->N * d->H * d->W * d->Cin * 4 > 4200000000.0 || (double)d->N * d->Ho * d->Wo * d->Cout * 4 > 4200000000.0)
        return fail(DGP_ERR_INVALID, "dgp_conv2d_wgrad_shadow: tensor exceeds 4 GiB");
    hipStream_t s = (hipStream_t)stream;
    const long long nx = (long long)d->N * d->H * d->W * d->Cin, ny = (long long)d->N * d->Ho * d->Wo * d->Cout;
    float* xs = (float*)scratch;
    float* dys = xs + nx;
    hipLaunchKernelGGL(f32_to_shadow_kernel, dim3(grid_for(nx / 8)), dim3(256), 0, s, reinterpret_cast<const float4*>(x), nx / 8, x_prev,
                       reinterpret_cast<uint4*>(xs));
    hipLaunchKernelGGL(f32_to_shadow_kernel, dim3(grid_for(ny / 8)), dim3(256), 0, s, reinterpret_cast<const float4*>(dy), ny / 8, dy_prev,
                       reinterpret_cast<uint4*>(dys));
    hipError_t e = wgrad_launch(x, d->N, d->H, d->W, d->Cin, dy, d->Ho, d->Wo, d->Cout, d->KH, d->KW, d->stride, d->rate, d->pad_t,
                                d->pad_l, dw_raw, colsum, s, false, x_absmax, dy_absmax, xs, x_prev, dys, dy_prev);
    if (e != hipSuccess) return fail(DGP_ERR_HIP, std::string("dgp_conv2d_wgrad_shadow: ") + hipGetErrorString(e));
    return DGP_OK;
}

/* dx = gate( convT(dy; w * scale) + dx_add ): the data gradient of the forward conv `d` (x [N,H,W,Cin] -> y [N,Ho,Wo,Cout]).
 * w_hwio: device HWIO weights; scale: [Cout] device or NULL; mask: [N,H,W,Cin] (gate: mask > 0) or NULL; dx_add: gradient of the
 * shortcut branch or NULL, on dx's grid (add_mode 1) or on the 2x coarser grid (add_mode -2: the subsample shortcut).
 * scratch: >= dgp_conv2d_dgrad_scratch_bytes(d) device bytes (data-gradient panel, its fp16 cells, range slots).
 * ranged bit 0: measure max |dy| and run the fp16-split kernels (what the trainer does), else the bf16x6 split; bit 1: `mask` is an H2
 * tensor (dgp_f32_to_h2 of the activation, any scale) as in the fast pass of the training step -- fp16-split kernels only. */
size_t dgp_conv2d_dgrad_scratch_bytes(const dgp_conv_desc* d) {
    if (!d) return 0;
    const size_t panel = (size_t)nk_for(d->KH, d->KW, d->Cout) * 8 * coutp_for(d->Cin) * 16;
    return 2 * panel + 4 * ABSMAX_SLOTS * sizeof(float) + (size_t)d->Cout * sizeof(float) + 1024;
}

int dgp_conv2d_dgrad(const dgp_conv_desc* d, const float* dy, const float* w_hwio, const float* scale, const float* mask,
                     const float* dx_add, int32_t add_mode, float* dx, void* scratch, int32_t ranged, void* stream) {
    g_ctx = nullptr;
    if (!d || !dy || !w_hwio || !dx || !scratch) return fail(DGP_ERR_INVALID, "dgp_conv2d_dgrad: null argument");
    // (Cin < 64: dx goes through the 32-column fp32 tile, which reads the gate as fp32 -- garbage with H2 cells; found by scripts/fuzz_backward_layers.py)
    if ((ranged & 2) && (!(ranged & 1) || !mask || (d->Cin & 7) || d->Cin < 64))
        return fail(DGP_ERR_INVALID, "dgp_conv2d_dgrad: an H2 gate needs ranged bit 0, a mask and Cin % 8 == 0, Cin >= 64");
    if ((d->Cout & 31) || (d->Cin & 3)) return fail(DGP_ERR_INVALID, "dgp_conv2d_dgrad: Cout % 32, Cin % 4");
    hipStream_t s = (hipStream_t)stream;
    const int taps = d->KH * d->KW, nkT = nk_for(d->KH, d->KW, d->Cout), cinP = coutp_for(d->Cin);
    const size_t panel_bytes = (size_t)nkT * 8 * cinP * 16;
    char* sp = (char*)scratch;
    float* panel = (float*)sp; sp += panel_bytes;
    void* cells = sp; sp += panel_bytes;
    float* rng = (float*)sp; sp += 4 * ABSMAX_SLOTS * sizeof(float);       // [0]: dy, [1]: panel, [2]: dx
    float* ones = (float*)sp;
    TRY_HIP(hipMemsetAsync(rng, 0, 4 * ABSMAX_SLOTS * sizeof(float), s));
    if (!scale) {
        std::vector<float> h(d->Cout, 1.f);
        TRY_HIP(hipMemcpyAsync(ones, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, s));
        TRY_HIP(hipStreamSynchronize(s));
        scale = ones;
    }
    const long long totT = (long long)nkT * 8 * cinP;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid_for(totT)), dim3(256), 0, s, w_hwio, scale, taps, d->Cin, d->Cout, cinP, nkT * 8,
                       panel, rng + ABSMAX_SLOTS);
    ConvArgs a{};
    const int keff_h = (d->KH - 1) * d->rate + 1, keff_w = (d->KW - 1) * d->rate + 1;
    a.in = dy; a.wpk = panel; a.out = dx; a.mask = mask; a.res = dx_add;
    a.N = d->N; a.H = d->Ho; a.W = d->Wo; a.Cin = d->Cout; a.log2cin4 = ilog2(d->Cout / 4);
    a.Ho = d->H; a.Wo = d->W; a.Cout = d->Cin; a.CoutP = cinP;
    a.KH = d->KH; a.KW = d->KW; a.stride = 1; a.dil = d->rate; a.pad_t = keff_h - 1 - d->pad_t; a.pad_l = keff_w - 1 - d->pad_l;
    a.ntaps = taps; a.nk = nkT; a.M = d->N * d->H * d->W;
    a.up = d->stride > 1 ? d->stride : 0;
    a.res_s = dx_add ? add_mode : 0;
    a.res_H = add_mode == -2 ? (d->H + 1) / 2 : d->H; a.res_W = add_mode == -2 ? (d->W + 1) / 2 : d->W;
    a.in_bytes = (unsigned)((size_t)d->N * d->Ho * d->Wo * d->Cout * 4);
    a.out_bytes = (unsigned)((size_t)a.M * d->Cin * 4);
    a.res_bytes = dx_add ? (unsigned)((size_t)d->N * a.res_H * a.res_W * d->Cin * 4) : 0u;
    a.w_bytes = (unsigned)panel_bytes;
    a.mask_fmt = (ranged & 2) ? 1 : 0;          // the gate tensor is H2 (fast pass of the training step): gate = stored value > 0
    if (ranged & 1) {
        TRY_HIP(launch_absmax(dy, (long long)d->N * d->Ho * d->Wo * d->Cout, rng, s));
        TRY_HIP(launch_pack_h3(panel, nkT, cinP, rng + ABSMAX_SLOTS, cells, s));
        a.in_absmax = rng; a.w_absmax = rng + ABSMAX_SLOTS; a.wh3 = cells; a.wh3_bytes = a.w_bytes;
        a.out_absmax = rng + 2 * ABSMAX_SLOTS;
    }
    TRY_HIP(launch_conv(a, pick_tile(a.M, cinP, nkT * BK, a.in_absmax && a.w_absmax), s));
    return DGP_OK;
}

/* host <-> device copies of a slice of one of the flat buffers (which: 0 params, 1 grads, 2 momentum, 3 stats) */
int dgp_trainer_upload(dgp_trainer* tr, int32_t which, int64_t offset, const float* host, int64_t n) {
    float* b = dgp_trainer_buffer(tr, which);
    if (!b || !host || offset < 0 || n < 0) return fail(DGP_ERR_INVALID, "dgp_trainer_upload: bad argument");
    TRY_HIP(hipMemcpy(b + offset, host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    return DGP_OK;
}
int dgp_trainer_download(dgp_trainer* tr, int32_t which, int64_t offset, float* host, int64_t n) {
    float* b = dgp_trainer_buffer(tr, which);
    if (!b || !host || offset < 0 || n < 0) return fail(DGP_ERR_INVALID, "dgp_trainer_download: bad argument");
    TRY_HIP(hipDeviceSynchronize());
    TRY_HIP(hipMemcpy(host, b + offset, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return DGP_OK;
}

int dgp_sgd_momentum_clip(dgp_trainer* tr, float lr, float momentum, float clip_norm, float* gnorm_host_or_null, void* stream) {
    if (!tr) return fail(DGP_ERR_INVALID, "dgp_sgd_momentum_clip: null");
    hipStream_t s = (hipStream_t)stream;
    const int k = tr->sumsq_k;
    tr->sumsq_k ^= 1;
    // (768 workgroups, three per CU: every workgroup ends in ONE fp64 atomic on the same address, and 4 096 of them in a row cost more than the 102 MB read)
    const int sumsq_grid = grid_for(tr->n_train) < 768 ? grid_for(tr->n_train) : 768;
    hipLaunchKernelGGL(sumsq_kernel, dim3(sumsq_grid), dim3(256), 0, s, tr->grads, tr->n_train, tr->d_sumsq + k);
    hipLaunchKernelGGL(momentum_kernel, dim3(grid_for(tr->n_train)), dim3(256), 0, s, tr->params, tr->grads, tr->mom, tr->n_train,
                       lr, momentum, clip_norm, tr->d_sumsq + k, tr->d_sumsq + (k ^ 1), tr->d_gnorm, tr->fwd_fast ? tr->d_fast_flag : nullptr);
    TRY_HIP(hipGetLastError());
    if (gnorm_host_or_null) {
        TRY_HIP(hipStreamSynchronize(s));
        TRY_HIP(hipMemcpy(gnorm_host_or_null, tr->d_gnorm, sizeof(float), hipMemcpyDeviceToHost));
    }
    return DGP_OK;
}

}  // extern "C"
