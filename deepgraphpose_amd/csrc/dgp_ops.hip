// gfx950 (MI355X / CDNA4): the non-convolution kernels of the Deep Graph Pose hot path (split off dgp_kernels.hip in round 4).
//
//   pack_h3 / pack_h3_all   fp32 weight panels -> fp16 high / low cells (the LDS order of the fp16-split conv kernels)
//   absmax, reduce_slabs, head_gather, f32 <-> H2 converters, h2_range_check
//   maxpool3x3s2_same       K3;  preprocess_u8  K1 (uint8 -> fp32, mean-pixel subtraction, pad C 3 -> 4)
//   stem_pool_fused         the whole root block in one kernel (uint8 frame -> conv1 7x7/2 + BN + ReLU -> max-pool -> H2 cells)
//   motion_energy           the hidden-frame selector's scan (DGP/dataset.py:29-43), bit-exact wrapped uint8 sums
//   soft_argmax             K9 + K10 (DGP/models/fitdgp_util.py:342-402, DGP/models/eval.py:331-343); maps of any size
//   hard_argmax             K11 (PET/nnet/predict.py:62-77);  pmap_threshold: argmax_2d_from_cm's `th` branch
//
// Wavefront = 64 lanes everywhere.  No CUDA-compat shims; this file only targets gfx950.
#include "dgp_internal.h"
#include "dgp_device.h"
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
#include <type_traits>

namespace dgp {

// fp32 weight panel -> fp16 high/low cells in the LDS order of the fp16-split kernels (one thread per k-group x column)
__global__ __launch_bounds__(256) void pack_h3_kernel(const float4* __restrict__ panel, int nkg, int CoutP,
                                                      const float* __restrict__ w_absmax, uint4* __restrict__ out) {
    const float s = pow2_scale_for(w_absmax, threadIdx.x & 63);
    const long long total = (long long)nkg * CoutP;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(g % CoutP);
        const long long kg = g / CoutP;
        uint2 h0, l0, h1, l1;
        split2_f16(panel[(2 * kg) * CoutP + col], s, h0, l0);
        split2_f16(panel[(2 * kg + 1) * CoutP + col], s, h1, l1);
        out[(kg * 2) * CoutP + col] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        out[(kg * 2 + 1) * CoutP + col] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

hipError_t launch_pack_h3(const float* panel, int nk, int CoutP, const float* w_absmax, void* out, hipStream_t s) {
    const long long total = (long long)nk * 4 * CoutP;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_h3_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(panel), nk * 4,
                       CoutP, w_absmax, reinterpret_cast<uint4*>(out));
    return hipGetLastError();
}

// fp32 weight panel -> H1 cells (the 16-bit tier): the HIGH halves only, [k-group of 8 channels][column][8 halves] -- the high plane of
// pack_h3's cells, compacted; same power-of-two scale (max |w| -> [2^14, 2^15)): 2 bytes per weight where the H2 cells and the fp32
// panel have 4.
__global__ __launch_bounds__(256) void pack_h1_kernel(const float4* __restrict__ panel, int nkg, int CoutP,
                                                      const float* __restrict__ w_absmax, uint4* __restrict__ out) {
    const float s = pow2_scale_for(w_absmax, threadIdx.x & 63);
    const long long total = (long long)nkg * CoutP;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(g % CoutP);
        const long long kg = g / CoutP;
        const float4 a = panel[(2 * kg) * CoutP + col], b = panel[(2 * kg + 1) * CoutP + col];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        out[g] = h1_pack8(v, s);
    }
}

hipError_t launch_pack_h1(const float* panel, int nk, int CoutP, const float* w_absmax, void* out, hipStream_t s) {
    const long long total = (long long)nk * 4 * CoutP;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_h1_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(panel), nk * 4,
                       CoutP, w_absmax, reinterpret_cast<uint4*>(out));
    return hipGetLastError();
}

// pack_h3_kernel for a table of panels in one launch (the trainer re-splits every panel after each optimiser step)
__global__ __launch_bounds__(256) void pack_h3_all_kernel(const PackH3Desc* __restrict__ table) {
    const PackH3Desc d = table[blockIdx.y];
    const long long total = (long long)d.nkg * d.CoutP;
    if ((long long)blockIdx.x * blockDim.x >= total) return;
    const float s = pow2_scale_for(d.rng, threadIdx.x & 63);
    const float4* panel = reinterpret_cast<const float4*>(d.panel);
    uint4* out = reinterpret_cast<uint4*>(d.out);
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(g % d.CoutP);
        const long long kg = g / d.CoutP;
        uint2 h0, l0, h1, l1;
        split2_f16(panel[(2 * kg) * d.CoutP + col], s, h0, l0);
        split2_f16(panel[(2 * kg + 1) * d.CoutP + col], s, h1, l1);
        out[(kg * 2) * d.CoutP + col] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        out[(kg * 2 + 1) * d.CoutP + col] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// pack_h1_kernel for a table of panels in one launch (the 16-bit trainer re-packs every panel after each optimiser step)
__global__ __launch_bounds__(256) void pack_h1_all_kernel(const PackH3Desc* __restrict__ table) {
    const PackH3Desc d = table[blockIdx.y];
    const long long total = (long long)d.nkg * d.CoutP;
    if ((long long)blockIdx.x * blockDim.x >= total) return;
    const float s = pow2_scale_for(d.rng, threadIdx.x & 63);
    const float4* panel = reinterpret_cast<const float4*>(d.panel);
    uint4* out = reinterpret_cast<uint4*>(d.out);
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(g % d.CoutP);
        const long long kg = g / d.CoutP;
        const float4 a = panel[(2 * kg) * d.CoutP + col], b = panel[(2 * kg + 1) * d.CoutP + col];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        out[g] = h1_pack8(v, s);
    }
}

hipError_t launch_pack_h1_all(const PackH3Desc* table_dev, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_h1_all_kernel, dim3(512, (unsigned)n), dim3(256), 0, s, table_dev);
    return hipGetLastError();
}

hipError_t launch_pack_h3_all(const PackH3Desc* table_dev, int n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_h3_all_kernel, dim3(512, (unsigned)n), dim3(256), 0, s, table_dev);
    return hipGetLastError();
}

// max |x| of a tensor into a device scalar (range of an operand of the fp16-split kernels when no producer tracked it)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n4, long long n, float* __restrict__ out) {
    float m = 0.f;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (long long)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(x + g * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long long i = n4 * 4; i < n; ++i) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        atomicMax(reinterpret_cast<unsigned*>(out) + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (ABSMAX_SLOTS - 1)), __float_as_uint(m));
}

hipError_t launch_absmax(const float* x, long long n, float* out, hipStream_t s) {
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n4, n, out);
    return hipGetLastError();
}

// Deterministic split-K combine for the heads: out[i] = sum_s slab[s][i] in fixed order (no float atomics).
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, long long n4, long long stride,
                                                           int nsplit, float* __restrict__ out) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (long long)gridDim.x * blockDim.x) {
        float4 a = *reinterpret_cast<const float4*>(slabs + g * 4);
        for (int s = 1; s < nsplit; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(slabs + (long long)s * stride + g * 4);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4*>(out + g * 4) = a;
    }
}

hipError_t launch_reduce_slabs(const float* slabs, long long n, long long stride, int nsplit, float* out, hipStream_t s) {
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slabs, n4, stride, nsplit, out);
    return hipGetLastError();
}

// Gather of the pointwise head (dgp_net.hip run_head_pointwise): T [B][h][w][ldt] holds, per input pixel, the contributions
// (tap, phase, joint) -> column (tap * 4 + phase) * njt + joint; output pixel (2 ho + a, 2 wo + b) sums its <= 4 taps in fixed order.
__global__ __launch_bounds__(256) void head_gather_kernel(const float* __restrict__ T, const float* __restrict__ bias, int B, int h, int w,
                                                          int njt, int ldt, float* __restrict__ out) {
    const long long n = (long long)B * 2 * h * 2 * w * njt;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(g % njt);
        long long r = g / njt;
        const int x = (int)(r % (2 * w)); r /= 2 * w;
        const int y = (int)(r % (2 * h));
        const int b = (int)(r / (2 * h));
        const int ho = y >> 1, pa = y & 1, wo = x >> 1, pb = x & 1;
        const int ph = pa * 2 + pb;
        float acc = bias ? bias[ph * njt + c] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            const int hi = ho - 1 + (tap >> 1), wi = wo - 1 + (tap & 1);
            if ((unsigned)hi < (unsigned)h && (unsigned)wi < (unsigned)w)
                acc += T[(((long long)b * h + hi) * w + wi) * ldt + (tap * 4 + ph) * njt + c];
        }
        out[g] = acc;
    }
}

hipError_t launch_head_gather(const float* T, const float* bias, int B, int h, int w, int njt, int ldt, float* out, hipStream_t s) {
    const long long n = (long long)B * 2 * h * 2 * w * njt;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(head_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, s, T, bias, B, h, w, njt, ldt, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// 3x3 / stride 2 max-pool, TF 'SAME' padding (pad_before = pad_total/2, padded cells never
// win).  NHWC fp32, 4 channels (16 B) per thread.  HBM-bound.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_same(const float* __restrict__ x, int N, int H, int W,
                                                         int C4, int Ho, int Wo, int pt, int pl,
                                                         float* __restrict__ y, float h2_scale) {
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(g % C4);
        long long pix = g / C4;
        const int wo = (int)(pix % Wo);
        pix /= Wo;
        const int ho = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int hi = ho * 2 - pt + a;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int wi = wo * 2 - pl + b;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(
                    x + ((((long long)n * H + hi) * W + wi) * C4 + c4) * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y);
                m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        if (h2_scale > 0.f) {      // H2 output: this thread's 4 channels are one half of an 8-channel cell pair [hi 16 B | lo 16 B]
            uint2 ph, pl2;
            split2_f16(m, h2_scale, ph, pl2);
            uint2* cell = reinterpret_cast<uint2*>(y + (g - c4 + (c4 & ~1)) * 4);      // first float slot of the 8-channel group
            cell[c4 & 1] = ph;
            cell[2 + (c4 & 1)] = pl2;
        } else
        *reinterpret_cast<float4*>(y + g * 4) = m;
    }
}

// fp32 <-> H2 cells, one thread per 8-channel group (test / boundary helpers: the engine's kernels convert in their epilogues)
__global__ __launch_bounds__(256) void f32_to_h2_kernel(const float4* __restrict__ x, long long ng, float scale, uint4* __restrict__ out) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (long long)gridDim.x * blockDim.x) {
        const float4 a = x[2 * g], b = x[2 * g + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint4 hi, lo;
        h2_pack8(v, scale, hi, lo);
        out[2 * g] = hi; out[2 * g + 1] = lo;
    }
}
__global__ __launch_bounds__(256) void h2_to_f32_kernel(const uint4* __restrict__ x, long long ng, float inv_scale, float4* __restrict__ out) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (long long)gridDim.x * blockDim.x) {
        float v[8];
        h2_unpack8(x[2 * g], x[2 * g + 1], inv_scale, v);
        out[2 * g] = make_float4(v[0], v[1], v[2], v[3]); out[2 * g + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}
__global__ __launch_bounds__(256) void f32_to_h1_kernel(const float4* __restrict__ x, long long ng, float scale, uint4* __restrict__ out) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (long long)gridDim.x * blockDim.x) {
        const float4 a = x[2 * g], b = x[2 * g + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        out[g] = h1_pack8(v, scale);
    }
}
__global__ __launch_bounds__(256) void h1_to_f32_kernel(const uint4* __restrict__ x, long long ng, float inv_scale, float4* __restrict__ out) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (long long)gridDim.x * blockDim.x) {
        float v[8];
        h1_unpack8(x[g], inv_scale, v);
        out[2 * g] = make_float4(v[0], v[1], v[2], v[3]); out[2 * g + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}
hipError_t launch_f32_to_h1(const float* x, long long ng, float scale, void* out, hipStream_t s) {
    long long blocks = (ng + 255) / 256; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(f32_to_h1_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(x), ng, scale, reinterpret_cast<uint4*>(out));
    return hipGetLastError();
}
hipError_t launch_h1_to_f32(const void* x, long long ng, float inv_scale, float* out, hipStream_t s) {
    long long blocks = (ng + 255) / 256; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(h1_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(x), ng, inv_scale, reinterpret_cast<float4*>(out));
    return hipGetLastError();
}
hipError_t launch_f32_to_h2(const float* x, long long ng, float scale, void* out, hipStream_t s) {
    long long blocks = (ng + 255) / 256; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(f32_to_h2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(x), ng, scale, reinterpret_cast<uint4*>(out));
    return hipGetLastError();
}
hipError_t launch_h2_to_f32(const void* x, long long ng, float inv_scale, float* out, hipStream_t s) {
    long long blocks = (ng + 255) / 256; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(h2_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(x), ng, inv_scale, reinterpret_cast<float4*>(out));
    return hipGetLastError();
}

// One wave per layer: flag[0] |= 1 when max |tensor| * 2^exp of a layer leaves the range the fp16 high parts can hold (inf in the
// cells); flag[1] counts the launches.  exps[li] == INT_MIN marks a layer whose output is not an H2 tensor.
__global__ __launch_bounds__(64) void h2_range_check_kernel(const float* __restrict__ amax, const int* __restrict__ exps, int n_layers,
                                                           int* __restrict__ flag) {
    const int li = blockIdx.x, lane = threadIdx.x;
    if (li >= n_layers || exps[li] == (int)0x80000000) return;
    float mx = 0.f;
    for (int i = lane; i < ABSMAX_SLOTS; i += 64) mx = fmaxf(mx, amax[(size_t)li * ABSMAX_SLOTS + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0 && (!(ldexpf(mx, exps[li]) < 60000.f))) atomicOr(flag, 1);          // (NaN compares false: flagged too)
}
hipError_t launch_h2_range_check(const float* amax_slots, const int* exps, int n_layers, int* flag, hipStream_t s) {
    hipLaunchKernelGGL(h2_range_check_kernel, dim3((unsigned)n_layers), dim3(64), 0, s, amax_slots, exps, n_layers, flag);
    return hipGetLastError();
}

hipError_t launch_maxpool(const float* x, int N, int H, int W, int C, float* y, hipStream_t s, float h2_scale) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pth = ((Ho - 1) * 2 + 3 - H) > 0 ? ((Ho - 1) * 2 + 3 - H) : 0;
    const int ptw = ((Wo - 1) * 2 + 3 - W) > 0 ? ((Wo - 1) * 2 + 3 - W) : 0;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool3x3s2_same, dim3((unsigned)blocks), dim3(256), 0, s, x, N, H, W, C / 4, Ho, Wo,
                       pth / 2, ptw / 2, y, h2_scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// uint8 RGB -> fp32 (x - mean_pixel), channel-padded 3 -> 4 (PET/nnet/pose_net.py:38-40).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_u8(const uint8_t* __restrict__ f, long long npix, float m0,
                                                     float m1, float m2, float* __restrict__ out) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < npix;
         g += (long long)gridDim.x * blockDim.x) {
        const uint8_t* q = f + g * 3;
        float4 v;
        v.x = (float)q[0] - m0;
        v.y = (float)q[1] - m1;
        v.z = (float)q[2] - m2;
        v.w = 0.f;
        *reinterpret_cast<float4*>(out + g * 4) = v;
    }
}

hipError_t launch_preprocess(const uint8_t* f, long long npix, float m0, float m1, float m2, float* out,
                             hipStream_t s) {
    long long blocks = (npix + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(preprocess_u8, dim3((unsigned)blocks), dim3(256), 0, s, f, npix, m0, m1, m2, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Motion energy of a frame sequence (DGP/dataset.py:29-43): sums[t] = sum over the bytes of frame t of
// (f_t - f_{t-1}) mod 256 -- the reference subtracts uint8 arrays, so the difference WRAPS and np.abs is the identity; the mean
// is sums[t] / frame_bytes on the host (exact in float64).  Integer work: exact, order-independent (integer atomics).
// A block owns one 4-KiB column of the frame and walks a group of MEG frames with the previous frame's 16 bytes in registers:
// every byte is read 1 + 1/MEG times.  HBM-bound: algorithmic bytes = n_frames * frame_bytes.
// ------------------------------------------------------------------------------------------------
constexpr int MEG = 16;
__device__ __forceinline__ unsigned bytes_sub_sum(unsigned a, unsigned b, unsigned acc) {
    const unsigned H = 0x80808080u;
    const unsigned d = ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);       // per-byte a - b mod 256, no borrow across bytes
    return __builtin_amdgcn_sad_u8(d, 0u, acc);                       // + the four bytes of d
}
// VEC = 16: frames whose byte size is a multiple of 16 (16-byte loads); VEC = 1: any size, byte loads.  A lane owns MEU units
// 256 apart (coalesced), so a wave issues MEU loads per frame and one integer atomic per frame.
constexpr int MEU = 4;
template <int VEC>
__global__ __launch_bounds__(256) void motion_energy_kernel(const uint8_t* __restrict__ frames, long long frame_bytes, int n_frames,
                                                            const uint8_t* __restrict__ prev_frame,
                                                            unsigned long long* __restrict__ sums) {
    const long long nu = frame_bytes / VEC;
    const long long u0 = (long long)blockIdx.x * (256 * MEU) + threadIdx.x;  // first unit of the frame owned by this lane
    const int t0 = blockIdx.y * MEG, t1 = min(n_frames, t0 + MEG);
    const uint8_t* pf = t0 > 0 ? frames + (long long)(t0 - 1) * frame_bytes : prev_frame;
    auto load = [&](const uint8_t* f, int k) {
        const long long u = u0 + 256 * k;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (u < nu) {
            if (VEC == 16) v = *reinterpret_cast<const uint4*>(f + u * 16);
            else v.x = f[u];
        }
        return v;
    };
    uint4 prev[MEU];
#pragma unroll
    for (int k = 0; k < MEU; ++k) prev[k] = pf ? load(pf, k) : make_uint4(0, 0, 0, 0);
    for (int t = t0; t < t1; ++t) {
        const uint8_t* f = frames + (long long)t * frame_bytes;
        uint4 cur[MEU];
#pragma unroll
        for (int k = 0; k < MEU; ++k) cur[k] = load(f, k);
        unsigned s = 0;
        if (pf) {
#pragma unroll
            for (int k = 0; k < MEU; ++k) {
                s = bytes_sub_sum(cur[k].x, prev[k].x, s);
                if (VEC == 16) {
                    s = bytes_sub_sum(cur[k].y, prev[k].y, s); s = bytes_sub_sum(cur[k].z, prev[k].z, s);
                    s = bytes_sub_sum(cur[k].w, prev[k].w, s);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MEU; ++k) prev[k] = cur[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(sums + t, (unsigned long long)s);
        pf = f;
    }
}

hipError_t launch_motion_energy(const uint8_t* frames, long long frame_bytes, int n_frames, const uint8_t* prev_frame,
                                unsigned long long* sums, hipStream_t s) {
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)n_frames * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const bool vec = (frame_bytes & 15) == 0 && (((uintptr_t)frames | (uintptr_t)prev_frame) & 15) == 0;
    const long long units = vec ? frame_bytes >> 4 : frame_bytes;
    const dim3 grid((unsigned)((units + 256 * MEU - 1) / (256 * MEU)), (unsigned)((n_frames + MEG - 1) / MEG));
    if (vec) hipLaunchKernelGGL(motion_energy_kernel<16>, grid, dim3(256), 0, s, frames, frame_bytes, n_frames, prev_frame, sums);
    else hipLaunchKernelGGL(motion_energy_kernel<1>, grid, dim3(256), 0, s, frames, frame_bytes, n_frames, prev_frame, sums);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Block-wide reductions (256 threads = 4 waves of 64).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------
// Root block in ONE kernel: uint8 frame -> (x - mean_pixel) -> conv2d_same(64, 7, stride 2) + BN + ReLU -> max_pool2d(3, 2, SAME)
// -> H2 cells of the pool output.  (PET/nnet/pose_net.py:36-54 -> slim resnet_v1 root block)
// Layer by layer the root block moves 1.8 GB per 32-frame step (centred fp32 frame out and in, the 64-channel conv1 map out and
// in) for 46 GFLOP; fused it reads the uint8 frames (29 MB) and writes the pool output (157 MB).
//   tile      5 x 16 pool pixels <- 11 x 33 conv1 pixels (363 GEMM rows, 23 blocks of 16: three per wave) <- 27 x 72 input pixels
//   phase 1   all 512 threads: input pixels -> (x - round(mean_c), 1) as fp16 -- exact small integers, so there is NO low plane: the
//             fractional part of the mean rides on the fourth channel (1 inside the frame, 0 in the padding), whose weight is
//             sum_c (round(mean_c) - mean_c) w_c (built with the row panel, dgp_net.hip).  One LDS plane of 8 bytes per pixel: a GEMM
//             row's k-group (2 adjacent pixels) is ONE 16-byte read
//   phase 2   GEMM M = 368, N = 64, K = 7 kernel rows x (8 pixels x 4 channels) on v_mfma_f32_16x16x32_f16, 2 MFMAs per product
//             (a x w_low, a x w_high; round 2: 3, with a centred fp32 input split into two planes); the
//             whole weight panel (57 KB of pre-split cells, the stem row panel of the layer kernels) stays in LDS for the kernel's
//             lifetime; a wave owns 2-3 row blocks and all 4 column blocks
//   phase 3   BN + ReLU, conv1 pixels outside the map := 0 (never win: ReLU outputs are >= 0 and every window holds a real pixel),
//             tile to LDS as fp32 (aliases the phase-1 planes)
//   phase 4   3 x 3 / 2 max over the LDS tile, H2 split with the pool output's scale, two 16-byte stores per 8 channels, range tracking
// One persistent workgroup per CU (156 KB of LDS) walks the tiles.
// H1T (round 6; the 16-bit tier): TWO workgroups per CU, so that one's GEMM runs under the other's fetch / convert / pool
// phases.  74 KB each: ONE weight plane (the tier's layers all multiply by the weights' high cells), a 3 x 16 pool tile (7 x 33 conv1
// pixels, 15 row blocks), the input planes in their OWN 11 KB (so a tile needs two barriers -- planes ready, conv1 tile ready -- instead of
// four) and the conv1 tile in LDS as fp16(x * out_scale) -- rounding is monotone, so the maximum of the rounded values
// is the rounded maximum: the cells are bit-identical to rounding after the pool.  GEMM column 16 j + l15 holds channel 4 l15 + j, so a
// lane's four accumulators of a pixel are four consecutive channels = one 8-byte LDS write, and a pool thread reads a cell in one 16-byte
// read.  The range is tracked on the fp32 values in phase 3 (every conv1 pixel of the map lies in some valid window: the same maximum).
// The training step's record of each window's first maximum is taken on these fp16 values (two fp32 values that round to one fp16 value tie).
// ------------------------------------------------------------------------------------
struct StemPoolArgs {
    const unsigned char* frames;      // [B, H, W, 3] starts fr_delta bytes behind this (4-byte aligned) address
    int fr_delta, fr_bytes;           // fr_bytes = fr_delta + B H W 3
    const uint4* wcells;              // [7 kernel rows][4 k-groups][2 planes][64][8 halves]  (launch_pack_h3 of the stem row panel)
    const float* w_absmax;            // its range slots
    const float* bn_scale; const float* bn_bias;
    float mean0, mean1, mean2, out_scale;   // mean: round(mean_pixel[c]) (the fraction is in the fourth channel's weights)
    float* out;                       // H2 [B, HP, WP, 64] (out_h1: H1 cells, half the bytes)
    float* out_absmax;
    int out_h1;
    const float* out_prev;            // training step: the output's scale is predicted on the device (shadow_scale_for) instead of out_scale
    unsigned char* idx;               // training step: [B, HP, WP, 64] position (a * 3 + b) of each window's first maximum (maxpool_fwd_idx_kernel's
                                      // record; windows whose maximum is 0 record position 0 -- their gradient is gated to zero anyway)
    int B, H, W, H1, W1, HP, WP, pbh, pbw, tiles_h, tiles_w, ntiles;
};

#ifndef DGP_STEM_PH
#define DGP_STEM_PH 5      // pool rows per tile of the parity tier's kernel.  5: the conv1 tile aliases the input planes, four barriers per tile; 4 / 3: own
                           // buffers, two barriers -- same-box A/B (scripts/r6_stem_ph.sh): 255-261 us at 4 against 256-259 at 5, 297-301 at 3: with ONE workgroup per
                           // CU the phases are serial either way
#endif
template <bool H1T>
__global__ __launch_bounds__(512, 2) void stem_pool_fused_kernel(const StemPoolArgs p) {
    constexpr int PH = H1T ? 3 : DGP_STEM_PH, PW = 16, SR = 2 * PH + 1, SC = 2 * PW + 1, MS = SR * SC, NRB = (MS + 15) / 16;     // 11, 33, 363, 23 (H1T: 7, 33, 231, 15)
    constexpr int IR = 2 * SR + 5, IC = 72;                                                                   // 27 x 72 input pixels (H1T: 19 x 72)
    static_assert(NRB <= 24 && NRB > 8, "one to three row blocks per wave");
    constexpr int NPL = H1T ? 1 : 2;                                                                          // weight planes in LDS
    constexpr int WCELLS = 7 * 4 * NPL * 64;
    constexpr int LDC = 68;                                                                                   // floats per conv1 pixel in LDS
    constexpr int LDH = 72;                                                                                   // H1T: halves per conv1 pixel
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* sW = reinterpret_cast<uint4*>(smem);
    char* sU = smem + WCELLS * 16;
    uint2* sHi = reinterpret_cast<uint2*>(sU);                     // [IR][IC]
    constexpr bool ALIAS = !H1T && PH >= 5;                         // the conv1 tile aliases the planes where both do not fit (5 x 16 tiles of the parity tier)
    float* sC = reinterpret_cast<float*>(sU + (ALIAS ? 0 : IR * IC * 8));      // [NRB * 16][LDC]
    _Float16* sCh = reinterpret_cast<_Float16*>(sU + IR * IC * 8); // H1T: [NRB * 16][LDH], BEHIND the planes (not aliased: two barriers per tile instead of four)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    for (int i = t; i < WCELLS; i += 512) {
        if constexpr (H1T) sW[i] = p.wcells[((i >> 6) * 2) * 64 + 4 * (i & 15) + ((i >> 4) & 3)];      // high plane; slot 16 j + l15 <- channel 4 l15 + j
        else sW[i] = p.wcells[i];
    }
    const float post = 1.f / pow2_scale_for(p.w_absmax, lane);
    const float oscale = p.out_prev ? shadow_scale_for(p.out_prev, lane) : p.out_scale;
    const int nrb = (NRB - wave + 7) / 8;                          // row blocks wave, wave + 8, wave + 16 below NRB (23 blocks: 3 each, wave 7: 2; H1T, 15: 2 each, wave 7: 1)
    float sc4[4][1], bi4[4][1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = H1T ? 4 * l15 + j : 16 * j + l15;
        sc4[j][0] = p.bn_scale[ch] * post; bi4[j][0] = p.bn_bias[ch];
    }
    float amax = 0.f;
    // the next tile's input pixels travel in registers: their global loads are issued before the GEMM and land under it.  A thread owns FOUR
    // consecutive pixels of one input row (12 bytes): ONE aligned 16-byte buffer load + three v_alignbyte (round 6; 12 byte loads before).
    // Bytes of rows / columns outside the frame are whatever the batch holds there (or zeros past its ends): phase 1 masks those pixels.
    constexpr int NG = IC / 4;                                      // four-pixel groups per input row
    static_assert(IC % 4 == 0 && IR * NG <= 512, "one group per thread");
    const int fr_r = t / NG, fr_c = 4 * (t - fr_r * NG);           // this thread's row and first column inside the tile's input window
    const bool fr_on = t < IR * NG;
    const __amdgpu_buffer_rsrc_t rs_fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.frames), 0, p.fr_bytes & ~3, 0x00020000);
    // (a second register set = two tiles of lookahead measured no gain: phase 1 does not wait for the frame bytes)
    unsigned pix[3] = {0u, 0u, 0u};                                 // byte 3 k + c = channel c of pixel k
    auto fetch = [&](int tile) {
        const int n = tile / (p.tiles_h * p.tiles_w), rem = tile - n * (p.tiles_h * p.tiles_w);
        const int ir0 = 2 * (2 * (rem / p.tiles_w) * PH - p.pbh) - 3, ic0 = 2 * (2 * (rem % p.tiles_w) * PW - p.pbw) - 3;
        const int gr = ir0 + fr_r, gc = ic0 + fr_c;
        if (!fr_on || (unsigned)gr >= (unsigned)p.H) return;       // (masked in phase 1)
        const int start = ((n * p.H + gr) * p.W + gc) * 3 + p.fr_delta;
        if (gc >= 0 && (start & ~3) + 16 <= (p.fr_bytes & ~3)) {
            const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs_fr, start & ~3, 0, 0);
            const unsigned sh = (unsigned)start & 3u;
            pix[0] = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
            pix[1] = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
            pix[2] = __builtin_amdgcn_alignbyte(d[3], d[2], sh);
        } else {                                                    // the group straddles the frame's left edge or the batch's last bytes: byte loads of the pixels inside
            unsigned b[12];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * k + c] = (gc + k >= 0 && start + 3 * k + c < p.fr_bytes) ? (unsigned)p.frames[start + 3 * k + c] : 0u;
#pragma unroll
            for (int w = 0; w < 3; ++w) pix[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
        }
    };
#ifndef DGP_SX
#define DGP_SX 0      // timing-only ablations of the phases (scripts/ablate_stem.sh; results are garbage): 1 no input fetch, 2 no phase 1,
#endif                // 4 no MFMAs, 8 no phase-3 tile store, 16 no pooling reads, 32 no global stores
    if (!(DGP_SX & 1) && (int)blockIdx.x < p.ntiles) fetch(blockIdx.x);
    if (DGP_SX & 1) { pix[0] = t; pix[1] = wave; pix[2] = lane; }
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int n = tile / (p.tiles_h * p.tiles_w), rem = tile - n * (p.tiles_h * p.tiles_w);
        const int ph0 = (rem / p.tiles_w) * PH, pw0 = (rem % p.tiles_w) * PW;
        const int r0 = 2 * ph0 - p.pbh, c0 = 2 * pw0 - p.pbw;      // first conv1 pixel of the tile
        const int ir0 = 2 * r0 - 3, ic0 = 2 * c0 - 3;              // first input pixel
        // ---- phase 1: input pixels (already in registers) -> fp16 high / low planes
        if (!(DGP_SX & 2) && fr_on) {
            const int gr = ir0 + fr_r, gc = ic0 + fr_c;
            const bool rok = (unsigned)gr < (unsigned)p.H;
            unsigned w[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // bytes 3 k .. 3 k + 2 of the 12
                const unsigned b0 = (pix[(3 * k) >> 2] >> (8 * ((3 * k) & 3))) & 255u, b1 = (pix[(3 * k + 1) >> 2] >> (8 * ((3 * k + 1) & 3))) & 255u,
                               b2 = (pix[(3 * k + 2) >> 2] >> (8 * ((3 * k + 2) & 3))) & 255u;
                half2v x01 = {(_Float16)0.f, (_Float16)0.f}, x23 = x01;
                if (rok && (unsigned)(gc + k) < (unsigned)p.W) {
                    x01 = half2v{(_Float16)((float)b0 - p.mean0), (_Float16)((float)b1 - p.mean1)};
                    x23 = half2v{(_Float16)((float)b2 - p.mean2), (_Float16)1.f};
                }
                w[2 * k] = __builtin_bit_cast(unsigned, x01); w[2 * k + 1] = __builtin_bit_cast(unsigned, x23);
            }
            uint4* dst = reinterpret_cast<uint4*>(sHi + fr_r * IC + fr_c);          // 4 pixels x 8 bytes, 32-byte aligned
            dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
            dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        __syncthreads();
        if (!(DGP_SX & 1) && tile + (int)gridDim.x < p.ntiles) fetch(tile + gridDim.x);
        // ---- phase 2: GEMM
        floatx4 acc[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        int abase[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int m = 16 * (wave + 8 * i) + l15;
            if (m > MS - 1) m = MS - 1;
            const int rl = m / SC, cl = m - rl * SC;
            abase[i] = (2 * rl) * IC + 2 * cl + 2 * g;             // pixel index of this lane's k-group in kernel row 0
        }
#pragma unroll 1
        for (int kh = 0; kh < 7; ++kh) {
            uint4 bh[4], bl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[j] = sW[((kh * 4 + g) * NPL + 0) * 64 + 16 * j + l15];
                if constexpr (!H1T) bl[j] = sW[((kh * 4 + g) * NPL + 1) * 64 + 16 * j + l15];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i < nrb) {
                    const uint4 ah = *reinterpret_cast<const uint4*>(sHi + abase[i] + kh * IC);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (DGP_SX & 4) { acc[i][j][0] += __builtin_bit_cast(float, ah.x ^ bh[j].y ^ bh[j].z); continue; }
                        if constexpr (!H1T)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah), __builtin_bit_cast(half8, bl[j]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ah), __builtin_bit_cast(half8, bh[j]), acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (ALIAS) __syncthreads();                     // every wave is done with the planes: the conv1 tile overwrites them (else: its own buffer; the
                                                                  // barrier behind phase 1 already says that every wave has left the previous tile's pool phase)
        // ---- phase 3: BN + ReLU -> LDS tile
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < nrb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * (wave + 8 * i) + 4 * g + r;
                    const int rl = m / SC, cl = m - rl * SC;
                    const bool ok = m < MS && (unsigned)(r0 + rl) < (unsigned)p.H1 && (unsigned)(c0 + cl) < (unsigned)p.W1;
                    if constexpr (H1T) {
                        float v4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v4[j] = ok ? fmaxf(acc[i][j][r] * sc4[j][0] + bi4[j][0], 0.f) : 0.f;
                            amax = fmaxf(amax, v4[j]);
                        }
                        unsigned h01, h23;                         // h1_pack8's rounding: fp16(scale * v)
                        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h01) : "v"(oscale), "v"(v4[0]));
                        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(oscale), "v"(v4[1]));
                        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h23) : "v"(oscale), "v"(v4[2]));
                        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(oscale), "v"(v4[3]));
                        if (!(DGP_SX & 8)) *reinterpret_cast<uint2*>(sCh + m * LDH + 4 * l15) = make_uint2(h01, h23);
                        continue;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = fmaxf(acc[i][j][r] * sc4[j][0] + bi4[j][0], 0.f);
                        if (DGP_SX & 8) { amax = fmaxf(amax, v); continue; }
                        sC[m * LDC + 16 * j + l15] = ok ? v : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 4: 3 x 3 / 2 max-pool of the tile, H2 cells out
        for (int q = t; q < PH * PW * 8; q += 512) {
            const int pp = q >> 3, cg = q & 7;
            const int ph = pp / PW, pw = pp - ph * PW;
            if constexpr (H1T) {
                half8 mx = {0, 0, 0, 0, 0, 0, 0, 0};
                unsigned kk[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
                if (p.idx) {                  // training step: the first maximum of the tier's OWN (fp16) values, strictly greater as in the fp32 kernel
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) {
                            const half8 x = *reinterpret_cast<const half8*>(sCh + ((2 * ph + a) * SC + 2 * pw + b) * LDH + 8 * cg);
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                if (x[k] > mx[k]) { mx[k] = x[k]; kk[k] = (unsigned)(a * 3 + b); }
                        }
                } else {
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        if (DGP_SX & 16) { mx[0] = (_Float16)(float)(a + b + q); continue; }
                        mx = __builtin_elementwise_max(mx, *reinterpret_cast<const half8*>(sCh + ((2 * ph + a) * SC + 2 * pw + b) * LDH + 8 * cg));
                    }
                }
                if (ph0 + ph < p.HP && pw0 + pw < p.WP) {
                    const size_t cell = (((size_t)n * p.HP + ph0 + ph) * p.WP + pw0 + pw) * 8 + cg;
                    if (p.idx)
                        *reinterpret_cast<uint2*>(p.idx + cell * 8) = make_uint2(kk[0] | (kk[1] << 8) | (kk[2] << 16) | (kk[3] << 24),
                                                                                 kk[4] | (kk[5] << 8) | (kk[6] << 16) | (kk[7] << 24));
                    if (!(DGP_SX & 32)) reinterpret_cast<uint4*>(p.out)[cell] = __builtin_bit_cast(uint4, mx);
                    else amax = fmaxf(amax, (float)mx[0]);
                }
                continue;
            }
            float v[8];
            unsigned kk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = 0.f; kk[k] = 0u; }
            if (p.idx) {                      // (strictly greater: the first maximum stays, as in maxpool_fwd_idx_kernel; same maxima as below)
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const float* src = sC + ((2 * ph + a) * SC + 2 * pw + b) * LDC + 8 * cg;
                        const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
                        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (x[k] > v[k]) { v[k] = x[k]; kk[k] = (unsigned)(a * 3 + b); }
                    }
            } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float* src = sC + ((2 * ph + a) * SC + 2 * pw + b) * LDC + 8 * cg;
                    if (DGP_SX & 16) { v[0] = fmaxf(v[0], (float)(a + b + q)); continue; }
                    const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
                    v[0] = fmaxf(v[0], x0.x); v[1] = fmaxf(v[1], x0.y); v[2] = fmaxf(v[2], x0.z); v[3] = fmaxf(v[3], x0.w);
                    v[4] = fmaxf(v[4], x1.x); v[5] = fmaxf(v[5], x1.y); v[6] = fmaxf(v[6], x1.z); v[7] = fmaxf(v[7], x1.w);
                }
            }
            if (ph0 + ph < p.HP && pw0 + pw < p.WP) {
                uint4 hi, lo;
                h2_pack8(v, oscale, hi, lo);
                const size_t cell = (((size_t)n * p.HP + ph0 + ph) * p.WP + pw0 + pw) * 8 + cg;       // 8-channel group of the pool output
                if (p.idx)
                    *reinterpret_cast<uint2*>(p.idx + cell * 8) = make_uint2(kk[0] | (kk[1] << 8) | (kk[2] << 16) | (kk[3] << 24),
                                                                             kk[4] | (kk[5] << 8) | (kk[6] << 16) | (kk[7] << 24));
                uint4* dst = reinterpret_cast<uint4*>(p.out) + (p.out_h1 ? cell : 2 * cell);
                if (!(DGP_SX & 32)) { dst[0] = hi; if (!p.out_h1) dst[1] = lo; } else amax = fmaxf(amax, __builtin_bit_cast(float, hi.x ^ lo.y));
#pragma unroll
                for (int k = 0; k < 8; ++k) amax = fmaxf(amax, v[k]);
            }
        }
        if constexpr (ALIAS) __syncthreads();                     // the next tile's planes overwrite the conv1 tile
    }
    if (p.out_absmax) track_absmax(p.out_absmax, amax, lane, (int)(blockIdx.x * 8u + (threadIdx.x >> 6)));
}

hipError_t launch_stem_pool_fused(const unsigned char* frames, int B, int H, int W, const void* wcells, const float* w_absmax,
                                  const float* bn_scale, const float* bn_bias, float m0, float m1, float m2, float out_scale,
                                  float* out, float* out_absmax, hipStream_t s, int out_h1, const float* out_prev, unsigned char* idx) {
    if ((long long)B * H * W * 3 >= (1LL << 31)) return hipErrorInvalidValue;      // (the frame batch is one buffer resource: 32-bit byte offsets)
    StemPoolArgs a{};
    a.out_h1 = out_h1; a.out_prev = out_prev; a.idx = idx;
    a.fr_delta = (int)(reinterpret_cast<uintptr_t>(frames) & 3u); a.frames = frames - a.fr_delta; a.fr_bytes = a.fr_delta + B * H * W * 3;
    a.wcells = reinterpret_cast<const uint4*>(wcells); a.w_absmax = w_absmax; a.bn_scale = bn_scale; a.bn_bias = bn_bias;
    a.mean0 = roundf(m0); a.mean1 = roundf(m1); a.mean2 = roundf(m2); a.out_scale = out_scale; a.out = out; a.out_absmax = out_absmax;
    a.B = B; a.H = H; a.W = W; a.H1 = (H + 1) / 2; a.W1 = (W + 1) / 2;
    a.HP = (a.H1 + 1) / 2; a.WP = (a.W1 + 1) / 2;
    const int pth = ((a.HP - 1) * 2 + 3 - a.H1) > 0 ? ((a.HP - 1) * 2 + 3 - a.H1) : 0;
    const int ptw = ((a.WP - 1) * 2 + 3 - a.W1) > 0 ? ((a.WP - 1) * 2 + 3 - a.W1) : 0;
    a.pbh = pth / 2; a.pbw = ptw / 2;
    // the two-workgroup variant: the 16-bit tier's inference pass (the trainer's pool records the first maximum of the fp32 values)
    static const int h1t_env = dgp_tune("DGP_STEM_H1T", 1);
    static const int h1t_train_env = dgp_tune("DGP_STEM_H1T_TRAIN", 1);
    const bool h1t = h1t_env && out_h1 && (!idx || h1t_train_env);      // (training: the pool's first-maximum record is taken on the tier's own fp16 values)
    const int ph = h1t ? 3 : DGP_STEM_PH;
    a.tiles_h = (a.HP + ph - 1) / ph; a.tiles_w = (a.WP + 15) / 16; a.ntiles = B * a.tiles_h * a.tiles_w;
    const size_t smem = h1t ? (size_t)7 * 4 * 64 * 16 + (size_t)19 * 72 * 8 + (size_t)15 * 16 * 72 * 2 : (size_t)7 * 4 * 2 * 64 * 16 + (DGP_STEM_PH >= 5 ? (size_t)23 * 16 * 68 * 4 : (size_t)(4 * DGP_STEM_PH + 7) * 72 * 8 + (size_t)(((2 * DGP_STEM_PH + 1) * 33 + 15) / 16) * 16 * 68 * 4);
    static bool attr_dev[16][2] = {};
    bool& attr = attr_dev[dgp_device_slot()][h1t ? 1 : 0];
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(h1t ? reinterpret_cast<const void*>(stem_pool_fused_kernel<true>) : reinterpret_cast<const void*>(stem_pool_fused_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    if (h1t) {
        const int grid = a.ntiles < 2 * n_cu ? a.ntiles : 2 * n_cu;
        hipLaunchKernelGGL(stem_pool_fused_kernel<true>, dim3((unsigned)grid), dim3(512), smem, s, a);
    } else {
        const int grid = a.ntiles < n_cu ? a.ntiles : n_cu;
        hipLaunchKernelGGL(stem_pool_fused_kernel<false>, dim3((unsigned)grid), dim3(512), smem, s, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// DGP 2-D soft-argmax + likelihood window.  One workgroup per (frame, joint) map.
//   p = softmax(gamma * s) over H*W; zero-pad by gauss_len; depthwise blur with
//   outer(g, g), g = exp(-x^2 / (2 sigma^2)) / sum, x = -r..r, r = sigma = gauss_len;
//   renormalise; mu = E[(h, w)].       (DGP/models/fitdgp_util.py:281-402)
//   likelihood: window [floor(mu), ceil(mu)+1) per axis clipped to the map, first row-major
//   arg-max of e^x/(e^x+1) on the RAW logits.        (DGP/models/eval.py:331-343)
// The map lives in LDS (H*W floats); sums are carried in fp64.
// ------------------------------------------------------------------------------------
// LARGE (round 4): maps that do not fit the LDS (the reference's placeholders are [None, None, None, nj], DGP/models/fitdgp.py:1130-1142:
// any size).  The softmax values are then not cached but RECOMPUTED from global memory wherever the blur reads them -- the same
// expressions on the same inputs, so every output is bit-identical to what the LDS variant would give; (2 r + 1)^2 exps per cell
// instead of one, a fallback for frames beyond ~1920 x 1280, not a fast path.
constexpr int SOFT_ARGMAX_THREADS = 1024;      // (256 until round 6: one workgroup per map is a latency chain -- 19 iterations per thread and pass on a 60 x 80 map)
template <bool LARGE>
__global__ __launch_bounds__(SOFT_ARGMAX_THREADS) void soft_argmax_kernel(const float* __restrict__ scmap, int H, int W, int C,
                                                          float gamma, int glen, float* __restrict__ mu,
                                                          float* __restrict__ conf, int* __restrict__ idx,
                                                          float* __restrict__ pmap, int rs) {
    // rs = elements per (frame, joint) record: 0 -> three dense arrays mu [.,2], conf [.], idx [.,2]; 5 -> mu / conf / idx point
    // into ONE packed [.,5] record (row, col, conf, iy, ix) -- the trajectory layout the RCCL all-gather moves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sp = reinterpret_cast<float*>(smem);       // H*W
    constexpr int NT = SOFT_ARGMAX_THREADS, NW = NT / 64;
    __shared__ double red[3][NW];
    __shared__ float redf[NW];
    __shared__ float gk[16];

    const int b = blockIdx.x / C;
    const int cj = blockIdx.x - b * C;
    const int HW = H * W;
    const float* src = scmap + (long long)b * HW * C + cj;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // Gaussian taps, fp32 like the reference graph
    const int r = glen;
    if (t == 0) {
        float s = 0.f;
        for (int i = -r; i <= r; ++i) {
            const float xs = (float)i / (float)glen;
            gk[i + r] = expf(-0.5f * xs * xs);
            s += gk[i + r];
        }
        for (int i = 0; i <= 2 * r; ++i) gk[i] = gk[i] / s;
    }

    // gamma * s ROUNDED to fp32 in both variants: the LDS variant stores the product before it subtracts the maximum, so the streaming
    // variant must not let hipcc (-ffp-contract=fast; __fmul_rn is a plain multiply in HIP) fuse product and subtraction into one fma.
    // The empty asm makes the product opaque to the optimiser; it costs no instruction.
    auto scaled = [&](int i) -> float { float v = src[(long long)i * C] * gamma; asm volatile("" : "+v"(v)); return v; };
    float mx = -INFINITY;
    for (int i = t; i < HW; i += NT) {
        const float v = scaled(i);
        if (!LARGE) sp[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    mx = redf[0];
#pragma unroll
    for (int k = 1; k < NW; ++k) mx = fmaxf(mx, redf[k]);

    double se = 0.0;
    for (int i = t; i < HW; i += NT) {
        const float e = expf((LARGE ? scaled(i) : sp[i]) - mx);
        if (!LARGE) sp[i] = e;
        se += (double)e;
    }
    se = wave_sum(se);
    if (lane == 0) red[0][wave] = se;
    __syncthreads();
    double dsum = red[0][0];
#pragma unroll
    for (int k = 1; k < NW; ++k) dsum += red[0][k];
    const float denom = (float)dsum;
    __syncthreads();
    if (!LARGE) {
        for (int i = t; i < HW; i += NT) sp[i] = sp[i] / denom;    // tf.nn.softmax output
    }
    __syncthreads();
    auto P = [&](int i) -> float { return LARGE ? expf(scaled(i) - mx) / denom : sp[i]; };

    // blur (zero padded) + moments
    double s0 = 0.0, sh = 0.0, sw = 0.0;
    for (int i = t; i < HW; i += NT) {
        const int h = i / W, w = i - h * W;
        float acc = 0.f;
        for (int a = -r; a <= r; ++a) {
            const int hh = h + a;
            if ((unsigned)hh >= (unsigned)H) continue;
            const float ga = gk[a + r];
            for (int bb = -r; bb <= r; ++bb) {
                const int ww = w + bb;
                if ((unsigned)ww >= (unsigned)W) continue;
                acc += (ga * gk[bb + r]) * P(hh * W + ww);
            }
        }
        s0 += (double)acc;
        sh += (double)acc * (double)h;
        sw += (double)acc * (double)w;
        if (pmap) pmap[((long long)b * HW + i) * C + cj] = acc;   // un-normalised; fixed below
    }
    s0 = wave_sum(s0); sh = wave_sum(sh); sw = wave_sum(sw);
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = sh; red[2][wave] = sw; }
    __syncthreads();
    double t0 = red[0][0], th = red[1][0], tw = red[2][0];
#pragma unroll
    for (int k = 1; k < NW; ++k) { t0 += red[0][k]; th += red[1][k]; tw += red[2][k]; }
    if (pmap) {
        const float inv_src = (float)t0;
        for (int i = t; i < HW; i += NT) {
            const long long o = ((long long)b * HW + i) * C + cj;
            pmap[o] = pmap[o] / inv_src;
        }
    }
    if (t == 0) {
        const float mh = (float)(th / t0);
        const float mw = (float)(tw / t0);
        const long long o = (long long)b * C + cj;
        const long long om = rs ? o * rs : o * 2, oc = rs ? o * rs : o;
        mu[om + 0] = mh;
        mu[om + 1] = mw;
        // likelihood window on raw logits
        int h0 = (int)floorf(mh), h1 = (int)ceilf(mh) + 1;
        int w0 = (int)floorf(mw), w1 = (int)ceilf(mw) + 1;
        if (h1 > H) h1 = H;
        if (w1 > W) w1 = W;
        if (h0 < 0) h0 = 0;
        if (w0 < 0) w0 = 0;
        float best = -1.f;
        int bh = h0, bw = w0;
        for (int hh = h0; hh < h1; ++hh)
            for (int ww = w0; ww < w1; ++ww) {
                const float x = src[(long long)(hh * W + ww) * C];
                const float e = expf(x);
                const float sg = e / (e + 1.f);          // x > 88.7: inf / inf = NaN, exactly like the reference's numpy expression
                // np.argmax treats NaN as the maximum and returns the FIRST one (eval.py:340-343): a NaN wins once and stays
                if (sg > best || (sg != sg && best == best)) { best = sg; bh = hh; bw = ww; }
            }
        conf[oc] = best;
        idx[om + 0] = bh;
        idx[om + 1] = bw;
    }
}

hipError_t launch_soft_argmax(const float* scmap, int B, int H, int W, int C, float gamma, int gauss_len,
                              float* mu, float* conf, int* idx, float* pmap, hipStream_t s, int record_stride) {
    const size_t smem = (size_t)H * W * sizeof(float);
    const char* force = getenv("DGP_SOFTARGMAX_STREAM");      // tests: the streaming variant on a map the LDS variant also takes (read per call)
    if (smem > SOFT_ARGMAX_LDS_LIMIT || (force && atoi(force) != 0)) {      // the map does not fit the LDS: streaming variant (same arithmetic, bit-identical)
        hipLaunchKernelGGL(soft_argmax_kernel<true>, dim3((unsigned)(B * C)), dim3(SOFT_ARGMAX_THREADS), 0, s, scmap, H, W, C, gamma, gauss_len, mu, conf, idx,
                           pmap, record_stride);
        return hipGetLastError();
    }
    static size_t attr_set_dev[16] = {};
    size_t& attr_set = attr_set_dev[dgp_device_slot()];
    if (smem > 64 * 1024 && smem > attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(soft_argmax_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_set = smem;
    }
    hipLaunchKernelGGL(soft_argmax_kernel<false>, dim3((unsigned)(B * C)), dim3(SOFT_ARGMAX_THREADS), smem, s, scmap, H, W, C, gamma,
                       gauss_len, mu, conf, idx, pmap, record_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// DLC hard arg-max (PET/nnet/predict.py:62-77): first row-major maximum of
// sigmoid(scmap[:, :, j]); returns index, probability and the raw locref pair there.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hard_argmax_kernel(const float* __restrict__ scmap,
                                                          const float* __restrict__ locref, int H, int W, int C,
                                                          int* __restrict__ idx, float* __restrict__ prob,
                                                          float* __restrict__ offs) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    const int b = blockIdx.x / C;
    const int cj = blockIdx.x - b * C;
    const int HW = H * W;
    const float* src = scmap + (long long)b * HW * C + cj;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int i = t; i < HW; i += 256) {
        const float x = src[(long long)i * C];
        const float sg = 1.f / (1.f + expf(-x));      // tf.sigmoid in fp32 (pose_net.py:86)
        if (sg > bv) { bv = sg; bi = i; }              // i increases: keeps the first maximum
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < 4; ++k)
            if (rv[k] > bv || (rv[k] == bv && ri[k] < bi)) { bv = rv[k]; bi = ri[k]; }
        const long long o = (long long)b * C + cj;
        const int h = bi / W, w = bi - h * W;
        idx[o * 2 + 0] = h;
        idx[o * 2 + 1] = w;
        prob[o] = bv;
        float dx = 0.f, dy = 0.f;
        if (locref) {
            const float* l = locref + ((long long)b * HW + bi) * (2 * C) + 2 * cj;
            dx = l[0];
            dy = l[1];
        }
        offs[o * 2 + 0] = dx;
        offs[o * 2 + 1] = dy;
    }
}

hipError_t launch_hard_argmax(const float* scmap, const float* locref, int B, int H, int W, int C, int* idx,
                              float* prob, float* offs, hipStream_t s) {
    hipLaunchKernelGGL(hard_argmax_kernel, dim3((unsigned)(B * C)), dim3(256), 0, s, scmap, locref, H, W, C, idx,
                       prob, offs);
    return hipGetLastError();
}

// argmax_2d_from_cm's `th` branch (fitdgp_util.py:377-388) on the normalised blurred softmax the soft-argmax kernel wrote:
// per (frame, joint) map: m = max p; p < m th -> 0; renormalise; expectation of the (row, col) grid.  One block per map, three
// strided passes over H W values (the map is a column of the [B,H,W,C] tensor: stride C).
__global__ __launch_bounds__(256) void pmap_threshold_kernel(float* __restrict__ pmap, int H, int W, int C, float th,
                                                             float* __restrict__ mu) {
    // The sums run in double: a relative error d of the normaliser moves mu by d x mu, and 200 fp32 additions per thread on a 250 x 250 map
    // left mu 1.3-2.1e-3 px from float64 (scripts/fuzz_readout.py); the stored map and mu stay fp32 like the reference's.
    __shared__ float red[4];
    __shared__ double redd[3][4];
    const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float* p = pmap + (size_t)b * H * W * C + c;
    const int n = H * W;
    float m = 0.f;                                         // p >= 0
    for (int i = t; i < n; i += 256) m = fmaxf(m, p[(size_t)i * C]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float cut = m * th;
    double sd = 0.0;
    for (int i = t; i < n; i += 256) { const float v = p[(size_t)i * C]; sd += v < cut ? 0.0 : (double)v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sd += __shfl_xor(sd, o, 64);
    if (lane == 0) redd[0][wave] = sd;
    __syncthreads();
    const float s = (float)((redd[0][0] + redd[0][1]) + (redd[0][2] + redd[0][3]));
    double sh = 0.0, sw = 0.0;
    for (int i = t; i < n; i += 256) {
        const float v = p[(size_t)i * C];
        const float q = (v < cut ? 0.f : v) / s;           // (+ 1e-100 in the reference: 0 in fp32; an all-zero map gives NaN there too)
        p[(size_t)i * C] = q;
        const int h = i / W, w = i - h * W;
        sh += (double)q * (double)h; sw += (double)q * (double)w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sh += __shfl_xor(sh, o, 64); sw += __shfl_xor(sw, o, 64); }
    if (lane == 0) { redd[1][wave] = sh; redd[2][wave] = sw; }
    __syncthreads();
    if (t == 0) {
        mu[((size_t)b * C + c) * 2 + 0] = (float)((redd[1][0] + redd[1][1]) + (redd[1][2] + redd[1][3]));
        mu[((size_t)b * C + c) * 2 + 1] = (float)((redd[2][0] + redd[2][1]) + (redd[2][2] + redd[2][3]));
    }
}

hipError_t launch_pmap_threshold(float* pmap, int B, int H, int W, int C, float th, float* mu, hipStream_t s) {
    hipLaunchKernelGGL(pmap_threshold_kernel, dim3(C, B), dim3(256), 0, s, pmap, H, W, C, th, mu);
    return hipGetLastError();
}

}  // namespace dgp
