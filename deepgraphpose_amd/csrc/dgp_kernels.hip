// gfx950 (MI355X / CDNA4): the convolution kernels of the Deep Graph Pose hot path (the other kernels: dgp_ops.hip).
//
//   conv_igemm_f32       implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chains), NHWC
//                        activations, k-chunked weight panels, LDS-staged double-buffered tiles, fused BN-affine / bias / residual /
//                        ReLU epilogue, optional transposed-conv phase scatter.  Covers K2, K4-K8 of SURVEY.md 2.3 under DGP_CONV_MODE=f32.
//   conv_igemm_f32_ls    the same with loader / compute wave specialisation
//   conv_igemm_split_ls  the dominant kernel: fp32-class products as 3 fp16 (or 6 bf16) MFMAs on the 16-bit matrix pipe; MODE 1 / 2 / 3 =
//                        per-tap loaders, pointwise loaders, halo walk; H2 activations in and out (ls_epilogue_h2)
//   tail_fixup*, launch_conv, pick_tile, conv_kernel_name
//
// Wavefront = 64 lanes everywhere.  No CUDA-compat shims; this file only targets gfx950.
#include "dgp_internal.h"
#include "dgp_device.h"
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <type_traits>

namespace dgp {

#ifdef DGP_DIAG
// diagnostic build only (scripts/diag.sh): s_memtime stamps fence the schedule; read SHARES, not totals
#define DIAG_STAMP(x)                                                                   \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#else
#define DIAG_STAMP(x)
#endif

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------
// Implicit-GEMM convolution
//
// GEMM view: out[m][co] = sum_k A[m][k] * Wt[k][co]
//   m  = (n, ho, wo) output pixel, k = (tap, ci) with ci fastest, taps row-major (kh, kw).
// K is walked in steps of BK = 32 floats = 8 chunks of 4 consecutive input channels.
// A chunk q = ks*8 + c maps to tap = q / (Cin/4), channel offset (q % (Cin/4)) * 4, so one
// K-step is one tap when Cin >= 32 and spans 8 taps for the Cin = 4 stem.
//
// MFMA operand trick: v_mfma_f32_32x32x2_f32 takes A[i = lane&31][k = lane>>5]; the order of
// k inside the reduction is free as long as A and B agree, so lanes 0-31 own chunk 2*kc and
// lanes 32-63 chunk 2*kc+1 and fetch their 4 k-values with ONE ds_read_b128; register j of
// both halves then feeds MFMA j.  LDS images are [chunk][row][4 floats]: a 32-lane half
// reads 512 contiguous bytes -> conflict-free ds_read_b128.  The A image pads each chunk
// plane by one 16-byte slot so that the 8 lanes that write one pixel's 8 chunks hit
// different banks.
// ------------------------------------------------------------------------------------

// WIDE = (Cin >= 32): all 8 chunks of a K-step lie in ONE tap, so tap / channel bookkeeping is
// wave-uniform (SALU, advanced incrementally) and each A load costs ~7 VALU ops.  The generic
// path (WIDE = false, used by the 4-channel stem) recomputes the tap per lane.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool WIDE>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_igemm_f32(const ConvArgs p) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;     // threads per workgroup (4 or 8 waves)
    constexpr int WM = BM / WAVES_M;       // wave tile rows
    constexpr int WN = BN / WAVES_N;       // wave tile cols
    constexpr int TM = WM / 32;
    constexpr int TN = WN / 32;
    constexpr int RG = NT / 8;             // pixel rows covered by one staging pass (8 chunks per row)
    constexpr int AROWS = BM / RG;         // A rows staged per thread
    constexpr int BSLOTS = 8 * BN / NT;    // B 16-byte slots staged per thread
    constexpr int LDA = BM + 1;            // slots per chunk plane of A
    constexpr int LDC = WN + 4;            // floats per row of a wave's epilogue staging tile
    static_assert(WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8, "4 or 8 waves per workgroup");
    static_assert(BSLOTS >= 1 && AROWS >= 1, "tile too small for the workgroup");
    static_assert(WAVES_M * WAVES_N * 32 * LDC * 4 <= (2 * 8 * LDA + 2 * 8 * BN) * 16, "epilogue staging must fit");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sA = reinterpret_cast<float4*>(smem);     // [2][8][LDA]
    float4* sB = sA + 2 * 8 * LDA;                    // [2][8][BN]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wave_m0 = (wave / WAVES_N) * WM;
    const int wave_n0 = (wave % WAVES_N) * WN;

    // XCD-aware tile mapping (blocks b and b+8 share an XCD / L2): give each XCD a
    // contiguous range of tile ids; inside it the n-tiles of one m-tile are adjacent so the
    // A tile is fetched into that L2 once.  Bijective for any grid size.
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // split-K (heads only): consecutive tile ids are the K-slices of one output tile
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int split = tile % ksplit;
    tile /= ksplit;
    const int nk_loc = p.nk / ksplit;
    const int ks_begin = split * nk_loc, ks_end = ks_begin + nk_loc;
    const int mt = tile / p.ntiles;
    const int nt = tile - mt * p.ntiles;
    const int m0 = mt * BM;
    const int n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, (int)p.w_bytes, 0x00020000);

    // ---- per-thread A row bookkeeping -------------------------------------------------
    const int c = t & 7;        // chunk within the K-step
    const int rg = t >> 3;      // 0..RG-1
    int hi0[AROWS], wi0[AROWS], pix0[AROWS], nbase[AROWS];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
        const int m = m0 + rg + RG * i;
        if (m < p.M) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int ho = rem / p.Wo;
            const int wo = rem - ho * p.Wo;
            hi0[i] = ho * p.stride - p.pad_t;
            wi0[i] = wo * p.stride - p.pad_l;
            pix0[i] = (n * p.H + hi0[i]) * p.W + wi0[i];
            nbase[i] = n * p.H * p.W;
        } else {
            hi0[i] = -(1 << 28);
            wi0[i] = -(1 << 28);
            pix0[i] = 0;
            nbase[i] = 0;
        }
    }
    const int cin4m1 = (p.Cin >> 2) - 1;
    const unsigned b_lane_off = (unsigned)(t / BN) * ((unsigned)p.CoutP * 16u) + (unsigned)((n0 + (t & (BN - 1))) * 16);
    const unsigned b_row_bytes = (unsigned)p.CoutP * 16u;
    int rowoff[AROWS];          // WIDE: byte offset of (pixel row, lane's chunk) at tap (0,0), channel 0
#pragma unroll
    for (int i = 0; i < AROWS; ++i) rowoff[i] = (pix0[i] * p.Cin + 4 * c) * 4;
    // wave-uniform tap walker (WIDE): K-step ks covers tap w_tap, channels [w_ch, w_ch + 32)
    int w_tap = (ks_begin * 8) >> p.log2cin4;
    int w_ch = ((ks_begin * 8) & cin4m1) << 2;
    int w_kh = w_tap / p.KW, w_kw = w_tap - w_kh * p.KW;

    float4 ra[AROWS];
    float4 rb[BSLOTS];

    auto gload = [&](int ks) {
        if constexpr (WIDE) {
            const int dh = w_kh * p.dil, dw = w_kw * p.dil;
            const int doff = ((dh * p.W + dw) * p.Cin + w_ch) * 4;
            const bool tapok = w_tap < p.ntaps;
            if (p.up == 2) {
                // data-gradient of a stride-2 conv: the input (dY) is read on the zero-stuffed grid, i.e. only
                // where the virtual coordinate is even:  dX[hi] += dY[(hi + pad' + kh' d) / 2] W[kh']
#pragma unroll
                for (int i = 0; i < AROWS; ++i) {
                    const int hv = hi0[i] + dh, wv = wi0[i] + dw;
                    const bool ok = tapok && !((hv | wv) & 1) && (unsigned)(hv >> 1) < (unsigned)p.H &&
                                    (unsigned)(wv >> 1) < (unsigned)p.W;
                    const unsigned off = (unsigned)((nbase[i] + (hv >> 1) * p.W + (wv >> 1)) * p.Cin + w_ch + 4 * c) << 2;
                    ra[i] = buf_load16(rs_in, ok ? off : OOB);
                }
            } else {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) {
                    const int hi = hi0[i] + dh, wi = wi0[i] + dw;
                    const bool ok = tapok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                    ra[i] = buf_load16(rs_in, ok ? (unsigned)(rowoff[i] + doff) : OOB);
                }
            }
            w_ch += 32;
            if (w_ch >= p.Cin) {
                w_ch = 0; ++w_tap;
                if (++w_kw == p.KW) { w_kw = 0; ++w_kh; }
            }
        } else {
            const int q = ks * 8 + c;
            const int tap = q >> p.log2cin4;
            const int ch = (q & cin4m1) << 2;
            const int kh = tap / p.KW;
            const int kw = tap - kh * p.KW;
            const int dh = kh * p.dil, dw = kw * p.dil;
            const bool tapok = tap < p.ntaps;
            const int doff = dh * p.W + dw;
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
                const int hi = hi0[i] + dh, wi = wi0[i] + dw;
                const bool ok = tapok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const unsigned off = ok ? ((unsigned)((pix0[i] + doff) * p.Cin + ch) << 2) : OOB;
                ra[i] = buf_load16(rs_in, off);
            }
        }
        const unsigned kbase = (unsigned)(ks * 8) * b_row_bytes;
#pragma unroll
        for (int i = 0; i < BSLOTS; ++i)
            rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        rs_w, (int)b_lane_off, (int)(kbase + (unsigned)(i * (NT / BN)) * b_row_bytes), 0));
    };
    auto lstore = [&](int buf) {
        float4* a = sA + buf * 8 * LDA + c * LDA + rg;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) a[RG * i] = ra[i];
        float4* b = sB + buf * 8 * BN + t;
#pragma unroll
        for (int i = 0; i < BSLOTS; ++i) b[NT * i] = rb[i];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef DGP_DIAG
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
#endif
#ifdef DGP_DIAG
    if (p.dbg) st0 = __builtin_amdgcn_s_memtime();
#endif
    gload(ks_begin);
    lstore(0);
    __syncthreads();
#ifdef DGP_DIAG
    if (p.dbg) st1 = __builtin_amdgcn_s_memtime();
#endif
    const int half = lane >> 5;
    const int l31 = lane & 31;
#ifdef DGP_DIAG
    unsigned long long d0, d1, d2, d3, d4, d5, acc_gl = 0, acc_mf = 0, acc_vm = 0, acc_ls = 0, acc_ba = 0;
#endif
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        const int buf = (ks - ks_begin) & 1;
        DIAG_STAMP(d0);
        if (ks + 1 < ks_end) gload(ks + 1);          // in flight under the MFMAs below
        DIAG_STAMP(d1);
        const float4* a_base = sA + buf * 8 * LDA + wave_m0 + l31;
        const float4* b_base = sB + buf * 8 * BN + wave_n0 + l31;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int chunk = 2 * kc + half;
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = a_base[chunk * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = b_base[chunk * BN + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        DIAG_STAMP(d2);
#ifdef DGP_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DIAG_STAMP(d3);
#endif
        if (ks + 1 < ks_end) lstore(buf ^ 1);
        DIAG_STAMP(d4);
        __syncthreads();
        DIAG_STAMP(d5);
#ifdef DGP_DIAG
        acc_gl += d1 - d0; acc_mf += d2 - d1; acc_vm += d3 - d2; acc_ls += d4 - d3; acc_ba += d5 - d4;
#endif
    }

#ifdef DGP_DIAG
    if (p.dbg) st2 = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue ----------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    if (p.out_mode == 1) {
        // transposed-conv phase scatter (heads only: tiny, per-element stores)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wave_n0 + 32 * j + l31;
            const bool cok = co < p.Cout;
            const float bi = (cok && p.bias && split == 0) ? p.bias[co] : 0.f;
            const int ph = co / p.dc_nj;
            const int cj = co - ph * p.dc_nj;
            const int ph_a = ph >> 1, ph_b = ph & 1;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int m = m0 + wave_m0 + 32 * i + row;
                    if (!cok || m >= p.M) continue;
                    const int n = m / HoWo;
                    const int rem = m - n * HoWo;
                    const int ho = rem / p.Wo;
                    const int wo = rem - ho * p.Wo;
                    const long long oidx =
                        (((long long)n * (2 * p.Ho) + 2 * ho + ph_a) * (2 * p.Wo) + 2 * wo + ph_b) * p.dc_nj + cj;
                    p.out[(long long)split * p.split_stride + oidx] = acc[i][j][r] + bi;
                }
            }
        }
        return;
    }

    // NHWC output: stage each wave's 32 x WN sub-tile through (now free) LDS so that global
    // traffic is 16 B per lane on full rows, with all residual loads of a pass in flight at once.
    float* sC = reinterpret_cast<float*>(smem) + wave * (32 * LDC);
    constexpr int C4 = WN / 4;                 // float4 per staged row
    constexpr int NV = 32 * C4 / 64;           // float4 per lane per pass
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.in), 0,
                                          p.res ? (int)p.res_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out =
        __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mask =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mask ? p.mask : p.in), 0,
                                          p.mask ? (int)p.out_bytes : 0, 0x00020000);
    const int my_c4 = lane % C4;
    const int my_r0 = lane / C4;
    const int co4 = n0 + wave_n0 + 4 * my_c4;
    const bool cok = co4 < p.Cout;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), bi4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok && p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + co4);
    if (cok && p.bias) bi4 = *reinterpret_cast<const float4*>(p.bias + co4);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __syncthreads();      // previous pass (or the main loop) is done with this LDS
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                sC[row * LDC + 32 * j + l31] = acc[i][j][r];
            }
        __syncthreads();
        unsigned ooff[NV];
        float4 rres[NV], rmask[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int row = my_r0 + v * (64 / C4);
            const int m = m0 + wave_m0 + 32 * i + row;
            const bool ok = cok && m < p.M;
            ooff[v] = ok ? ((unsigned)(m * p.Cout + co4) << 2) : OOB;
            unsigned roff = OOB;
            if (p.res_s == 1) {
                roff = ooff[v];
            } else if (p.res_s == -2 && ok) {        // residual lives on the 2x coarser grid (zero-stuffed upsample)
                const int n = m / HoWo;
                const int rem = m - n * HoWo;
                const int ho = rem / p.Wo;
                const int wo = rem - ho * p.Wo;
                if (!((ho | wo) & 1))
                    roff = (unsigned)(((n * p.res_H + (ho >> 1)) * p.res_W + (wo >> 1)) * p.Cout + co4) << 2;
            } else if (p.res_s > 1 && ok) {
                const int n = m / HoWo;
                const int rem = m - n * HoWo;
                const int ho = rem / p.Wo;
                const int wo = rem - ho * p.Wo;
                roff = (unsigned)(((n * p.res_H + ho * p.res_s) * p.res_W + wo * p.res_s) * p.Cout + co4) << 2;
            }
            rres[v] = buf_load16(rs_res, roff);      // zeros when there is no residual
            rmask[v] = buf_load16(rs_mask, ooff[v]);  // ReLU gate of the backward pass (zeros when unused)
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int row = my_r0 + v * (64 / C4);
            const float4 a = *reinterpret_cast<const float4*>(sC + row * LDC + 4 * my_c4);
            float4 o;
            o.x = a.x * sc4.x + bi4.x + rres[v].x;
            o.y = a.y * sc4.y + bi4.y + rres[v].y;
            o.z = a.z * sc4.z + bi4.z + rres[v].z;
            o.w = a.w * sc4.w + bi4.w + rres[v].w;
            if (p.relu) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            if (p.mask) {          // d/dx of ReLU: pass the gradient where the saved activation is positive
                o.x = rmask[v].x > 0.f ? o.x : 0.f; o.y = rmask[v].y > 0.f ? o.y : 0.f;
                o.z = rmask[v].z > 0.f ? o.z : 0.f; o.w = rmask[v].w > 0.f ? o.w : 0.f;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, (int)ooff[v], 0, 0);
            if (ooff[v] != OOB) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (p.out_absmax) track_absmax(p.out_absmax, amax, lane, (int)(blockIdx.x * 8u + (threadIdx.x >> 6)));
#ifdef DGP_DIAG
    if (p.dbg && t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st3 = __builtin_amdgcn_s_memtime();
        unsigned long long* d = p.dbg + 10ull * blockIdx.x;
        d[0] = st1 - st0; d[1] = st2 - st1; d[2] = st3 - st2;
        d[3] = acc_gl; d[4] = acc_mf; d[5] = acc_vm; d[6] = acc_ls; d[7] = acc_ba;
    }
#endif
}

// Epilogue of the loader-specialised kernels (compute waves only; wave-private LDS staging, no workgroup
// barriers: the loader waves are gone).  acc -> scale/bias/residual/ReLU/mask -> 16-byte stores.
// M16: the caller's accumulators are the 2 x 8 blocks of v_mfma_f32_16x16x32 and it has staged them into the wave's LDS tile itself
// (block (i, j) register r of lane l = row 16 i + 4 (l >> 4) + r, column 16 j + (l & 15)); `acc` is not read
#ifndef DGP_SHADOW_AUX
#define DGP_SHADOW_AUX 0
#endif
template <int TM, int TN, int WN, bool M16 = false>
__device__ __forceinline__ void ls_epilogue(const ConvArgs& p, floatx16 (&acc)[TM][TN], char* smem, int wave, int lane,
                                            int m0, int n0, int wave_m0, int wave_n0, float post = 1.f) {
    constexpr int LDC = WN + 4;
    const int half = lane >> 5, l31 = lane & 31;
    const int HoWo = p.Ho * p.Wo;
    float* sC = reinterpret_cast<float*>(smem) + wave * (32 * LDC);
    // the wave's range slot, read NOW (see ls_epilogue_h2: nothing may wait at the end of the wave)
    unsigned slot_bits = 0u;
    if (p.out_absmax)
        slot_bits = __hip_atomic_load(reinterpret_cast<const unsigned*>(p.out_absmax) + ((int)(blockIdx.x * 8u + (threadIdx.x >> 6)) & (ABSMAX_SLOTS - 1)),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int C4 = WN / 4;
    constexpr int NV = 32 * C4 / 64;
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.in), 0, p.res ? (int)p.res_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    // fp16 high / low copy of the output (ConvArgs::shadow): this lane's 4 channels are half of an 8-channel cell
    const float sh_scale = p.shadow ? shadow_scale_for(p.shadow_prev, lane) : 0.f;
    const __amdgpu_buffer_rsrc_t rs_sh =
        __builtin_amdgcn_make_buffer_rsrc(p.shadow ? p.shadow : p.out, 0, sh_scale > 0.f ? (int)(p.shadow_fmt == 2 ? p.out_bytes >> 1 : p.out_bytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mask =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.mask ? p.mask : p.in), 0, p.mask ? (int)p.out_bytes : 0, 0x00020000);
    const int my_c4 = lane % C4;
    const int my_r0 = lane / C4;
    const int co4 = n0 + wave_n0 + 4 * my_c4;
    const bool cok = co4 < p.Cout;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), bi4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok && p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + co4);
    if (cok && p.bias) bi4 = *reinterpret_cast<const float4*>(p.bias + co4);
    sc4.x *= post; sc4.y *= post; sc4.z *= post; sc4.w *= post;       // exact: post is a power of two (1 unless fp16-split)
    float amax = 0.f;
    // Rows go out in chunks of VC (register pressure: 8 rows of residual + gate next to the accumulators spilled them) and the
    // chunks are software-pipelined: the residual / gate loads of chunk q + 1 are in flight while chunk q is combined and stored,
    // so a wave pays one HBM round trip per tile instead of one per chunk.  The loads exist only when the layer has them.
    constexpr int VC = NV > 2 ? 2 : NV;
    constexpr int CPP = NV / VC;                   // chunks per pass (pass = one 32-row block of the wave tile)
    constexpr int NQ = TM * CPP;
    unsigned ooff[2][VC];
    float4 rres[2][VC];
    auto issue = [&](int q, int slot) {
        const int i = q / CPP, v0 = (q % CPP) * VC;
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            const int row = my_r0 + (v0 + u) * (64 / C4);
            const int m = m0 + wave_m0 + 32 * i + row;
            const bool ok = cok && m < p.M;
            ooff[slot][u] = ok ? ((unsigned)(m * p.Cout + co4) << 2) : OOB;
            rres[slot][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.res) {
                unsigned roff = OOB;
                if (p.res_s == 1) {
                    roff = ooff[slot][u];
                } else if (p.res_s == -2 && ok) {
                    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    if (!((ho | wo) & 1)) roff = (unsigned)(((n * p.res_H + (ho >> 1)) * p.res_W + (wo >> 1)) * p.Cout + co4) << 2;
                } else if (p.res_s > 1 && ok) {
                    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    roff = (unsigned)(((n * p.res_H + ho * p.res_s) * p.res_W + wo * p.res_s) * p.Cout + co4) << 2;
                }
                rres[slot][u] = buf_load16(rs_res, roff);
            }
        }
    };
#ifdef DGP_DIAG
    unsigned long long g0, g1, g2, g3, g4;
    DIAG_STAMP(g0);
#endif
    issue(0, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = q / CPP, v0 = (q % CPP) * VC, slot = q & 1;
#ifdef DGP_DIAG
        if (q == 1) DIAG_STAMP(g2);
#endif
        if (q % CPP == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own reads of the previous pass are done
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!M16) {      // (M16: the caller has staged its 16x16 blocks already)
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                        sC[row * LDC + 32 * j + l31] = acc[i][j][r];
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // a wave's LDS ops complete in order
#ifdef DGP_DIAG
            if (q == 0) DIAG_STAMP(g1);
#endif
        }
        if (q + 1 < NQ) issue(q + 1, slot ^ 1);
        float4 rmask[VC];          // ReLU gate of the training step's data-gradient convs: not pipelined (keeps the forward lean)
        if (p.mask) {
#pragma unroll
            for (int u = 0; u < VC; ++u) {
                if (p.mask_fmt == 2) {      // H1 gate tensor: this lane's 4 channels are 8 bytes at half the fp32 offset
                    const u32x2 mh = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_mask, (int)(ooff[slot][u] >> 1), 0, 0));
                    rmask[u] = h2_gate4(mh, u32x2{0u, 0u});
                } else if (p.mask_fmt) {      // H2 gate tensor: this lane's 4 channels are 8 bytes of the cell's high chunk and 8 of its low chunk
                    const unsigned mo = (ooff[slot][u] & ~31u) + ((ooff[slot][u] & 16u) >> 1);
                    // (bit_cast: the builtin's result converts to a vector by splatting its low dword)
                    const u32x2 mh = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_mask, (int)mo, 0, 0));
                    const u32x2 ml = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_mask, (int)(mo + 16u), 0, 0));
                    rmask[u] = h2_gate4(mh, ml);
                } else rmask[u] = buf_load16(rs_mask, ooff[slot][u]);
            }
        }
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            const int row = my_r0 + (v0 + u) * (64 / C4);
            const float4 a = *reinterpret_cast<const float4*>(sC + row * LDC + 4 * my_c4);
            float4 o;
            o.x = a.x * sc4.x + bi4.x + rres[slot][u].x;
            o.y = a.y * sc4.y + bi4.y + rres[slot][u].y;
            o.z = a.z * sc4.z + bi4.z + rres[slot][u].z;
            o.w = a.w * sc4.w + bi4.w + rres[slot][u].w;
            if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (p.mask) {
                o.x = rmask[u].x > 0.f ? o.x : 0.f; o.y = rmask[u].y > 0.f ? o.y : 0.f;
                o.z = rmask[u].z > 0.f ? o.z : 0.f; o.w = rmask[u].w > 0.f ? o.w : 0.f;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, (int)ooff[slot][u], 0, 0);
            if (sh_scale > 0.f && p.shadow_fmt == 2) {
                // H1 copy (the 16-bit trainer): this lane's 4 channels are 8 bytes of the tensor at half the fp32 offset
                uint2 sh, sl;
                split2_f16(o, sh_scale, sh, sl);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{sh.x, sh.y}, rs_sh, (int)(ooff[slot][u] >> 1), 0, DGP_SHADOW_AUX);
            } else if (sh_scale > 0.f) {
                // this lane holds 4 of a cell's 8 channels: trade halves with the neighbour (quad_perm 1,0,3,2) so that the even lane
                // stores the whole high chunk and the odd lane the whole low chunk -- one 16-byte store per lane, whole cells per pair
                uint2 sh, sl;
                split2_f16(o, sh_scale, sh, sl);
                const bool odd = my_c4 & 1;
                const unsigned g0 = odd ? sh.x : sl.x, g1 = odd ? sh.y : sl.y;
                const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)g0, 0xB1, 0xF, 0xF, true);
                const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)g1, 0xB1, 0xF, 0xF, true);
                const u32x4 cell = odd ? u32x4{r0, r1, sl.x, sl.y} : u32x4{sh.x, sh.y, r0, r1};
                __builtin_amdgcn_raw_buffer_store_b128(cell, rs_sh, (int)((ooff[slot][u] & ~31u) + (odd ? 16u : 0u)), 0, DGP_SHADOW_AUX);
            }
            if (ooff[slot][u] != OOB) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
#ifdef DGP_DIAG
    DIAG_STAMP(g3);
#endif
    if (p.out_absmax) track_absmax_known(p.out_absmax, amax, lane, (int)(blockIdx.x * 8u + (threadIdx.x >> 6)), slot_bits);
#ifdef DGP_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DIAG_STAMP(g4);
    if (p.dbg && threadIdx.x == 0) {      // (slots 3, 5, 6 are the loader stamps of the register-staged kernels: read these with DGP_DMA=1)
        unsigned long long* d = p.dbg + 10ull * blockIdx.x;
        d[3] = g1 - g0; d[5] = g2 - g1; d[6] = g3 - g2; d[8] = g4 - g3;
    }
#endif
}

// Raw accumulators of a wave tile -> a dense [rows][ld] fp32 matrix (K-split slabs of the grid's tail): same wave-private LDS
// staging as ls_epilogue, no epilogue arithmetic.
template <int TM, int TN, int WN, bool M16 = false>
__device__ __forceinline__ void ls_store_raw(floatx16 (&acc)[TM][TN], char* smem, int wave, int lane, int wave_m0, int wave_n0,
                                             float* out, int ld) {
    constexpr int LDC = WN + 4;
    constexpr int C4 = WN / 4;
    constexpr int NV = 32 * C4 / 64;
    const int half = lane >> 5, l31 = lane & 31;
    float* sC = reinterpret_cast<float*>(smem) + wave * (32 * LDC);
    const int my_c4 = lane % C4, my_r0 = lane / C4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!M16) sC[((r & 3) + 8 * (r >> 2) + 4 * half) * LDC + 32 * j + l31] = acc[i][j][r];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int row = my_r0 + v * (64 / C4);
            *reinterpret_cast<float4*>(out + (size_t)(wave_m0 + 32 * i + row) * ld + wave_n0 + 4 * my_c4) =
                *reinterpret_cast<const float4*>(sC + row * LDC + 4 * my_c4);
        }
    }
}

// ------------------------------------------------------------------------------------
// Loader-specialised variant (Cin >= 32, NHWC output): the workgroup has 4 COMPUTE waves (2 x 2 over the tile)
// that only issue ds_read_b128 + MFMA, and 4 LOADER waves that only issue buffer loads + ds_write.  A
// buffer_load costs the issuing wave hundreds of cycles under load (the L2 -> L1 path is the co-bottleneck
// of this kernel, see EXPERIMENTS.md); with the loads on their own waves that stall no longer sits in front of
// the MFMAs, and every load has a whole K-step (8k cycles) in flight before its LDS write.
//   step ks:  compute reads buf[ks&1]      | loaders write buf[(ks+1)&1] (loaded during step ks-1),
//                                          | then issue the loads of step ks+2;   one s_barrier per step.
// ------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(512, 4) void conv_igemm_f32_ls(const ConvArgs p) {
    constexpr int WM = BM / 2, WN = BN / 2;        // compute waves 2 x 2
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NLT = 256;                       // loader threads
    constexpr int RG = NLT / 8;
    constexpr int AROWS = BM / RG;
    constexpr int BSLOTS = 8 * BN / NLT;
    constexpr int LDA = BM + 1;
    constexpr int LDC = WN + 4;
    static_assert(4 * 32 * LDC * 4 <= (2 * 8 * LDA + 2 * 8 * BN) * 16, "epilogue staging must fit");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sA = reinterpret_cast<float4*>(smem);     // [2][8][LDA]
    float4* sB = sA + 2 * 8 * LDA;                    // [2][8][BN]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = tile / p.ntiles;
    const int nt = tile - mt * p.ntiles;
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    const int HoWo = p.Ho * p.Wo;

    if (wave >= 4) {
        // ================================ loader waves ================================
        const int t = threadIdx.x - 256;
        const __amdgpu_buffer_rsrc_t rs_in =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, (int)p.w_bytes, 0x00020000);
        const int c = t & 7, rg = t >> 3;
        int hi0[AROWS], wi0[AROWS], rowoff[AROWS], nbase[AROWS];
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            const int m = m0 + rg + RG * i;
            if (m < p.M) {
                const int n = m / HoWo;
                const int rem = m - n * HoWo;
                const int ho = rem / p.Wo;
                const int wo = rem - ho * p.Wo;
                hi0[i] = ho * p.stride - p.pad_t;
                wi0[i] = wo * p.stride - p.pad_l;
                rowoff[i] = (((n * p.H + hi0[i]) * p.W + wi0[i]) * p.Cin + 4 * c) * 4;
                nbase[i] = n * p.H * p.W;
            } else {
                hi0[i] = -(1 << 28); wi0[i] = -(1 << 28); rowoff[i] = 0; nbase[i] = 0;
            }
        }
        const unsigned b_lane_off = (unsigned)(t / BN) * ((unsigned)p.CoutP * 16u) + (unsigned)((n0 + (t & (BN - 1))) * 16);
        const unsigned b_row_bytes = (unsigned)p.CoutP * 16u;
        int w_kh = 0, w_kw = 0, w_ch = 0, w_tap = 0;
        // two register sets: the loads of step ks+3 are issued while those of step ks+2 are still in flight, so
        // every load has two K-steps to land (one K-step was not enough: the loaders, not the MFMAs, set the pace)
        float4 ra0[AROWS], rb0[BSLOTS], ra1[AROWS], rb1[BSLOTS];
        auto gload = [&](int ks, float4 (&ra)[AROWS], float4 (&rb)[BSLOTS]) {
            const int dh = w_kh * p.dil, dw = w_kw * p.dil;
            const int doff = ((dh * p.W + dw) * p.Cin + w_ch) * 4;
            const bool tapok = w_tap < p.ntaps;
            if (p.up == 2) {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) {
                    const int hv = hi0[i] + dh, wv = wi0[i] + dw;
                    const bool ok = tapok && !((hv | wv) & 1) && (unsigned)(hv >> 1) < (unsigned)p.H &&
                                    (unsigned)(wv >> 1) < (unsigned)p.W;
                    const unsigned off = (unsigned)((nbase[i] + (hv >> 1) * p.W + (wv >> 1)) * p.Cin + w_ch + 4 * c) << 2;
                    ra[i] = buf_load16(rs_in, ok ? off : OOB);
                }
            } else {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) {
                    const int hi = hi0[i] + dh, wi = wi0[i] + dw;
                    const bool ok = tapok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                    ra[i] = buf_load16(rs_in, ok ? (unsigned)(rowoff[i] + doff) : OOB);
                }
            }
            w_ch += 32;
            if (w_ch >= p.Cin) {
                w_ch = 0; ++w_tap;
                if (++w_kw == p.KW) { w_kw = 0; ++w_kh; }
            }
            // the walker is wave-uniform: say so, or hipcc keeps it in VGPRs / scratch behind exec-masked updates
            w_tap = __builtin_amdgcn_readfirstlane(w_tap); w_kw = __builtin_amdgcn_readfirstlane(w_kw);
            w_kh = __builtin_amdgcn_readfirstlane(w_kh); w_ch = __builtin_amdgcn_readfirstlane(w_ch);
            const unsigned kbase = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(ks * 8) * b_row_bytes));
#pragma unroll
            for (int i = 0; i < BSLOTS; ++i)
                rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                            rs_w, (int)b_lane_off, (int)(kbase + (unsigned)(i * (NLT / BN)) * b_row_bytes), 0));
        };
        auto lstore = [&](int buf, float4 (&ra)[AROWS], float4 (&rb)[BSLOTS]) {
            float4* a = sA + buf * 8 * LDA + c * LDA + rg;
#pragma unroll
            for (int i = 0; i < AROWS; ++i) a[RG * i] = ra[i];
            float4* b = sB + buf * 8 * BN + t;
#pragma unroll
            for (int i = 0; i < BSLOTS; ++i) b[NLT * i] = rb[i];
        };
        gload(0, ra0, rb0);
        lstore(0, ra0, rb0);
        if (p.nk > 1) gload(1, ra1, rb1);           // set 1 holds odd steps, set 0 even steps
        if (p.nk > 2) gload(2, ra0, rb0);
        __syncthreads();
        for (int ks = 0; ks < p.nk; ks += 2) {
            // step ks (even): publish step ks+1 (set 1), refill it with step ks+3
            if (ks + 1 < p.nk) lstore(1, ra1, rb1);
            if (ks + 3 < p.nk) gload(ks + 3, ra1, rb1);
            __syncthreads();
            if (ks + 1 >= p.nk) break;
            // step ks+1 (odd): publish step ks+2 (set 0), refill it with step ks+4
            if (ks + 2 < p.nk) lstore(0, ra0, rb0);
            if (ks + 4 < p.nk) gload(ks + 4, ra0, rb0);
            __syncthreads();
        }
        return;
    }

    // ================================== compute waves ==================================
    const int wave_m0 = (wave >> 1) * WM;
    const int wave_n0 = (wave & 1) * WN;
    const int half = lane >> 5, l31 = lane & 31;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#ifdef DGP_DIAG
    unsigned long long e0, e1, e2, e3, acc_mf = 0, acc_ba = 0;
    DIAG_STAMP(e0);
#endif
    __syncthreads();
#ifdef DGP_DIAG
    DIAG_STAMP(e1);
    const unsigned long long t_pro = e1 - e0;
#endif
    for (int ks = 0; ks < p.nk; ++ks) {
        const int buf = ks & 1;
        const float4* a_base = sA + buf * 8 * LDA + wave_m0 + l31;
        const float4* b_base = sB + buf * 8 * BN + wave_n0 + l31;
        DIAG_STAMP(e1);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int chunk = 2 * kc + half;
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = a_base[chunk * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = b_base[chunk * BN + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        DIAG_STAMP(e2);
        __syncthreads();
        DIAG_STAMP(e3);
#ifdef DGP_DIAG
        acc_mf += e2 - e1; acc_ba += e3 - e2;
#endif
    }
#ifdef DGP_DIAG
    if (p.dbg && threadIdx.x == 0) {
        unsigned long long* d = p.dbg + 10ull * blockIdx.x;
        d[0] = t_pro; d[1] = acc_mf + acc_ba; d[2] = 0; d[3] = 0; d[4] = acc_mf; d[5] = 0; d[6] = 0; d[7] = acc_ba;
    }
#endif

    ls_epilogue<TM, TN, WN>(p, acc, smem, wave, lane, m0, n0, wave_m0, wave_n0);
}

template <int BM, int BN>
static hipError_t launch_conv_ls(ConvArgs a, hipStream_t s) {
    const size_t smem = (size_t)(2 * 8 * (BM + 1) + 2 * 8 * BN) * 16;
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.CoutP + BN - 1) / BN;
    if (a.CoutP % BN != 0 || a.Cin < 32 || a.out_mode != 0) return hipErrorInvalidValue;
    auto kern = conv_igemm_f32_ls<BM, BN>;
    static bool attr_done_dev[16] = {};
    bool& attr_done = attr_done_dev[dgp_device_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const long long nwg = (long long)a.mtiles * a.ntiles;
#ifdef DGP_DIAG
    static unsigned long long* dbg_buf = nullptr;
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, 10 * 8 * 65536);
    a.dbg = nwg <= 65536 ? dbg_buf : nullptr;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(512), smem, s, a);
#ifdef DGP_DIAG
    if (a.dbg) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h(10 * nwg);
        (void)hipMemcpy(h.data(), a.dbg, 80 * nwg, hipMemcpyDeviceToHost);
        double v[8] = {0};
        for (long long b = 0; b < nwg; ++b) for (int k = 0; k < 8; ++k) v[k] += (double)h[10 * b + k];
        for (int k = 0; k < 8; ++k) v[k] /= (double)nwg;
        printf("[diag LS %dx%d] tiles %lld nk %d | compute wave 0: first barrier %.0f cyc | per K-step: mfma+ldsread %.0f barrier-wait %.0f\n",
               BM, BN, nwg, a.nk, v[0], v[4] / a.nk, v[7] / a.nk);
    }
#endif
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Split-bf16 variant of the loader-specialised kernel: fp32-equivalent products on the bf16 matrix pipe.
//
// Every fp32 operand is cut into three bf16 pieces by truncation, x = x1 + x2 + x3 EXACTLY (3 x 8 significant
// bits, each piece starts at the leading one of what is left), and a product is accumulated in fp32 as
//     NT = 6:  x1 y1 + (x1 y2 + x2 y1) + (x2 y2 + x1 y3 + x3 y1)      dropped terms <= 2^-23 |x y|  (fp32-class)
//     NT = 3:  x1 y1 + (x1 y2 + x2 y1)                                 dropped terms <= 2^-15 |x y|
// bf16 x bf16 products are exact in fp32, so the only roundings are the fp32 accumulations -- the same kind of
// error as the fp32 MFMA chain.  v_mfma_f32_32x32x16_bf16 runs 16x the FLOP rate of v_mfma_f32_32x32x2_f32, so six of
// them per fp32-equivalent product are 2.67x the fp32 matrix peak.
//
// The 4 loader waves fetch fp32 (same bytes as the fp32 kernel: the L2 -> L1 path is the other ceiling), split in
// registers (22 VALU per 16-byte load) and write three bf16 planes in MFMA operand order:
//     plane[pl][kg = k/8][row][8 bf16]      (16 B per row and k-group; a lane pair writes one 16-B cell)
// A compute lane (row = lane & 31, half = lane >> 5) reads k-group 2*kk + half of its row with ONE ds_read_b128 per
// plane; 32 lanes read 512 contiguous bytes.  Plane pitch BM + 4 rows keeps the loaders' 8-byte writes conflict-free.
// ------------------------------------------------------------------------------------
#define DGP_RFL(x) __builtin_amdgcn_readfirstlane(x)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool IS_B = false>
__device__ __forceinline__ void split3_bf16(const float4 v, uint2& p1, uint2& p2, uint2& p3) {
    const unsigned M = 0xFFFF0000u;
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned h1[4], h2[4], h3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h1[i] = __float_as_uint(x[i]) & M;
        const float r1 = x[i] - __uint_as_float(h1[i]);           // exact
        h2[i] = __float_as_uint(r1) & M;
        h3[i] = __float_as_uint(r1 - __uint_as_float(h2[i]));     // exact; <= 8 significant bits left
    }
    p1.x = __builtin_amdgcn_perm(h1[1], h1[0], 0x07060302); p1.y = __builtin_amdgcn_perm(h1[3], h1[2], 0x07060302);
    p2.x = __builtin_amdgcn_perm(h2[1], h2[0], 0x07060302); p2.y = __builtin_amdgcn_perm(h2[3], h2[2], 0x07060302);
    p3.x = __builtin_amdgcn_perm(h3[1], h3[0], 0x07060302); p3.y = __builtin_amdgcn_perm(h3[3], h3[2], 0x07060302);
}

// fp16 variant (NT = 2): x * s = h + l with h = fp16(x s), l = fp16(x s - h): 22 significant bits, and a product is
//     h_x h_y + (h_x l_y + l_x h_y)          dropped l_x l_y <= 2^-22 |x y|: three MFMAs instead of six.
// fp16 has 5 exponent bits, so both operands are pre-scaled by powers of two (exact) chosen from the tensors' max
// magnitudes so that max |x s| lies in [2^14, 2^15): no overflow by construction, and every element within 2^-17 of
// the tensor maximum keeps a NORMAL low part (smaller ones still carry >= 11 bits and an absolute error below
// 2^-38 of the maximum).  The maxima are device scalars: the producing conv's epilogue tracks max |out|
// (ConvArgs::out_absmax), the weight panel's is computed at load.  The epilogue undoes the scales exactly.

// Epilogue of the CS kernels when the OUTPUT tensor is H2 (the inference engine's activations): a lane owns 8 consecutive channels of
// a row -- two 16-byte reads of the staged tile, the residual's cell pair (H2 or fp32: the same 32 bytes at the same address),
// scale / bias / residual / ReLU in fp32, max |out| tracking, ONE split with the tensor's scale and two 16-byte stores (32 contiguous
// bytes per lane, 512 per row of a 128-column tile).  Chunks of two row groups are software-pipelined like ls_epilogue's.
// O1: the output (and the residual, when there is one) is an H1 tensor -- ONE 16-byte cell of 8 halves per lane at HALF the byte offset
// (the 16-bit tier; ConvArgs::out_fmt == 2).
template <int TM, int TN, int WN, bool M16 = false, bool O1 = false>
__device__ __forceinline__ void ls_epilogue_h2(const ConvArgs& p, floatx16 (&acc)[TM][TN], char* smem, int wave, int lane,
                                               int m0, int n0, int wave_m0, int wave_n0, float post, float out_scale, float res_inv_scale
#ifdef DGP_DIAG
                                               , unsigned long long (&eps)[4]      // diagnostic build: set-up | chunk 0 | chunks 1.. | absmax
#endif
                                               ) {
#ifdef DGP_DIAG
    unsigned long long s0_, s1_, s2_, s3_, s4_;
    DIAG_STAMP(s0_); s2_ = s0_;
#endif
    constexpr int OSH = O1 ? 1 : 2;                // log2 bytes per channel of the output / residual tensors
    constexpr int LDC = WN + 4;
    constexpr int C8 = WN / 8;                     // lanes per row
    constexpr int RPI = 64 / C8;                   // rows per wave-instruction
    constexpr int NV = 32 / RPI;                   // row groups per 32-row pass
    constexpr int VC = NV > 2 ? 2 : NV;
    constexpr int CPP = NV / VC, NQ = TM * CPP;
    const int half = lane >> 5, l31 = lane & 31;
    const int HoWo = p.Ho * p.Wo;
    // (16x16x32 loops: the caller staged the wave's WHOLE tile, TM x 32 rows; the 32x32x16 loops stage one 32-row pass at a time below)
    float* sC = reinterpret_cast<float*>(smem) + wave * ((M16 ? TM : 1) * 32 * LDC);
    const __amdgpu_buffer_rsrc_t rs_res =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.in), 0, p.res ? (int)p.res_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    const int my_c8 = lane % C8, my_r0 = lane / C8;
    const int co8 = n0 + wave_n0 + 8 * my_c8;
    const bool cok = co8 < p.Cout;                 // (Cout % 8 == 0 for every H2 tensor)
    // The wave's range slot is read NOW (one lane-uniform dword), not behind the last chunk: with the load, its compare and six
    // ds_bpermute round trips at the END of the wave a tile carried ~1.5 k cycles (of 7-9 k of epilogue) that nothing overlapped --
    // 16-bit tier 7 980 -> 8 490 frames/s on one stream, parity tier +1.4 % (same box; track_absmax_known)
    unsigned slot_bits = 0u;
    if (p.out_absmax)
        slot_bits = __hip_atomic_load(reinterpret_cast<const unsigned*>(p.out_absmax) + ((int)(blockIdx.x * 8u + (threadIdx.x >> 6)) & (ABSMAX_SLOTS - 1)),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // scale / bias are REQUESTED here and consumed behind the residual requests below: with `post * scale` computed on the spot the
    // compiler waited for them (an L2 round trip) before it issued the first residual load
    float4 sc_a = make_float4(1.f, 1.f, 1.f, 1.f), sc_b = sc_a, bi_a = make_float4(0.f, 0.f, 0.f, 0.f), bi_b = bi_a;
    if (cok && p.scale) { sc_a = *reinterpret_cast<const float4*>(p.scale + co8); sc_b = *reinterpret_cast<const float4*>(p.scale + co8 + 4); }
    if (cok && p.bias) { bi_a = *reinterpret_cast<const float4*>(p.bias + co8); bi_b = *reinterpret_cast<const float4*>(p.bias + co8 + 4); }
    // (out_scale / res_inv_scale: the tensors' scales, read by the caller BEFORE its K loop -- in the trainer they are predictions from range
    //  slots, a load and a wave reduction each, which stood in front of the first chunk of every tile)
    // O1 only (the 16-bit trainer's data-gradient convs): ReLU gate read from an H1 tensor of the output's shape (gate = stored half > 0)
    const __amdgpu_buffer_rsrc_t rs_mask =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((O1 && p.mask) ? p.mask : p.in), 0, (O1 && p.mask) ? (int)p.out_bytes : 0, 0x00020000);
    float amax = 0.f;
    // PRE (the 16x16x32 loops: the accumulators were staged by the caller and are dead): ALL residual (and gate) cells of the wave's
    // tile are requested up front -- NQ x VC loads in flight instead of VC, so the memory round trip is paid once per tile, not NQ times
    constexpr bool PRE = M16;
    constexpr int NS = PRE ? NQ : 2;               // register slots of the residual pipeline
    unsigned ooff[NS][VC];
    uint4 rres[NS][VC][O1 ? 1 : 2];
    uint4 rgate[O1 ? NS : 1][VC];
    auto issue = [&](int q, int slot) {
        const int i = q / CPP, v0 = (q % CPP) * VC;
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            const int row = my_r0 + (v0 + u) * RPI;
            const int m = m0 + wave_m0 + 32 * i + row;
            const bool ok = cok && m < p.M;
            ooff[slot][u] = ok ? ((unsigned)(m * p.Cout + co8) << OSH) : OOB;
            if (p.res) {
                unsigned roff = OOB;
                if (p.res_s == 1) {
                    roff = ooff[slot][u];
                } else if (O1 && p.res_s == -2 && ok) {        // residual on the 2x coarser grid (gradient of a subsample shortcut)
                    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    if (!((ho | wo) & 1)) roff = (unsigned)(((n * p.res_H + (ho >> 1)) * p.res_W + (wo >> 1)) * p.Cout + co8) << OSH;
                } else if (p.res_s > 1 && ok) {
                    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    roff = (unsigned)(((n * p.res_H + ho * p.res_s) * p.res_W + wo * p.res_s) * p.Cout + co8) << OSH;
                }
                const unsigned roff1 = roff == OOB ? OOB : roff + 16u;
                if (p.epi_nt & 1) { // streamed once: keep it from evicting the operand rows the other column tiles still need
                    rres[slot][u][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)roff, 0, 2));
                    if (!O1) rres[slot][u][O1 ? 0 : 1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)roff1, 0, 2));
                } else {
                    rres[slot][u][0] = __builtin_bit_cast(uint4, buf_load16(rs_res, roff));
                    if (!O1) rres[slot][u][O1 ? 0 : 1] = __builtin_bit_cast(uint4, buf_load16(rs_res, roff1));
                }
            }
            if constexpr (O1 && PRE) {
                if (p.mask) rgate[slot][u] = __builtin_bit_cast(uint4, buf_load16(rs_mask, ooff[slot][u]));
            }
        }
    };
    if constexpr (PRE) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) issue(q, q);
    } else {
        issue(0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);             // (keeps the requests above in front of the first use of scale / bias)
    const float sc[8] = {post * sc_a.x, post * sc_a.y, post * sc_a.z, post * sc_a.w, post * sc_b.x, post * sc_b.y, post * sc_b.z, post * sc_b.w};
    const float bi[8] = {bi_a.x, bi_a.y, bi_a.z, bi_a.w, bi_b.x, bi_b.y, bi_b.z, bi_b.w};
    DIAG_STAMP(s1_);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q == 1) DIAG_STAMP(s2_);
        const int i = q / CPP, v0 = (q % CPP) * VC, slot = PRE ? q : (q & 1);
        if (q % CPP == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!M16) sC[((r & 3) + 8 * (r >> 2) + 4 * half) * LDC + 32 * j + l31] = acc[i][j][r];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (!PRE) { if (q + 1 < NQ) issue(q + 1, slot ^ 1); }
#pragma unroll
        for (int u = 0; u < VC; ++u) {
            const int row = my_r0 + (v0 + u) * RPI;
            const float* srow = sC + ((M16 ? 32 * i : 0) + row) * LDC + 8 * my_c8;
            const float4 a0 = *reinterpret_cast<const float4*>(srow);
            const float4 a1 = *reinterpret_cast<const float4*>(srow + 4);
            float o[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.res) {
                if (O1) h1_unpack8(rres[slot][u][0], res_inv_scale, r);
                else if (p.res_fmt) h2_unpack8(rres[slot][u][0], rres[slot][u][O1 ? 0 : 1], res_inv_scale, r);
                else {
                    const float4 r0 = __builtin_bit_cast(float4, rres[slot][u][0]), r1 = __builtin_bit_cast(float4, rres[slot][u][O1 ? 0 : 1]);
                    r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r0.w; r[4] = r1.x; r[5] = r1.y; r[6] = r1.z; r[7] = r1.w;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                o[k] = o[k] * sc[k] + bi[k] + r[k];
                if (p.relu) o[k] = fmaxf(o[k], 0.f);
            }
            if constexpr (O1) {
                if (p.mask) {          // d/dx of ReLU: pass the gradient where the saved (H1) activation is positive
                    uint4 gq;
                    if constexpr (PRE) gq = rgate[slot][u]; else gq = __builtin_bit_cast(uint4, buf_load16(rs_mask, ooff[slot][u]));
                    const half8 g = __builtin_bit_cast(half8, gq);
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = (float)g[k] > 0.f ? o[k] : 0.f;
                }
                const uint4 hc = h1_pack8(o, out_scale);
                if (p.epi_nt & 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hc), rs_out, (int)ooff[slot][u], 0, 2);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hc), rs_out, (int)ooff[slot][u], 0, 0);
            } else {
            uint4 hi, lo;
            h2_pack8(o, out_scale, hi, lo);
            const unsigned ooff1 = ooff[slot][u] == OOB ? OOB : ooff[slot][u] + 16u;
            if (p.epi_nt & 2) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), rs_out, (int)ooff[slot][u], 0, 2);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), rs_out, (int)ooff1, 0, 2);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), rs_out, (int)ooff[slot][u], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), rs_out, (int)ooff1, 0, 0);
            }
            }
            if (ooff[slot][u] != OOB) {
#pragma unroll
                for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(o[k]));
            }
        }
    }
    DIAG_STAMP(s3_);
    if (p.out_absmax) {
        track_absmax_known(p.out_absmax, amax, lane, (int)(blockIdx.x * 8u + (threadIdx.x >> 6)), slot_bits);
    }
#ifdef DGP_DIAG
    DIAG_STAMP(s4_);
    eps[0] = s1_ - s0_; eps[1] = s2_ - s1_; eps[2] = s3_ - s2_; eps[3] = s4_ - s3_;
#endif
}

// BK = 32: one workgroup per CU (101 KB of LDS for 128 x 128); BK = 16: half the LDS and <= 128 registers, so TWO
// workgroups share a CU and one's prologue / epilogue / barrier waits run under the other's MFMAs.
// CW = compute waves: 4 (2 x 2, wave tile BM/2 x BN/2) or 8 (2 x 4, wave tile BM/2 x BN/4: 12-wave workgroups whose
// small wave tiles fit 85 registers, so a SIMD holds FOUR MFMA-issuing waves of two workgroups instead of two).
// MODE specialises the loader's A-side address walk (same arithmetic, fewer registers and instructions on the common paths):
//   0  everything at run time (zero-stuffed data-gradient gather `up`, row-walk stem, second source, taps)
//   1  plain convolution: taps / stride / dilation / padding, nothing else
//   2  pointwise: 1x1, stride 1, no padding (every row is its own input pixel), optional second source
//   CS (fp16 path with pre-split weights only): the loaders stage A as plain fp32 chunks and the COMPUTE waves split their
//   own operand rows in registers, in the shadow of their MFMAs; the loaders -- the critical path -- carry no arithmetic at all.
//   DMA (CS kernels, MODE 1 / 2, 128 columns): the loaders move both operands with LDS-DMA (buffer_load ... lds: no VGPR round trip,
//   no ds_write -- ds_write_b128 holds a SIMD pair's LDS data path for 13 cycles and the compute waves' reads queue behind it).
//   A image: row-major [row][8 chunks] with the chunk slot XOR-swizzled by (row >> 1) & 7 on the SOURCE side (the DMA destination is
//   lane-linear), three stages (two K-steps of lookahead); B image: the weight cells unpadded, two stages.  80 KB per workgroup.
//   AH2 (CS kernels): the A operand arrives in the H2 activation format (fp16 high / low cells written by the producer's epilogue,
//   ConvArgs::in_fmt): chunk 2 kg of a row IS the high cell of k-group kg and chunk 2 kg + 1 the low cell, so the loaders are
//   unchanged (same addresses, same bytes) and the compute waves drop the split -- ds_read + MFMA only.
//   OH2 (AH2 kernels): the output tensor is H2 too (ls_epilogue_h2); false: fp32 output (the heads' pointwise GEMM).
//   DEEP (DMA kernels, 128 columns): five A stages and four B stages (144 KB, one workgroup per CU) instead of three and two -- four / three
//   K-steps of lookahead instead of two / one.  For grids of at most one tile per CU (the training step's 11-frame batches in blocks
//   3 / 4, small inference batches): there a CU holds one workgroup and nothing else hides the L2 round trip of the weight cells.
//   Measured per launch, one stream (rocprofv3, 11 frames): 208 tiles pointwise K = 1024 44.7 -> 40.3 us, 3x3 67.6 -> 63.7 us; with more
//   than one tile per CU the 80-KB kernels' second resident workgroup is worth more (416 tiles: 88 -> 111 us), so those keep them.
//   H1 (AH2 DMA kernels, round 5; the 16-bit tier, dgp_net_set_tier(net, 1)): the A operand is an H1 tensor -- ONE 16-byte cell of 8
//   halves per 8 channels, half the bytes of H2 -- and the weights are high cells only.  Everything that moves bytes is UNCHANGED: the
//   launcher hands the kernel the tensor in 4-byte units (Cin / 2 "floats" per pixel), so a K-step's 128-byte line per pixel now holds 64
//   channels, and the weight cells [k-group of 8 channels][column] have the addresses the [k-group of 16][plane] pairs had.  What was the
//   (high, low) cell pair of k-group g is now the pair of k-groups (2 g, 2 g + 1): the compute waves issue a_even b_even + a_odd b_odd --
//   TWO MFMAs per 64 channels where the parity tier issues three per 32 -- on the same LDS reads.  A third of the matrix work and half the
//   operand bytes per FLOP; 11-bit operands: NOT inside the 1e-3 px gate (bench.py reports what it measures).  O1 = H1 output.
//   BM = 256 (round 6; the 16-bit tier only, MODE 1 / 2): FOUR compute waves of 64 rows x 128 columns each on a 256 x 128 tile, ONE
//   workgroup per CU (256 registers per wave: 128 accumulators).  At one MFMA per product the 128 x 128 tile needs 32 KB of operands
//   per 32 MFMAs and wave -- two co-resident workgroups pull 64 KB through the L2 -> LDS path per 1 024 matrix cycles, which is that
//   path's rate (EXPERIMENTS.md R5 (6): the loaders need as long to ISSUE a step's LDS-DMA pieces as the MFMAs of the step take) --
//   and every wave reads 20 fragments from LDS per 32 MFMAs.  A 64 x 128 wave tile reads 24 fragments per 64 MFMAs (-40 % LDS bytes
//   per MFMA), the workgroup moves 48 KB per 64 MFMAs and wave (-25 % L2 -> LDS bytes per MFMA) and meets at half as many barriers.
//   (The round-3 / round-5 "tall" tile -- 256 rows as EIGHT waves of 32 x 128 -- kept the per-wave tile and measured slower; removed.)
//   MEASURED (round 6, profiles/r6_w64_*.txt): correct -- the network's outputs are bit-identical on either tile -- and SLOWER on every
//   layer: 1 530-1 640 cycles per K-step where the 64 MFMAs take 1 053, with the loaders idle (barrier wait ~ 100 cycles): the tile is
//   not operand-bound any more, it is bound by its ONE compute wave per SIMD.  scripts/micro/mfma_w64.hip / mfma_w64b.hip rebuild the
//   loop piece by piece: MFMA stream 1 053 cycles per step, + weight-fragment reads 1 088, + A-cell reads and the ra -> ah / al copies
//   1 220, + one workgroup barrier per step 1 375 -- every stall two co-resident 128-row workgroups hide from each other is exposed --
//   and prologue + epilogue (12-18 k cycles per tile) have nothing to run under.  Instantiated for the parity tier's H2 cells too (three
//   MFMAs per product: 96 per K-step, so the fixed per-step cost weighs a third as much): still +8..+34 % slower per layer.  Opt-in only (DGP_W64).
//   MODE 3 (round 4; H2 DMA kernels, 3x3 / stride 1 / any dilation: the "halo walk").  With MODE 1 the A rows of a 3x3 conv cross the
//   L2 -> LDS path NINE times, once per tap (16 KB per K-step and workgroup, as much as the weight cells).  On a stride-1 conv the tap
//   (kh, kw) of output pixel m reads input pixel m + d ((kh - 1) W + (kw - 1)) of the FLATTENED [N H W] pixel list -- a constant
//   shift -- so one channel chunk's nine taps read nine shifted 128-pixel windows of ONE strip of S = 128 + 2 d (W + 1) consecutive
//   pixels.  The loaders stream those strips, chunk after chunk, ONCE through a ring of HALO_C pixels x 128 B in LDS (37 KB instead of
//   9 x 16 KB per chunk at W = 40, d = 2); the ring position of (chunk c, strip pixel j) is (c S8 + j) mod HALO_C, S8 = S rounded up
//   to 8.  What the per-tap zero padding did in MODE 1 (the loader sent out-of-image taps to an out-of-range offset) moves to the
//   compute waves: a lane knows, as a 9-bit mask per row block, which taps of its pixel fall outside the image (or belong to the
//   neighbouring row / frame the flattened shift wraps into) and reads a zero cell instead.  Ring bookkeeping, all wave-uniform: step
//   s = (c, t) reads stream pixels [lo(s), lo(s) + 128), lo = c S8 + d (kh W + kw), non-decreasing in s, so after barrier #s every
//   pixel below lo(s) is dead and the loaders may write up to lo(s) + HALO_C; they keep >= 2 K-steps (and a smooth S8 / 9 pixels per
//   step) ahead, counted vmcnt waits as in the DMA loaders.  Capacity rule: the largest jump between consecutive steps,
//   max(d (W - 2), 128 + pad) + 128 <= HALO_C - 8 (launcher).  LDS image of an 8-pixel group (one DMA instruction, 1 KiB): cell
//   index (b2 b0 | q | b1) for pixel q, chunk (b2 b1 b0): ds_read_b128's 16-lane groups pair row sets {0-3, 12-15} and {4-11} of two
//   k-groups whose chunks differ in bit 1 -- with (q, b1) as the low four bits of the cell index every such group covers the 16
//   slots of a bank row exactly once for ANY ring offset (the (row >> 1) & 7 XOR image of MODE 1 / 2 is 2-way on these reads), and
//   the low cell of a k-group is +256 B from its high cell (an instruction offset, no XOR).
//   Who computes the addresses: the compute waves are issue-bound (a first version that walked the ring and tested the masks in the
//   compute waves -- 17 VALU + 20 SALU per K-step -- cut the bank conflicts by 73 % and the loads by 37 % and was 8 % SLOWER), so the
//   loader waves do: two of them write, two steps ahead, a table of 128 cell addresses (row's ring cell for k-group 0, or the zero
//   cells for a masked tap) and a compute lane reads its two entries with one ds_read_b64 and adds its k-group's constant.
constexpr int HALO_C = 360;                       // ring capacity in pixels (multiple of 8)
constexpr int HALO_ZERO = HALO_C * 128;           // 1 KiB of zeros: what masked taps read (k-group constants up to 528, low cell at +256)
constexpr int HALO_TBL = HALO_ZERO + 1024;        // three stages of the address table: [wave][row & 15][row block] x 4 B = 512 B each (step s: stage s % 3;
                                                  // written two steps ahead, read one step ahead: the stage being written is never one being read)
constexpr int HALO_B0 = HALO_TBL + 2048;          // two stages of weight cells behind it: 49 152 + 32 768 = 80 KB, two workgroups per CU
static_assert(HALO_B0 == 49152, "80 KB per workgroup");
template <int BM, int BN, int NT, int BK, int CW, bool PB = false, int MODE = 0, bool CS = false, bool DMA = false, bool AH2 = false, bool OH2 = false, bool DEEP = false, bool H1 = false>
__global__ __launch_bounds__(64 * (CW + 4), CW == 8 ? 6 : BM == 256 ? 2 : ((BK == 16 || NT == 2) ? 4 : 2)) void conv_igemm_split_ls(const ConvArgs p) {
    static_assert(MODE != 3 || (DMA && AH2 && OH2 && BM == 128 && BN == 128 && CW == 4 && !DEEP), "halo walk: H2 / H1 tensors, LDS-DMA, 128 x 128 tiles");
    static_assert(!DEEP || (DMA && BN == 128 && BM == 128), "deep ring: LDS-DMA kernels with 128 x 128 tiles");
    static_assert(!H1 || (AH2 && DMA && !DEEP), "16-bit tier: cell input, LDS-DMA kernels");
    static_assert(BM == 128 || (BM == 256 && AH2 && OH2 && DMA && CW == 4 && BN == 128 && MODE != 3), "256-row tile: cell tensors, 64 x 128 wave tiles");
    static_assert(!AH2 || CS, "pre-split A operand: compute-side-split kernels only");
    static_assert(!OH2 || AH2, "H2 output: kernels with H2 input only (the stem writes fp32, the pool converts)");
    // compute waves: 2 x (CW / 2) over the tile; with the compute-side split 4 x 1 (each wave owns 32 rows and ALL columns, so no
    // two waves split the same A rows -- half the split arithmetic for 25 % more B fragment reads)
    constexpr int WAVES_M = (CS && CW == 4) ? 4 : 2;
    constexpr int WAVES_N = CW / WAVES_M;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NLT = 256;                       // loader threads
    constexpr int CH = BK / 4;                     // 16-byte chunks per row and K-step
    constexpr int KG = BK / 8;                     // k-groups (8 bf16 = one MFMA operand register quad) per K-step
    constexpr int RG = NLT / CH;
    constexpr int AROWS = BM / RG;
    constexpr int BSLOTS = KG * BN / 128;          // cells per loader lane pair
    constexpr int NP = NT == 6 ? 3 : 2;            // bf16 planes per operand
    constexpr int LDA = BM + (BK == 32 ? 4 : 8);   // k-group pitch in rows: the loaders' 8-byte writes of a half-wave cover all banks once
    constexpr int LDB = DMA ? BN : BN + 4;
    constexpr int LDAF = BM + 1;                   // CS: fp32 image [chunk][row][4 floats], one pad slot per chunk plane
    constexpr int A_CELLS = DMA ? BM * CH : (CS ? CH * LDAF : NP * KG * LDA), B_CELLS = NP * KG * LDB;     // 16-byte cells per buffer
    constexpr int NSA = DEEP ? 5 : (DMA ? 3 : 2);  // A stages
    constexpr int NSB = DEEP ? 4 : 2;              // B stages
    static_assert(!DMA || (CS && MODE != 0 && (BN == 128 || BN == 64) && CW == 4),
                  "LDS-DMA loaders: CS kernels, plain or pointwise walk, 4 compute waves of 32 (128-row tile) or 64 rows (256-row tile)");
    static_assert(!CS || (PB && NT == 2 && BK == 32), "compute-side split: fp16 path, pre-split weights, BK 32");
    constexpr int LDC = WN + 4;
    static_assert(NT == 2 || NT == 3 || NT == 6, "6 / 3 bf16 products, or NT = 2: fp16 high/low pair (3 products)");
    static_assert(BK == 16 || BK == 32, "K-step of 16 or 32 floats");
    static_assert(CW == 4 || CW == 8, "4 or 8 compute waves");
    static_assert(!PB || NT == 2, "pre-split weight cells exist for the fp16 path only");
    static_assert(BN == 128 || BN == 64, "B staging map assumes 64 or 128 columns");     // (launcher sizes LDS for the epilogue too)
    static_assert(BSLOTS >= 1 && AROWS >= 1, "tile too small for 256 loader lanes");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* sA = reinterpret_cast<uint4*>(smem);       // [2][NP][KG][LDA]
    uint4* sB = MODE == 3 ? reinterpret_cast<uint4*>(smem + HALO_B0) : sA + NSA * A_CELLS;                   // [2][NP][KG][LDB]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef DGP_DIAG
    unsigned long long rt0;                          // 100-MHz wall clock at the workgroup's first instruction (the per-CU timeline of the launch)
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
#endif
    int tile;
    int part = -1, tail_slot = 0;                    // part >= 0: this block computes one K-slice of a tail tile into a slab
    if (p.tail_ksplit > 1 && (int)blockIdx.x >= p.n_main) {
        tail_slot = (int)blockIdx.x - p.n_main;
        tile = p.n_main + tail_slot / p.tail_ksplit;
        part = tail_slot % p.tail_ksplit;
    } else if (p.st_gn > 0) {
        // Supertile order (wide layers whose weight panel is larger than an XCD's L2, block4's conv3): XCD x = blockIdx & 7 owns a band
        // of st_mb row tiles and walks it in supertiles of st_mc row tiles x st_gn column tiles, so that the ~64 tiles in flight on an
        // XCD share ONE chunk of A rows (st_mc x 128 rows) and 1-2 groups of weight columns that fit the L2 together -- in the plain
        // order they span all column tiles of 4 row tiles and the whole panel streams through the L2 for every 4 row tiles.
        const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
        const int per_chunk = p.st_mc * p.ntiles;
        const int c = loc / per_chunk, r = loc - c * per_chunk;
        const int left = p.st_mb - c * p.st_mc, mc = left < p.st_mc ? left : p.st_mc;       // row tiles of this (maybe last, shorter) chunk
        const int per_group = mc * p.st_gn;
        const int cg = r / per_group, r2 = r - cg * per_group;
        const int mt_ = xcd * p.st_mb + c * p.st_mc + r2 / p.st_gn;
        if (mt_ >= p.mtiles) return;                 // (the last band is padded to whole row tiles: nothing to do, before any barrier)
        tile = mt_ * p.ntiles + cg * p.st_gn + r2 % p.st_gn;
    } else {
        const int nwg = p.tail_ksplit > 1 ? p.n_main : (int)gridDim.x, b = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // Pointwise launches with TWO column tiles (block3's conv1: K 1024 -> 256): groups of 32 row tiles, column-major inside a group, so that
    // workgroups `loc` and `loc + 32` of an XCD -- the two a CU holds when the dispatcher fills the CUs once and then again -- are the two
    // column tiles of ONE row tile: the A rows, which come from the memory-side cache, are requested by both from the same CU at about the
    // same time (one L2 miss path instead of two), where the plain order pairs two row tiles of one column (sharing the L2-resident weight
    // cells).  Bit-identical; measured (profiles/r6_tile_order_ab.txt): 16-bit tier 44-46 -> 39-41 us per launch (-10 %), parity tier -2..-5 %.
    // With four or more column tiles the same pairing is a wash (+-2 %: it gives up the shared weight cells), the 3x3 layers lose 2-4 %.
    int mt, nt;
    if (MODE == 2 && p.ntiles == 2 && p.st_gn == 0 && !(p.tail_ksplit > 1)) {
        const int G = tile >> 6, lg = tile & 63;
        const int left = p.mtiles - 32 * G, hh = left < 32 ? left : 32;       // (the last group may be shorter)
        nt = lg / hh;
        mt = 32 * G + (lg - nt * hh);
    } else { mt = tile / p.ntiles; nt = tile - mt * p.ntiles; }
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    const int HoWo = p.Ho * p.Wo;
    const int nks_all = p.nk * (32 / BK);           // K-steps of BK floats (weight panels are padded to 32)
    const int nks = part >= 0 ? nks_all / p.tail_ksplit : nks_all;      // steps of THIS block
    const int ks0 = part >= 0 ? part * nks : 0;

    if (wave >= CW) {
        // ================================ loader waves ================================
        const int t = threadIdx.x - 64 * CW;
        const __amdgpu_buffer_rsrc_t rs_in =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, (int)p.w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in2 ? p.in2 : p.in), 0, p.in2 ? (int)p.in2_bytes : 0, 0x00020000);
        const int rg = t / CH;
        const int c = DMA ? ((t % CH) ^ ((rg >> 1) & 7)) : t % CH;      // DMA: the lane-linear LDS slot t % CH receives chunk slot ^ f(row)
        // MODE 2 keeps ONE byte offset per source (row m0 + rg, chunk c): row i of the lane is 32 i pixels further, a wave-uniform
        // stride, and rows past M fall off the end of the buffer (num_records = M * Cin * 4 exactly), so the hardware range check
        // zero-fills them -- no per-lane predicate, one v_add per load.  MODE 1 drops the gather base.
        int hi0[MODE == 2 ? 1 : AROWS], wi0[MODE == 2 ? 1 : AROWS], rowoff[MODE == 2 ? 1 : AROWS], nbase[MODE == 0 ? AROWS : 1];
        unsigned rowoff2[MODE == 0 ? AROWS : 1];          // second source (1x1, same pixel grid): byte offset of pixel m, chunk c
        const int cin1 = p.in2 ? p.cin_split : p.Cin, cin2 = p.Cin - cin1;
        const unsigned rowbase = (unsigned)(((m0 + rg) * cin1 + 4 * c) * 4), rowbase2 = (unsigned)(((m0 + rg) * cin2 + 4 * c) * 4);
        const int rstride = DGP_RFL(RG * cin1 * 4), rstride2 = DGP_RFL(RG * cin2 * 4);
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
            if (MODE == 2) break;
            const int m = m0 + rg + RG * i;
            if (MODE == 0) rowoff2[i] = (p.in2 && m < p.M) ? (unsigned)((m * cin2 + 4 * c) * 4) : OOB;
            if (m < p.M) {
                const int n = m / HoWo;
                const int rem = m - n * HoWo;
                const int ho = rem / p.Wo;
                const int wo = rem - ho * p.Wo;
                hi0[i] = ho * p.stride - p.pad_t;
                wi0[i] = wo * p.stride - p.pad_l;
                rowoff[i] = (((n * p.H + hi0[i]) * p.W + wi0[i]) * cin1 + 4 * c) * 4;
                if (MODE == 0) nbase[i] = n * p.H * p.W;
            } else {
                hi0[i] = -(1 << 28); wi0[i] = -(1 << 28); rowoff[i] = 0;
                if (MODE == 0) nbase[i] = 0;
            }
        }
        // B: lane pair (2j, 2j+1) owns chunks (2 kg, 2 kg + 1) of one column, so the pair fills one 16-byte LDS cell
        const unsigned b_row_bytes = (unsigned)p.CoutP * 16u;
        const int bq = t >> 1, bpar = t & 1;                    // pair index over [kg][col], chunk parity
        const int bcol = bq & (BN - 1), bkg0 = bq / BN;         // slot i covers k-group bkg0 + i * (128 / BN)
        const unsigned b_lane_off = (unsigned)(2 * bkg0 + bpar) * b_row_bytes + (unsigned)((n0 + bcol) * 16);
        // PB: the weights arrive pre-split as 16-byte cells [k-group][plane][col]; slot i of lane t is cell t + 256 i of
        // the step's [KG][2][BN] block -> no split arithmetic and one ds_write_b128 per cell
        const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(PB ? p.wh3 : (const void*)p.wpk), 0, PB ? (int)p.wh3_bytes : 0, 0x00020000);
        // cell q = t + 256 i of [KG][2][BN]: the column and the plane do not depend on i and the k-group advances by 256 / (2 BN)
        // per slot, so both the global offset and the LDS cell are (one lane value) + i * (a wave-uniform stride)
        constexpr int PB_KG_STEP = NLT / (2 * BN);
        const int pb_col = t & (BN - 1), pb_r = t / BN, pb_plane = pb_r & 1, pb_kg0 = pb_r >> 1;
        const unsigned pb_goff0 = (unsigned)(((pb_kg0 * 2 + pb_plane) * p.CoutP + n0 + pb_col) * 16);
        const int pb_gstride = DGP_RFL(PB_KG_STEP * 2 * p.CoutP * 16);
        const int pb_cell0 = (pb_plane * KG + pb_kg0) * LDB + pb_col;
        int w_kh = 0, w_kw = 0, w_ch = 0, w_tap = 0;
        if (ks0) {                                   // K-slice of a tail tile: start the walker at step ks0
            if (p.tap_minor) { w_ch = (ks0 / p.ntaps) * BK; w_tap = ks0 % p.ntaps; }
            else { const int spt = p.Cin > BK ? p.Cin / BK : 1; w_tap = ks0 / spt; w_ch = (ks0 % spt) * BK; }
            w_kh = w_tap / p.KW; w_kw = w_tap % p.KW;
        }
        float scA = 1.f, scW = 1.f;           // fp16 operand scales: read AFTER the first operand loads are in flight (below)
        if constexpr (MODE == 3) {
            typedef __attribute__((address_space(3))) void lds_void;
            const int lw = wave - CW;
            // Roles.  EVERY loader wave first issues its four one-KiB pieces of the next step's weight cells -- the cost of a step's
            // loader side is (pieces per wave) x ~115 cycles of issue + one L2 round trip, and only the weight cells have a single step
            // of lookahead, so they go first and four per wave is the minimum (measured: 5 / 5 / 6 pieces on three waves put the loaders
            // 170 cycles behind the compute waves).  What follows runs under that round trip: waves 0 and 1 write the address table (one
            // entry per lane, own two-variable walker), wave 3 keeps the ring's books and issues the ~4 strip pieces of a step, wave 2
            // nothing.  Wave 3 waits with a counted vmcnt (this iteration's strip pieces stay in flight), the others for everything.
            const int d = DGP_RFL(p.dil), Wd = DGP_RFL(p.W);
            const int S8 = DGP_RFL((128 + 2 * d * (Wd + 1) + 7) & ~7), GS = S8 >> 3;
            const int d_kh = DGP_RFL(d * (Wd - 2)), d_ch = DGP_RFL(S8 - 2 * d * (Wd + 1));
            const unsigned b_dst0 = (unsigned)DGP_RFL(pb_cell0 * 16);
            char* smB = smem + HALO_B0;
            int b_ch = 0, b_tap = 0, sb = 0;
            auto issue_cells = [&]() {                                   // weight cells of the next step (tap-minor walk)
                const unsigned kbase = (unsigned)DGP_RFL((int)((unsigned)(b_tap * p.tap_rows + (b_ch >> 2)) * b_row_bytes));
                const unsigned kgbase = (kbase >> 1) * 2u;
                char* dst = smB + sb * (B_CELLS * 16) + b_dst0;
#pragma unroll
                for (int i = 0; i < BSLOTS; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (lds_void*)(dst + i * (PB_KG_STEP * LDB * 16)), 16, (int)pb_goff0,
                                                             (int)kgbase + i * pb_gstride, 0, 0);
                if (++b_tap == 9) { b_tap = 0; b_ch += BK; }
                sb ^= 1;
            };
            if (lw < 3) {
                // table entry e = [wave w][l15][row block i] <-> tile row 32 w + 16 i + l15; this lane's entry: e = t (waves 0, 1)
                const int te = t & 127, trow = (te >> 5) * 32 + (te & 1) * 16 + ((te >> 1) & 15);
                unsigned tmask = 0;                                      // bit tp: tap tp of this row's pixel lies inside the image
                if (lw < 2) {
                    const int m = m0 + trow;
                    if (m < p.M) {
                        const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
#pragma unroll
                        for (int tp = 0; tp < 9; ++tp) {
                            const int hi = ho + (tp / 3 - 1) * d, wi = wo + (tp % 3 - 1) * d;
                            if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) tmask |= 1u << tp;
                        }
                    }
                }
                int wt = 0, wr = 0, ws = 0, tb_st = 0;                   // tap, ring position, index and table stage of the step whose table is written next
                auto write_table = [&]() {
                    unsigned u = (unsigned)(wr + trow);
                    u = u < u - (unsigned)HALO_C ? u : u - (unsigned)HALO_C;                  // min_u32(u, u - C): ring wrap
                    const unsigned a = (u << 5) + (u >> 3) * 768u;                            // (u >> 3) 1024 + (u & 7) 32
                    *reinterpret_cast<unsigned*>(smem + HALO_TBL + tb_st * 512 + te * 4) = ((tmask >> wt) & 1u) ? a : (unsigned)HALO_ZERO;
                    const int dl = wt == 8 ? d_ch : ((wt == 2 || wt == 5) ? d_kh : d);
                    wr += dl;
                    wr = wr >= HALO_C ? wr - HALO_C : wr;
                    wt = wt == 8 ? 0 : wt + 1;
                    tb_st = tb_st == 2 ? 0 : tb_st + 1;
                    ++ws;
                };
                if (lw < 2) { write_table(); if (nks > 1) write_table(); }                    // steps 0 and 1, before barrier #0
                for (int it = -1; it < nks; ++it) {
                    if (it + 1 < nks) issue_cells();                     // step it + 1
                    if (lw < 2 && it >= 0 && ws < nks) write_table();    // step it + 2: read by the compute waves behind barrier it + 1
#if defined(DGP_X) && DGP_X == 10      // timing only: this step's weight cells are not waited for (what a second step of lookahead would buy)
                    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
#else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
                    __builtin_amdgcn_s_barrier();
                }
                return;
            }
            // ---- wave 3: the ring
            const int total_g = DGP_RFL((p.Cin >> 5) * GS);
            const int q0 = m0 - d * (Wd + 1);                            // input pixel of strip position 0 (may be negative: out of range -> zeros)
            // this lane's cell of an 8-pixel group: lane = (b2 << 5) | (b0 << 4) | (q << 1) | b1 fetches chunk (b2 b1 b0) of pixel q
            const int lq = (lane >> 1) & 7, lch = ((lane >> 5) << 2) | ((lane & 1) << 1) | ((lane >> 4) & 1);
            const unsigned a_lane = (unsigned)(lq * p.Cin * 4 + lch * 16);
            int G = 0, g_c = 0, g_j = 0, Gr = 0;                         // next stream group, its chunk, its index inside the chunk, its ring group
            auto issue_group = [&]() {
                const unsigned voff = (unsigned)((q0 + 8 * g_j) * p.Cin * 4 + g_c * 128) + a_lane;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(smem + Gr * 1024), 16, (int)voff, 0, 0, 0);
                ++G;
                if (++g_j == GS) { g_j = 0; ++g_c; }
                if (++Gr == HALO_C / 8) Gr = 0;
            };
            // lo(s) of steps it, it + 1, it + 2 (negative steps count as step 0, steps past the end as the last one) as a three-deep queue
            // fed by the tap deltas -- no divisions in the loop; the smooth pacing target advances by ~S8 / 9 pixels per step
            const int pace = DGP_RFL((S8 * 7282 + 65535) >> 16);
            int l0 = 0, l1 = 0, l2 = 0, wl = 0, wt = 0, ws = 0;          // queue; walker: lo, tap and index of the newest step generated
            int smooth = 4 * pace;
#ifdef DGP_DIAG
            unsigned long long h0, h1, h2, h3, hs_is = 0, hs_wt = 0, hs_ba = 0;
#endif
            for (int it = -2; it < nks; ++it) {
                DIAG_STAMP(h0);
#if !(defined(DGP_X) && DGP_X == 10)
                if (it + 1 >= 0 && it + 1 < nks) issue_cells();          // weight cells of step it + 1 first
#endif
                if (it > -2) {                                           // ring bookkeeping
                    if (ws + 1 < nks) {
                        wl += wt == 8 ? d_ch : ((wt == 2 || wt == 5) ? d_kh : d);
                        wt = wt == 8 ? 0 : wt + 1;
                        ++ws;
                    }
                    l0 = l1; l1 = l2; l2 = wl;
                }
                const int alw_g = (l0 + HALO_C) >> 3;                    // every pixel below lo(it) is dead (see above)
                int tgt = l2 + 128;                                      // hi(it + 2)
                smooth += pace;
                if (smooth > tgt) tgt = smooth;
                int end_g = (tgt + 7) >> 3;
                end_g = end_g < alw_g ? end_g : alw_g;
                end_g = end_g < total_g ? end_g : total_g;
                const int need_g = (it + 1 >= 0 && it + 1 < nks) ? (l1 + 128 + 7) >> 3 : 0;      // must have landed at barrier it + 1
                while (G < end_g && G < need_g) issue_group();           // (normally empty: issued as lookahead earlier)
                int n_far = end_g - G;
                n_far = n_far < 0 ? 0 : n_far;
                while (G < end_g) issue_group();
#if defined(DGP_X) && DGP_X == 10
                if (it + 1 >= 0 && it + 1 < nks) { issue_cells(); n_far += 4; }
#endif
                DIAG_STAMP(h1);
                if (it >= -1) {
                    switch (n_far) {                                     // everything but this iteration's lookahead pieces has landed
                        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
#if defined(DGP_X) && DGP_X == 10
                        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
                        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
#else
                        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
#endif
                    }
                    DIAG_STAMP(h2);
                    __builtin_amdgcn_s_barrier();
                    DIAG_STAMP(h3);
#ifdef DGP_DIAG
                    if (it >= 0) { hs_is += h1 - h0; hs_wt += h2 - h1; hs_ba += h3 - h2; }
#endif
                }
            }
#ifdef DGP_DIAG
            if (p.dbg && lane == 0) {
                unsigned long long* dd = p.dbg + 10ull * blockIdx.x;
                dd[3] = hs_wt; dd[5] = hs_is; dd[6] = hs_ba;
            }
#endif
            return;
        }
        if constexpr (DMA) {
            typedef __attribute__((address_space(3))) void lds_void;
            const int lw = wave - CW;                                    // loader wave 0..3: rows 8 lw + 32 i of instruction i
            const unsigned a_dst0 = (unsigned)DGP_RFL(lw * 8 * 128);      // byte offset inside an A stage
            const unsigned b_dst0 = (unsigned)DGP_RFL(pb_cell0 * 16);     // lane 0's cell: the wave's 64 cells are consecutive
            int a_kh = w_kh, a_kw = w_kw, a_ch = w_ch, a_tap = w_tap;     // the A walker runs one K-step ahead of the B walker
            int b_ch = w_ch, b_tap = w_tap;
            char* smA = smem;
            char* smB = smem + NSA * A_CELLS * 16;
            // one loop, one issue site per operand (the walkers stay in SGPRs): iteration `it` issues B(it + 1) and A(it + 2), waits
            // until everything but A(it + 2) has landed and meets the compute waves at barrier it + 1
            static_assert((AROWS == 4 || (AROWS == 8 && !DEEP)) && (BSLOTS == 4 || BSLOTS == 2), "counted waits below: vmcnt(AROWS) = everything but the newest A stage");
            // LA / LB: K-steps of lookahead of the A / B issue sites (LA = LB + 1: a step's A rows are issued one iteration before its
            // weight cells, so everything issued after B(it + 1) is the A rows of steps it + 2 .. it + LA and the cells of it + 2 .. it + LB)
            constexpr int LA = NSA - 1, LB = NSB - 1;
            static_assert(LA == LB + 1, "the counted waits assume that a step's A rows are issued one iteration before its cells");
            static_assert(!DEEP || BSLOTS == 4, "deep ring: counted waits for 4 + 4 instructions per step");
            int sa = 0, sb = 0;
            for (int it = -LA; it < nks; ++it) {
                if (it + LB >= 0 && it + LB < nks) {
                    const unsigned kbase = (unsigned)DGP_RFL((int)((unsigned)(b_tap * p.tap_rows + (b_ch >> 2)) * b_row_bytes));
                    const unsigned kgbase = (kbase >> 1) * 2u;
                    char* dst = smB + sb * (B_CELLS * 16) + b_dst0;
#pragma unroll
                    for (int i = 0; i < BSLOTS; ++i)
#if defined(DGP_FEEDX) && (DGP_FEEDX & 2)
                        if (it < 0)
#endif
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (lds_void*)(dst + i * (PB_KG_STEP * LDB * 16)), 16, (int)pb_goff0,
                                                                 (int)kgbase + i * pb_gstride, 0, 0);
                    if (p.tap_minor) {
                        if (++b_tap == p.ntaps) { b_tap = 0; b_ch += BK; }
                    } else {
                        b_ch += BK;
                        if (b_ch >= p.Cin) { b_ch = 0; ++b_tap; }
                    }
                    b_tap = DGP_RFL(b_tap); b_ch = DGP_RFL(b_ch);
                    sb = sb == NSB - 1 ? 0 : sb + 1;
                }
                const bool moreA = it + LA < nks;
                if (moreA) {
                    const int dh = DGP_RFL(a_kh * p.dil), dw = DGP_RFL(a_kw * p.dil);
                    const int doff = DGP_RFL(((dh * p.W + dw) * p.Cin + a_ch) * 4);
                    char* dst = smA + sa * (A_CELLS * 16) + a_dst0;
                    if (MODE == 2) {
                        // (source, base and stride are chosen ONCE per step, by selects: with the choice inside the loop every LDS-DMA
                        //  instruction sat behind its own branch -- the cost the unit kernel's loader showed, DESIGN section 6a'')
                        const bool second = p.in2 && a_ch >= p.cin_split;
                        const int d2 = DGP_RFL((a_ch - p.cin_split) * 4);
                        const __amdgpu_buffer_rsrc_t rs_a = second ? rs_in2 : rs_in;
                        const unsigned base_a = second ? rowbase2 + (unsigned)d2 : rowbase + (unsigned)doff;
                        const int stride_a = second ? rstride2 : rstride;
#pragma unroll
                        for (int i = 0; i < AROWS; ++i) {
#if defined(DGP_FEEDX) && (DGP_FEEDX & 1)
                            if (it < 0)
#endif
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_void*)(dst + i * (32 * 128)), 16, (int)(base_a + (unsigned)(i * stride_a)), 0, 0, 0);
                        }
                    } else {
                        const bool tapok = DGP_RFL(a_tap) < p.ntaps;
#pragma unroll
                        for (int i = 0; i < AROWS; ++i) {
                            const int hi = hi0[i] + dh, wi = wi0[i] + dw;
                            const bool ok = tapok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
#if defined(DGP_FEEDX) && (DGP_FEEDX & 1)
                            if (it < 0)
#endif
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(dst + i * (32 * 128)), 16,
                                                                     (int)(ok ? (unsigned)(rowoff[i] + doff) : OOB), 0, 0, 0);
                        }
                    }
                    if (p.tap_minor) {
                        ++a_tap;
                        if (++a_kw == p.KW) { a_kw = 0; ++a_kh; }
                        if (a_tap == p.ntaps) { a_tap = 0; a_kw = 0; a_kh = 0; a_ch += BK; }
                    } else {
                        a_ch += BK;
                        if (a_ch >= p.Cin) {
                            a_ch = 0; ++a_tap;
                            if (++a_kw == p.KW) { a_kw = 0; ++a_kh; }
                        }
                    }
                    a_tap = DGP_RFL(a_tap); a_kw = DGP_RFL(a_kw); a_kh = DGP_RFL(a_kh); a_ch = DGP_RFL(a_ch);
                    sa = sa == NSA - 1 ? 0 : sa + 1;
                }
                if (it >= -1) {
                    if constexpr (!DEEP) {
#if defined(DGP_X) && DGP_X == 10
                        if (moreA) { if constexpr (AROWS == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
                        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
                        if (moreA) { if constexpr (AROWS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    } else {
                        // step it + 1 must have landed; what may stay in flight was issued after its cells: the A rows of steps
                        // it + 2 .. min(it + LA, nks - 1) and the cells of steps it + 2 .. min(it + LB, nks - 1), 4 instructions each
                        const int last = nks - 1;
                        int nA = (it + LA < last ? it + LA : last) - (it + 1), nB = (it + LB < last ? it + LB : last) - (it + 1);
                        nA = nA < 0 ? 0 : nA; nB = nB < 0 ? 0 : nB;
                        switch (nA + nB) {
                            case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
                            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                        }
                    }
                    __builtin_amdgcn_s_barrier();
                }
            }
            return;
        }
        float4 ra0[AROWS], rb0[BSLOTS], ra1[AROWS], rb1[BSLOTS];
        auto gload = [&](int ks, float4 (&ra)[AROWS], float4 (&rb)[BSLOTS]) {
#if defined(DGP_X) && DGP_X == 4
            if (CS) {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
#pragma unroll
                for (int i = 0; i < BSLOTS; ++i) asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w));
                return;
            }
#endif
            const int dh = DGP_RFL(w_kh * p.dil), dw = DGP_RFL(w_kw * p.dil);
            const int doff = DGP_RFL(((dh * p.W + dw) * p.Cin + w_ch) * 4);
            const bool tapok = DGP_RFL(w_tap) < p.ntaps;
            if (MODE == 2) {
                if (p.in2 && w_ch >= p.cin_split) {       // second source of a K-concatenated 1x1 conv
                    const int d2 = DGP_RFL((w_ch - p.cin_split) * 4);
#pragma unroll
                    for (int i = 0; i < AROWS; ++i) ra[i] = buf_load16(rs_in2, rowbase2 + (unsigned)(d2 + i * rstride2));
                } else {
#pragma unroll
                    for (int i = 0; i < AROWS; ++i) ra[i] = buf_load16(rs_in, rowbase + (unsigned)(doff + i * rstride));
                }
            } else if (MODE == 0 && p.up == 2) {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) {
                    const int hv = hi0[i] + dh, wv = wi0[i] + dw;
                    const bool ok = tapok && !((hv | wv) & 1) && (unsigned)(hv >> 1) < (unsigned)p.H &&
                                    (unsigned)(wv >> 1) < (unsigned)p.W;
                    const unsigned off = (unsigned)((nbase[i] + (hv >> 1) * p.W + (wv >> 1)) * p.Cin + w_ch + 4 * c) << 2;
                    ra[i] = buf_load16(rs_in, ok ? off : OOB);
                }
            } else if (MODE == 0 && p.in2 && w_ch >= p.cin_split) {        // second source of a K-concatenated 1x1 conv
                const unsigned d2 = (unsigned)(w_ch - p.cin_split) * 4u;
#pragma unroll
                for (int i = 0; i < AROWS; ++i) ra[i] = buf_load16(rs_in2, rowoff2[i] == OOB ? OOB : rowoff2[i] + d2);
            } else {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) {
                    const int hi = hi0[i] + dh, wi = wi0[i] + dw + ((MODE == 0 && p.stem) ? c : 0);
                    const bool ok = tapok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                    ra[i] = buf_load16(rs_in, ok ? (unsigned)(rowoff[i] + doff) : OOB);
                }
            }
            // weight-panel rows of this step: k = tap * Cin + channel, 4 k-values per row
            // (readfirstlane: the walker state is wave-uniform but the compiler keeps it in VGPRs and would wrap every load that
            //  takes it as the scalar offset in a waterfall loop)
            const unsigned kbase = (unsigned)DGP_RFL((int)((unsigned)(w_tap * p.tap_rows + (w_ch >> 2)) * b_row_bytes));
            if (p.tap_minor) {
                // taps fastest: the 9 taps of one channel chunk re-read nearly the same input pixels back to back,
                // so most A loads of a 3x3 conv hit the CU's L1 instead of queueing on the L2 path
                ++w_tap;
                if (++w_kw == p.KW) { w_kw = 0; ++w_kh; }
                if (w_tap == p.ntaps) { w_tap = 0; w_kw = 0; w_kh = 0; w_ch += BK; }
            } else {
                w_ch += BK;
                if (w_ch >= p.Cin) {
                    w_ch = 0; ++w_tap;
                    if (++w_kw == p.KW) { w_kw = 0; ++w_kh; }
                }
            }
            // the walker is wave-uniform: say so, or hipcc keeps it in VGPRs / scratch behind exec-masked updates
            w_tap = DGP_RFL(w_tap); w_kw = DGP_RFL(w_kw); w_kh = DGP_RFL(w_kh); w_ch = DGP_RFL(w_ch);
            if (PB) {
                const unsigned kgbase = (kbase >> 1) * 2u;          // (k-group index) * 2 planes * CoutP * 16 bytes = kbase rows / 2 * 2
#pragma unroll
                for (int i = 0; i < BSLOTS; ++i)
                    rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w3, (int)pb_goff0, (int)kgbase + i * pb_gstride, 0));
            } else {
#pragma unroll
                for (int i = 0; i < BSLOTS; ++i)
                    rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                rs_w, (int)b_lane_off, (int)(kbase + (unsigned)(2 * i * (128 / BN)) * b_row_bytes), 0));
            }
        };
        auto lstore = [&](int buf, float4 (&ra)[AROWS], float4 (&rb)[BSLOTS]) {
#if defined(DGP_X) && DGP_X == 3
            if (CS) {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) asm volatile("" :: "v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w));
#pragma unroll
                for (int i = 0; i < BSLOTS; ++i) asm volatile("" :: "v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
                return;
            }
#endif
            if (CS) {
#pragma unroll
                for (int i = 0; i < AROWS; ++i) sA[buf * A_CELLS + c * LDAF + rg + RG * i] = __builtin_bit_cast(uint4, ra[i]);
            }
            uint2* a = reinterpret_cast<uint2*>(sA + buf * A_CELLS) + (((c >> 1) * LDA + rg) * 2 + (c & 1));
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
                if (CS) break;
                uint2 p1, p2, p3;
                if (NT == 2) split2_f16(ra[i], scA, p1, p2);
                else split3_bf16(ra[i], p1, p2, p3);
                a[(RG * i) * 2] = p1;
                a[(KG * LDA + RG * i) * 2] = p2;
                if (NP == 3) a[(2 * KG * LDA + RG * i) * 2] = p3;
            }
            if (PB) {
#pragma unroll
                for (int i = 0; i < BSLOTS; ++i) sB[buf * B_CELLS + pb_cell0 + i * (PB_KG_STEP * LDB)] = __builtin_bit_cast(uint4, rb[i]);
                return;
            }
            uint2* b = reinterpret_cast<uint2*>(sB + buf * B_CELLS) + ((bkg0 * LDB + bcol) * 2 + bpar);
#pragma unroll
            for (int i = 0; i < BSLOTS; ++i) {
                uint2 p1, p2, p3;
                if (NT == 2) split2_f16(rb[i], scW, p1, p2);
                else split3_bf16<true>(rb[i], p1, p2, p3);
                const int o = i * (128 / BN) * LDB * 2;
                b[o] = p1;
                b[KG * LDB * 2 + o] = p2;
                if (NP == 3) b[2 * KG * LDB * 2 + o] = p3;
            }
        };
        gload(0, ra0, rb0);
        if (nks > 1) gload(1, ra1, rb1);
        if (NT == 2) {                         // the range reads (1 KB each + a wave reduction) overlap the first operand loads
            scA = pow2_scale_for(p.in_absmax, lane, p.in2_absmax);
            scW = PB ? 1.f : pow2_scale_for(p.w_absmax, lane);
        }
        lstore(0, ra0, rb0);
        if (nks > 2) gload(2, ra0, rb0);
        __syncthreads();
#ifdef DGP_DIAG
        unsigned long long l0, l1, l2, l3, t_st = 0, t_ld = 0, t_ba = 0;
#endif
        for (int ks = 0; ks < nks; ks += 2) {
            DIAG_STAMP(l0);
            if (ks + 1 < nks) lstore(1, ra1, rb1);
            DIAG_STAMP(l1);
            if (ks + 3 < nks) gload(ks + 3, ra1, rb1);
            DIAG_STAMP(l2);
            __syncthreads();
            DIAG_STAMP(l3);
#ifdef DGP_DIAG
            t_st += l1 - l0; t_ld += l2 - l1; t_ba += l3 - l2;
#endif
            if (ks + 1 >= nks) break;
            DIAG_STAMP(l0);
            if (ks + 2 < nks) lstore(0, ra0, rb0);
            DIAG_STAMP(l1);
            if (ks + 4 < nks) gload(ks + 4, ra0, rb0);
            DIAG_STAMP(l2);
            __syncthreads();
            DIAG_STAMP(l3);
#ifdef DGP_DIAG
            t_st += l1 - l0; t_ld += l2 - l1; t_ba += l3 - l2;
#endif
        }
#ifdef DGP_DIAG
        if (p.dbg && t == 0) {
            unsigned long long* d = p.dbg + 10ull * blockIdx.x;
            d[3] = t_st; d[5] = t_ld; d[6] = t_ba;
        }
#endif
        return;
    }

    // ================================== compute waves ==================================
    const int wave_m0 = (wave / WAVES_N) * WM;
    const int wave_n0 = (wave % WAVES_N) * WN;
    const int half = lane >> 5, l31 = lane & 31;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float scA_c = (CS && !AH2) ? pow2_scale_for(p.in_absmax, lane, p.in2_absmax) : 1.f;       // operand scale of the compute-side split
    // exact power of two that undoes the fp16 operand scales; read now, while this wave waits for the first tile anyway
    const float post = (NT == 2 && part < 0) ? 1.f / ((AH2 ? h2_in_scale(p, lane) : pow2_scale_for(p.in_absmax, lane, p.in2_absmax)) * pow2_scale_for(p.w_absmax, lane)) : 1.f;
    // scales of the H2 / H1 output and residual tensors (wave-uniform; host floats in the inference engine, predictions from range slots in the trainer)
    const float epi_out_scale = OH2 ? h2_out_scale(p, lane) : 1.f, epi_res_inv_scale = (OH2 && p.res_fmt) ? h2_res_inv_scale(p, lane) : 1.f;
#ifdef DGP_DIAG
    unsigned long long e0, e1, e2, e3, acc_mf = 0, acc_ba = 0;
    DIAG_STAMP(e0);
#endif
    if constexpr (MODE == 3) {        // the zero cells masked taps read (before barrier #0)
        if (threadIdx.x < 64) *reinterpret_cast<uint4*>(smem + HALO_ZERO + 16 * threadIdx.x) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
#ifdef DGP_DIAG
    DIAG_STAMP(e1);
    const unsigned long long t_pro = e1 - e0;
#endif
    // 16x16x32 MFMAs in the pipelined loop of the 32 x 128 wave tile: the same FLOPs, LDS bytes and register reads as the 32x32x16
    // shape in twice as many, half as long matrix instructions -- +5.3 % end to end (block4 3x3: 0.499 -> 0.453 ms)
    constexpr bool M16 = CS && (TM == 1 || (TM == 2 && AH2 && DMA)) && (TN == 4 || (TN == 2 && DMA)) && NT == 2 && BK == 32;      // (32 x 64 wave tiles: DMA images only)
    static_assert(!DMA || ((TM == 1 || (TM == 2 && AH2)) && (TN == 4 || TN == 2)), "DMA image is read by the pipelined loops only");
    static_assert(!(DMA && AH2) || M16, "pre-split A + DMA image: 16x16x32 loop only");
    if constexpr (MODE == 3) {
        // The 16x16x32 loop of the branch below on the pixel ring: same fragment ring, same barrier placement; the A cells of the NEXT
        // step come from the loaders' address table (one ds_read_b64 behind the previous barrier + this lane's k-group constant).
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        const int l15 = lane & 15, g = lane >> 4;
        constexpr int NJ = 8, NF = 16;
        floatx4 c[2][NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) c[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const unsigned lane_c = (unsigned)((g >> 1) * 512 + (g & 1) * 16);       // k-group g: chunk 2 g = (b2 b1 0) -> cell (b2 0 | q | b1)
        const char* tbl = smem + HALO_TBL + (wave * 32 + l15 * 2) * 4;           // this lane's two entries of a table stage
        int tst = 0;                                                             // stage of the table to read next
        u32x2 tq;
        unsigned a_adr[2];
        const uint4* B = sB + wave_n0 + l15 + g * LDB;
        int db = B_CELLS;
        uint4 ra[2][2], ah[2], al[2], bq[4];
        auto mma = [](const uint4& x, const uint4& y, floatx4 cc) {
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), cc, 0, 0, 0);
        };
        // (barrier #0 above: step 0's pixels and cells have landed, the zero cells and the tables of steps 0 and 1 are written)
        tq = *reinterpret_cast<const u32x2*>(tbl);
        a_adr[0] = tq[0] + lane_c; a_adr[1] = tq[1] + lane_c;
        tq = *reinterpret_cast<const u32x2*>(tbl + 512);
        tst = 2;                                            // (next: the table of step 2)
#define DGP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define DGP_RA(I) do { ra[I][0] = *reinterpret_cast<const uint4*>(smem + a_adr[I]);                                 \
                       ra[I][1] = *reinterpret_cast<const uint4*>(smem + a_adr[I] + 256); } while (0)
#define DGP_RB(F) do { bq[(F) & 3] = B[((((F) & 1) ? 0 : 1) * KG) * LDB + 16 * (((F) % NF) >> 1)]; } while (0)
        // (H1: fragment F even = plane 1 = the odd k-groups' weights x the odd chunks `al`; F odd = plane 0 x the even chunks `ah`)
#define DGP_MM(F) do { constexpr int j_ = (F) >> 1;                                                                \
        if constexpr (H1) { if (((F) & 1) == 0) { c[0][j_] = mma(al[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(al[1], bq[(F) & 3], c[1][j_]); } \
                            else { c[0][j_] = mma(ah[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(ah[1], bq[(F) & 3], c[1][j_]); } }              \
        else if (((F) & 1) == 0) { c[0][j_] = mma(ah[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(ah[1], bq[(F) & 3], c[1][j_]); }    \
        else { c[0][j_] = mma(al[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(al[1], bq[(F) & 3], c[1][j_]);         \
               c[0][j_] = mma(ah[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(ah[1], bq[(F) & 3], c[1][j_]); } } while (0)
#define DGP_STEP(F) do { DGP_MM(F); DGP_FENCE(); DGP_RB((F) + 4); DGP_FENCE(); } while (0)
        DGP_RA(0); DGP_RA(1);
        DGP_RB(0); DGP_RB(1); DGP_RB(2); DGP_RB(3);
        DGP_FENCE();
        for (int ks = 0; ks < nks; ++ks) {
            ah[0] = ra[0][0]; al[0] = ra[0][1]; ah[1] = ra[1][0]; al[1] = ra[1][1];
            DGP_FENCE();
            a_adr[0] = tq[0] + lane_c; a_adr[1] = tq[1] + lane_c;      // the next step's A cells (table read behind the previous barrier)
            DGP_FENCE();
            DGP_STEP(0); DGP_STEP(1); DGP_STEP(2); DGP_STEP(3);
            DGP_STEP(4); DGP_STEP(5);
            DGP_STEP(6); DGP_STEP(7); DGP_STEP(8); DGP_STEP(9); DGP_STEP(10); DGP_STEP(11);
            DGP_MM(12); DGP_FENCE();
            DGP_MM(13); DGP_FENCE();
            DIAG_STAMP(e2);
            __syncthreads();
            DIAG_STAMP(e3);
#ifdef DGP_DIAG
            acc_mf += e2 - e1; acc_ba += e3 - e2; e1 = e3;
#endif
            B += db; db = -db;
            const bool more = ks + 1 < nks;
            if (more) {
                DGP_RA(0); DGP_RA(1); DGP_RB(0); DGP_RB(1);
                tq = *reinterpret_cast<const u32x2*>(tbl + tst * 512);      // table of step ks + 2 (written before this barrier)
                tst = tst == 2 ? 0 : tst + 1;
            }
            DGP_FENCE();
            DGP_MM(14); DGP_FENCE();
            if (more) DGP_RB(2);
            DGP_FENCE();
            DGP_MM(15); DGP_FENCE();
            if (more) DGP_RB(3);
            DGP_FENCE();
        }
#undef DGP_FENCE
#undef DGP_RA
#undef DGP_RB
#undef DGP_MM
#undef DGP_STEP
        {
            constexpr int LDCW = WN + 4;
            float* sCw = reinterpret_cast<float*>(smem) + wave * (32 * LDCW);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sCw[(16 * i + 4 * g + r) * LDCW + 16 * j + l15] = c[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if constexpr (M16 && TM == 2) {
        // 64 x 128 wave tile of the 256-row tile (16-bit tier): four row blocks x eight column blocks of 16x16x32 MFMAs, 64 per K-step
        // of 64 channels; a weight fragment feeds FOUR MFMAs, so a wave reads 16 + 8 cells per 64 MFMAs where the 32 x 128 tile reads
        // 16 + 4 per 32.  ONE compute wave per SIMD: nothing hides an exposed LDS round trip (the first version -- the 32 x 128 loop's
        // four-fragment ring, the barrier before the last two fragments -- ran 1 530 cycles per K-step for 1 024 cycles of MFMAs), so
        // the fragment ring is EIGHT quads deep (fragment F + 8 is read behind the MFMAs of fragment F: seven fragments = 448 matrix
        // cycles of cover) and the barrier sits before the last FOUR fragments, whose 16 MFMAs cover the next step's A cells.
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        const int l15 = lane & 15, g = lane >> 4;
        constexpr int NJ = 8, NF = 16;
        floatx4 c[4][NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) c[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        // DMA image: row-major, chunk ch of row r in slot ch ^ ((r >> 1) & 7) (the same for r + 16 i); this lane wants chunks 2 g, 2 g + 1
        const unsigned a_row0 = (unsigned)((wave_m0 + l15) * 128 + (((2 * g) ^ (((wave_m0 + l15) >> 1) & 7)) << 4));
        unsigned a_cur = a_row0;
        int sa_c = 0;
        const uint4* B = sB + wave_n0 + l15 + g * LDB;
        int db = B_CELLS;
        uint4 ra[4][2], ah[4], al[4], bq[8];
        auto mma = [](const uint4& x, const uint4& y, floatx4 cc) {
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), cc, 0, 0, 0);
        };
#define DGP_FENCE() __builtin_amdgcn_sched_barrier(0)
#if defined(DGP_X) && (DGP_X == 8 || DGP_X == 9)      // timing-only: no A cell reads
#define DGP_RA(I) do { asm volatile("" : "+v"(ra[I][0].x), "+v"(ra[I][0].y), "+v"(ra[I][0].z), "+v"(ra[I][0].w), "+v"(ra[I][1].x), "+v"(ra[I][1].y), "+v"(ra[I][1].z), "+v"(ra[I][1].w)); } while (0)
#else
#define DGP_RA(I) do { ra[I][0] = *reinterpret_cast<const uint4*>(smem + a_cur + (I) * 2048);                        \
                       ra[I][1] = *reinterpret_cast<const uint4*>(smem + (a_cur ^ 16u) + (I) * 2048); } while (0)
#endif
#if defined(DGP_X) && (DGP_X == 6 || DGP_X == 8)      // timing-only: no B fragment reads
#define DGP_RB(F) do { asm volatile("" : "+v"(bq[(F) & 7].x), "+v"(bq[(F) & 7].y), "+v"(bq[(F) & 7].z), "+v"(bq[(F) & 7].w)); } while (0)
#else
#define DGP_RB(F) do { bq[(F) & 7] = B[((((F) & 1) ? 0 : 1) * KG) * LDB + 16 * (((F) % NF) >> 1)]; } while (0)
#endif
        // H1: fragment F even = plane 1 = the odd k-groups' weights x the odd chunks `al`; F odd = plane 0 x the even chunks `ah`.
        // H2 (parity tier): chunk 2 g = the HIGH cell `ah` of k-group g, chunk 2 g + 1 its LOW cell `al`; F even = the low weight plane x ah,
        // F odd = the high plane x al, then x ah -- per accumulator the same order as the 32 x 128 loop (a_hi b_lo, a_lo b_hi, a_hi b_hi).
#define DGP_MM4(X, F, J) do { c[0][J] = mma(X[0], bq[(F) & 7], c[0][J]); c[1][J] = mma(X[1], bq[(F) & 7], c[1][J]);                     \
                              c[2][J] = mma(X[2], bq[(F) & 7], c[2][J]); c[3][J] = mma(X[3], bq[(F) & 7], c[3][J]); } while (0)
#define DGP_MM(F) do { constexpr int j_ = (F) >> 1;                                                                 \
        if constexpr (H1) { if (((F) & 1) == 0) DGP_MM4(al, F, j_); else DGP_MM4(ah, F, j_); }                      \
        else if (((F) & 1) == 0) DGP_MM4(ah, F, j_);                                                                \
        else { DGP_MM4(al, F, j_); DGP_MM4(ah, F, j_); } } while (0)
#define DGP_STEP(F) do { DGP_MM(F); DGP_FENCE(); DGP_RB((F) + 8); DGP_FENCE(); } while (0)
#define DGP_TAIL(F) do { DGP_MM(F); DGP_FENCE(); if (more) DGP_RB((F) - 8); DGP_FENCE(); } while (0)
        DGP_RA(0); DGP_RA(1); DGP_RA(2); DGP_RA(3);
        DGP_RB(0); DGP_RB(1); DGP_RB(2); DGP_RB(3); DGP_RB(4); DGP_RB(5); DGP_RB(6); DGP_RB(7);
        DGP_FENCE();
        for (int ks = 0; ks < nks; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { ah[i] = ra[i][0]; al[i] = ra[i][1]; }
            DGP_FENCE();
            DGP_STEP(0); DGP_STEP(1); DGP_STEP(2); DGP_STEP(3); DGP_STEP(4); DGP_STEP(5); DGP_STEP(6); DGP_STEP(7);
            DGP_MM(8); DGP_FENCE();
            DGP_MM(9); DGP_FENCE();
            DGP_MM(10); DGP_FENCE();
            DGP_MM(11); DGP_FENCE();
            DIAG_STAMP(e2);
            __syncthreads();
            DIAG_STAMP(e3);
#ifdef DGP_DIAG
            acc_mf += e2 - e1; acc_ba += e3 - e2; e1 = e3;
#endif
            sa_c = sa_c == NSA - 1 ? 0 : sa_c + 1; a_cur = a_row0 + (unsigned)(sa_c * (A_CELLS * 16));
            static_assert(NSB == 2, "256-row tile: two weight stages");
            B += db; db = -db;
            const bool more = ks + 1 < nks;
            if (more) { DGP_RA(0); DGP_RA(1); DGP_RA(2); DGP_RA(3); DGP_RB(0); DGP_RB(1); DGP_RB(2); DGP_RB(3); }
            DGP_FENCE();
            DGP_TAIL(12); DGP_TAIL(13); DGP_TAIL(14); DGP_TAIL(15);
        }
#undef DGP_FENCE
#undef DGP_RA
#undef DGP_RB
#undef DGP_MM
#undef DGP_MM4
#undef DGP_STEP
#undef DGP_TAIL
        {      // stage the wave's 64 x 128 tile (its own LDS slice: the ring is dead behind the last barrier)
            constexpr int LDCW = WN + 4;
            float* sCw = reinterpret_cast<float*>(smem) + wave * (64 * LDCW);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sCw[(16 * i + 4 * g + r) * LDCW + 16 * j + l15] = c[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if constexpr (M16) {
        // One K-step = one MFMA depth (32).  Per step a wave splits its 2 x 16 rows (A: two fp32 chunks per lane and row block ->
        // a_hi / a_lo), and walks 16 B fragments f = (column block j = f / 2, plane: low first) through a ring of four register
        // quads, three fragments ahead of the MFMAs: low plane -> a_hi b_lo for both row blocks, high plane -> a_lo b_hi, a_hi b_hi.
        // The barrier sits before the last two fragments' MFMAs and the next step's first reads fly under them.  (An 8-quad ring,
        // 5-7 ahead, measured the same; the 32 x 64 wave tile of the 128 x 64 kernels on this loop measured 3 % slower: its
        // register-staged images are laid out for the 32-lane fragment reads.)
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        const int l15 = lane & 15, g = lane >> 4;
        constexpr int NJ = 2 * TN, NF = 2 * NJ;          // 16-wide column blocks of the wave tile (8 or 4), B fragments per K-step
        floatx4 c[2][NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) c[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const uint4* a_lane = sA + (2 * g) * LDAF + wave_m0 + l15;                 // register-staged image [chunk][row]
        // DMA image: row-major, chunk ch of row r in slot ch ^ ((r >> 1) & 7) (same for r and r + 16); this lane wants chunks 2 g, 2 g + 1
        const unsigned a_row0 = (unsigned)((wave_m0 + l15) * 128 + (((2 * g) ^ (((wave_m0 + l15) >> 1) & 7)) << 4));
        unsigned a_cur = a_row0;
        int sa_c = 0;
        const uint4* A = a_lane;
        const uint4* B = sB + wave_n0 + l15 + g * LDB;
        int da = A_CELLS, db = B_CELLS;            // to the other buffer and back (deep ring: to the next stage, wrapping)
        int sb_c = 0; (void)sb_c;
        uint4 ra[2][2], ah[2], al[2], bq[4];
#if defined(DGP_X) && DGP_X == 7      // timing-only: no MFMAs in the 16x16x32 loop
        auto mma = [](const uint4& x, const uint4& y, floatx4 cc) { asm volatile("" :: "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "v"(y.x), "v"(y.y), "v"(y.z), "v"(y.w)); return cc; };
#else
        auto mma = [](const uint4& x, const uint4& y, floatx4 cc) {
            return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), cc, 0, 0, 0);
        };
#endif
#define DGP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define DGP_RA(I) do { if constexpr (DMA) {                                                                         \
            ra[I][0] = *reinterpret_cast<const uint4*>(smem + a_cur + (I) * 2048);                                 \
            ra[I][1] = *reinterpret_cast<const uint4*>(smem + (a_cur ^ 16u) + (I) * 2048);                         \
        } else { ra[I][0] = A[16 * (I)]; ra[I][1] = A[LDAF + 16 * (I)]; } } while (0)
#if defined(DGP_X) && DGP_X == 6      // timing-only: no B fragment reads in the 16x16x32 loop
#define DGP_RB(F) do { asm volatile("" : "+v"(bq[(F) & 3].x), "+v"(bq[(F) & 3].y), "+v"(bq[(F) & 3].z), "+v"(bq[(F) & 3].w)); } while (0)
#else
#define DGP_RB(F) do { bq[(F) & 3] = B[((((F) & 1) ? 0 : 1) * KG) * LDB + 16 * (((F) % NF) >> 1)]; } while (0)
#endif
#if defined(DGP_X) && DGP_X == 1      // timing-only stand-in: the A operand as if it arrived pre-split (no split arithmetic)
#define DGP_SPLIT(I) do { ah[I] = ra[I][0]; al[I] = ra[I][1]; } while (0)
#else
#define DGP_SPLIT(I) do { if constexpr (AH2) { ah[I] = ra[I][0]; al[I] = ra[I][1]; break; }                       \
        uint2 h0_, l0_, h1_, l1_;                                                                                  \
        split2_f16(__builtin_bit_cast(float4, ra[I][0]), scA_c, h0_, l0_);                                         \
        split2_f16(__builtin_bit_cast(float4, ra[I][1]), scA_c, h1_, l1_);                                         \
        ah[I] = make_uint4(h0_.x, h0_.y, h1_.x, h1_.y); al[I] = make_uint4(l0_.x, l0_.y, l1_.x, l1_.y); } while (0)
#endif
#define DGP_MM(F) do { constexpr int j_ = (F) >> 1;                                                                \
        if constexpr (H1) { if (((F) & 1) == 0) { c[0][j_] = mma(al[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(al[1], bq[(F) & 3], c[1][j_]); } \
                            else { c[0][j_] = mma(ah[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(ah[1], bq[(F) & 3], c[1][j_]); } }              \
        else if (((F) & 1) == 0) { c[0][j_] = mma(ah[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(ah[1], bq[(F) & 3], c[1][j_]); }    \
        else { c[0][j_] = mma(al[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(al[1], bq[(F) & 3], c[1][j_]);         \
               c[0][j_] = mma(ah[0], bq[(F) & 3], c[0][j_]); c[1][j_] = mma(ah[1], bq[(F) & 3], c[1][j_]); } } while (0)
#define DGP_STEP(F) do { DGP_MM(F); DGP_FENCE(); DGP_RB((F) + 4); DGP_FENCE(); } while (0)
#ifdef DGP_M16_SIMPLE      // debugging aid: the same arithmetic without any software pipelining
        for (int ks = 0; ks < nks; ++ks) {
            DGP_RA(0); DGP_RA(1);
            DGP_SPLIT(0); DGP_SPLIT(1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const uint4 bl = B[(1 * KG) * LDB + 16 * j], bh = B[16 * j];
#if DGP_M16_SIMPLE != 2
                c[0][j] = mma(ah[0], bl, c[0][j]); c[1][j] = mma(ah[1], bl, c[1][j]);
#endif
#if DGP_M16_SIMPLE != 3
                c[0][j] = mma(al[0], bh, c[0][j]); c[1][j] = mma(al[1], bh, c[1][j]);
#endif
                c[0][j] = mma(ah[0], bh, c[0][j]); c[1][j] = mma(ah[1], bh, c[1][j]);
            }
            __syncthreads();
            if constexpr (DMA) { sa_c = sa_c == NSA - 1 ? 0 : sa_c + 1; a_cur = a_row0 + (unsigned)(sa_c * (A_CELLS * 16)); }
            else { A += da; da = -da; }
            if constexpr (NSB == 2) { B += db; db = -db; } else { B += db; if (++sb_c == NSB) { sb_c = 0; B -= NSB * B_CELLS; } }
        }
#else
        DGP_RA(0); DGP_RA(1);
        DGP_RB(0); DGP_RB(1); DGP_RB(2); DGP_RB(3);
        DGP_FENCE();
        for (int ks = 0; ks < nks; ++ks) {
            DGP_SPLIT(0);
            DGP_SPLIT(1);
            DGP_FENCE();
            if constexpr (TN == 2) {       // 32 x 64 wave tile: 8 fragments per step, same ring and barrier placement
                DGP_STEP(0); DGP_STEP(1); DGP_STEP(2); DGP_STEP(3);
                DGP_MM(4); DGP_FENCE();
                DGP_MM(5); DGP_FENCE();
                __syncthreads();
                if constexpr (DMA) { sa_c = sa_c == NSA - 1 ? 0 : sa_c + 1; a_cur = a_row0 + (unsigned)(sa_c * (A_CELLS * 16)); }
                else { A += da; da = -da; }
                if constexpr (NSB == 2) { B += db; db = -db; } else { B += db; if (++sb_c == NSB) { sb_c = 0; B -= NSB * B_CELLS; } }
                const bool more2 = ks + 1 < nks;
                if (more2) { DGP_RA(0); DGP_RA(1); DGP_RB(0); DGP_RB(1); }
                DGP_FENCE();
                DGP_MM(6); DGP_FENCE();
                if (more2) DGP_RB(2);
                DGP_FENCE();
                DGP_MM(7); DGP_FENCE();
                if (more2) DGP_RB(3);
                DGP_FENCE();
                continue;
            }
            DGP_STEP(0); DGP_STEP(1); DGP_STEP(2); DGP_STEP(3); DGP_STEP(4); DGP_STEP(5);
            DGP_STEP(6); DGP_STEP(7); DGP_STEP(8); DGP_STEP(9); DGP_STEP(10); DGP_STEP(11);
            DGP_MM(12); DGP_FENCE();
            DGP_MM(13); DGP_FENCE();
            DIAG_STAMP(e2);
            __syncthreads();
            DIAG_STAMP(e3);
#ifdef DGP_DIAG
            acc_mf += e2 - e1; acc_ba += e3 - e2; e1 = e3;
#endif
            if constexpr (DMA) { sa_c = sa_c == NSA - 1 ? 0 : sa_c + 1; a_cur = a_row0 + (unsigned)(sa_c * (A_CELLS * 16)); }
            else { A += da; da = -da; }
            if constexpr (NSB == 2) { B += db; db = -db; } else { B += db; if (++sb_c == NSB) { sb_c = 0; B -= NSB * B_CELLS; } }
            const bool more = ks + 1 < nks;
            if (more) { DGP_RA(0); DGP_RA(1); DGP_RB(0); DGP_RB(1); }
            DGP_FENCE();
            DGP_MM(14); DGP_FENCE();
            if (more) DGP_RB(2);
            DGP_FENCE();
            DGP_MM(15); DGP_FENCE();
            if (more) DGP_RB(3);
            DGP_FENCE();
        }
#endif
#undef DGP_FENCE
#undef DGP_RA
#undef DGP_RB
#undef DGP_SPLIT
#undef DGP_MM
#undef DGP_STEP
        {      // stage the 16x16 blocks into this wave's LDS tile (what ls_epilogue / ls_store_raw do for the 32x32 blocks)
            constexpr int LDCW = WN + 4;
            float* sCw = reinterpret_cast<float*>(smem) + wave * (32 * LDCW);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sCw[(16 * i + 4 * g + r) * LDCW + 16 * j + l15] = c[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else
    // (nothing but the next `if constexpr` may stand between this `else` and its statement: two static_asserts used to, which made
    //  THEM the else branch and let the generic K loop below run -- as nks extra barriers, its MFMAs being dead code -- after the
    //  16x16x32 loop as well.  Harmless while a block ran one tile; found when a persistent variant desynchronised its barriers)
    if constexpr (CS && TM == 1 && TN == 4 && NT == 2 && BK == 32 && !AH2) {
        // Software-pipelined K loop of the 32 x 128 wave tile.  hipcc's schedule read each B fragment right before its MFMAs
        // (ds_read, s_waitcnt 0, mfma: ~6 exposed LDS round trips per 16-wide slice); here the 16 B fragments of a K-step flow
        // through a ring of three fragment PAIRS (24 registers) two pairs ahead of the MFMAs that consume them, the A rows of the
        // next slice are read and split under the current slice's MFMAs, and the workgroup barrier sits BEFORE the last MFMA group
        // of a step so that the first reads of the next step fly under it.  Ring slot roles (1 <-> 2) alternate from step to step,
        // hence the body is instantiated for both parities.  Order per accumulator: a_hi b_lo, a_lo b_hi, a_hi b_hi.
        const uint4* a_lane = sA + wave_m0 + l31 + 2 * half * LDAF;
        const uint4* b_lane = sB + wave_n0 + l31 + half * LDB;
        // DMA image: row-major, chunk c of row r in slot c ^ ((r >> 1) & 7); this lane wants chunks 2 half + {0, 1, 4, 5}
        const unsigned a_row0 = (unsigned)((wave_m0 + l31) * 128 + (((2 * half) ^ (((wave_m0 + l31) >> 1) & 7)) << 4));
        unsigned a_cur = a_row0;
        int sa_c = 0;
        uint4 ra[2], ah[1], al[1], bq[3][2];
        auto mma = [](const uint4& x, const uint4& y, floatx16 c) {
            return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), c, 0, 0, 0);
        };
#define DGP_RA(A, KK) do { if constexpr (DMA) {                                                                       \
            ra[0] = *reinterpret_cast<const uint4*>(smem + (a_cur ^ (unsigned)((4 * (KK)) << 4)));                 \
            ra[1] = *reinterpret_cast<const uint4*>(smem + (a_cur ^ (unsigned)((4 * (KK) + 1) << 4)));             \
        } else { ra[0] = (A)[(4 * (KK)) * LDAF]; ra[1] = (A)[(4 * (KK) + 1) * LDAF]; } } while (0)
#define DGP_RB(B, S, KK, PL, J0) do { bq[S][0] = (B)[((PL) * KG + 2 * (KK)) * LDB + 32 * (J0)];        \
                                      bq[S][1] = (B)[((PL) * KG + 2 * (KK)) * LDB + 32 * ((J0) + 1)]; } while (0)
#if defined(DGP_X) && DGP_X == 1
#define DGP_SPLIT_HALF(KK, C) do { uint2 h_, l_; h_.x = ra[C].x; h_.y = ra[C].y; l_.x = ra[C].z; l_.y = ra[C].w;   \
        if ((C) == 0) { ah[0].x = h_.x; ah[0].y = h_.y; al[0].x = l_.x; al[0].y = l_.y; }                         \
        else { ah[0].z = h_.x; ah[0].w = h_.y; al[0].z = l_.x; al[0].w = l_.y; } } while (0)
#else
#define DGP_SPLIT_HALF(KK, C) do { uint2 h_, l_; split2_f16(__builtin_bit_cast(float4, ra[C]), scA_c, h_, l_);    \
        if ((C) == 0) { ah[0].x = h_.x; ah[0].y = h_.y; al[0].x = l_.x; al[0].y = l_.y; }                         \
        else { ah[0].z = h_.x; ah[0].w = h_.y; al[0].z = l_.x; al[0].w = l_.y; } } while (0)
#endif
#define DGP_GL(KK, J0, S) do { acc[0][J0] = mma(ah[0], bq[S][0], acc[0][J0]); acc[0][(J0) + 1] = mma(ah[0], bq[S][1], acc[0][(J0) + 1]); } while (0)
#define DGP_GH(KK, J0, S) do { acc[0][J0] = mma(al[0], bq[S][0], acc[0][J0]); acc[0][(J0) + 1] = mma(al[0], bq[S][1], acc[0][(J0) + 1]); \
                               acc[0][J0] = mma(ah[0], bq[S][0], acc[0][J0]); acc[0][(J0) + 1] = mma(ah[0], bq[S][1], acc[0][(J0) + 1]); } while (0)
#define DGP_FENCE() __builtin_amdgcn_sched_barrier(0)
        DGP_RA(a_lane, 0);
        DGP_RB(b_lane, 0, 0, 1, 0);
        DGP_RB(b_lane, 1, 0, 0, 0);
#if defined(DGP_X) && DGP_X == 2
        DGP_RB(b_lane, 2, 0, 0, 2);
#undef DGP_RB
#define DGP_RB(B, S, KK, PL, J0) do { asm volatile("" : "+v"(bq[S][0].x), "+v"(bq[S][0].y), "+v"(bq[S][0].z), "+v"(bq[S][0].w), "+v"(bq[S][1].x), "+v"(bq[S][1].y), "+v"(bq[S][1].z), "+v"(bq[S][1].w)); } while (0)
#endif
#if defined(DGP_X) && DGP_X == 5
        auto mma5 = [](const uint4& x, const uint4& y, floatx16 c) { asm volatile("" :: "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w), "v"(y.x), "v"(y.y), "v"(y.z), "v"(y.w)); return c; };
#define mma mma5
#endif
        DGP_FENCE();
        const uint4* A = a_lane;
        const uint4* B = b_lane;
        int da = A_CELLS, db = B_CELLS;            // to the other buffer and back (deep ring: to the next stage, wrapping)
        int sb_c = 0; (void)sb_c;
        for (int ks = 0; ks < nks; ++ks) {
            constexpr int S0 = 0, S1 = 1, S2 = 2;
            DGP_SPLIT_HALF(0, 0);
            DGP_SPLIT_HALF(0, 1);
            DGP_FENCE();
            DGP_RA(A, 1);
            DGP_RB(B, S2, 0, 1, 2);
            DGP_FENCE();
            DGP_GL(0, 0, S0);
            DGP_FENCE();
            DGP_RB(B, S0, 0, 0, 2);
            DGP_FENCE();
            DGP_GH(0, 0, S1);
            DGP_FENCE();
            DGP_RB(B, S1, 1, 1, 0);
            DGP_FENCE();
            DGP_GL(0, 2, S2);
            DGP_FENCE();
            DGP_RB(B, S2, 1, 0, 0);
            DGP_FENCE();
            DGP_GH(0, 2, S0);
            DGP_FENCE();
            DGP_RB(B, S0, 1, 1, 2);
            DGP_SPLIT_HALF(1, 0);
            DGP_SPLIT_HALF(1, 1);
            DGP_FENCE();
            DGP_GL(1, 0, S1);
            DGP_FENCE();
            DGP_RB(B, S1, 1, 0, 2);
            DGP_FENCE();
            DGP_GH(1, 0, S2);
            DGP_FENCE();
            DGP_GL(1, 2, S0);
            DGP_FENCE();
            DIAG_STAMP(e2);
            __syncthreads();
            DIAG_STAMP(e3);
#ifdef DGP_DIAG
            acc_mf += e2 - e1; acc_ba += e3 - e2; e1 = e3;
#endif
            if constexpr (DMA) { sa_c = sa_c == NSA - 1 ? 0 : sa_c + 1; a_cur = a_row0 + (unsigned)(sa_c * (A_CELLS * 16)); }
            else { A += da; da = -da; }
            if constexpr (NSB == 2) { B += db; db = -db; } else { B += db; if (++sb_c == NSB) { sb_c = 0; B -= NSB * B_CELLS; } }
            const bool more = ks + 1 < nks;
            if (more) {
                DGP_RA(A, 0);
                DGP_RB(B, S0, 0, 1, 0);
            }
            DGP_FENCE();
            DGP_GH(1, 2, S1);
            DGP_FENCE();
            if (more) DGP_RB(B, S1, 0, 0, 0);
            DGP_FENCE();
        }
#undef DGP_RA
#undef mma
#undef DGP_RB
#undef DGP_SPLIT_HALF
#undef DGP_GL
#undef DGP_GH
#undef DGP_FENCE
    } else
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        DIAG_STAMP(e1);
        const uint4* a_base = sA + buf * A_CELLS + wave_m0 + l31;
        const uint4* b_base = sB + buf * B_CELLS + wave_n0 + l31;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int kg = 2 * kk + half;
            uint4 af[NP][TM], bf[NP][TN];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) if (!CS) af[pl][i] = a_base[(pl * KG + kg) * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = b_base[(pl * KG + kg) * LDB + 32 * j];
            }
            if (CS) {       // k-group kg = chunks 2 kg and 2 kg + 1 of this lane's row, split here into the fp16 high / low operand
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float4 f0 = __builtin_bit_cast(float4, a_base[(2 * kg) * LDAF + 32 * i]);
                    const float4 f1 = __builtin_bit_cast(float4, a_base[(2 * kg + 1) * LDAF + 32 * i]);
#if defined(DGP_X) && DGP_X == 1
                    af[0][i] = __builtin_bit_cast(uint4, f0);
                    af[1][i] = __builtin_bit_cast(uint4, f1);
#else
                    if constexpr (AH2) {       // chunk 2 kg = high cell, chunk 2 kg + 1 = low cell of this row's k-group
                        af[0][i] = __builtin_bit_cast(uint4, f0);
                        af[1][i] = __builtin_bit_cast(uint4, f1);
                        continue;
                    }
                    uint2 h0, l0, h1, l1;
                    split2_f16(f0, scA_c, h0, l0);
                    split2_f16(f1, scA_c, h1, l1);
                    af[0][i] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    af[1][i] = make_uint4(l0.x, l0.y, l1.x, l1.y);
#endif
                }
            }
            auto mma = [](const uint4& x, const uint4& y, floatx16 c) {
                if (NT == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), c, 0, 0, 0);
                return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    floatx16 a = acc[i][j];
                    if (NT == 6) {          // smallest terms first
                        a = mma(af[NP - 1][i], bf[0][j], a);
                        a = mma(af[0][i], bf[NP - 1][j], a);
                        a = mma(af[1][i], bf[1][j], a);
                    }
                    a = mma(af[1][i], bf[0][j], a);
                    a = mma(af[0][i], bf[1][j], a);
                    a = mma(af[0][i], bf[0][j], a);
                    acc[i][j] = a;
                }
        }
        DIAG_STAMP(e2);
        __syncthreads();
        DIAG_STAMP(e3);
#ifdef DGP_DIAG
        acc_mf += e2 - e1; acc_ba += e3 - e2;
#endif
    }
#ifdef DGP_DIAG
    const unsigned long long e_loop_end = e1;        // (16x16x32 loops: the stamp behind the last barrier)
    DIAG_STAMP(e1);
#endif
    if (part >= 0) {        // raw accumulators of this K-slice -> slab [BM][BN]; tail_fixup sums the slices and applies the epilogue
        ls_store_raw<TM, TN, WN, M16>(acc, smem, wave, lane, wave_m0, wave_n0, p.slab + (size_t)tail_slot * (BM * BN), BN);
        return;
    }
#ifdef DGP_DIAG
    unsigned long long eps[4] = {0, 0, 0, 0};
    if constexpr (OH2) ls_epilogue_h2<TM, TN, WN, M16, H1>(p, acc, smem, wave, lane, m0, n0, wave_m0, wave_n0, post, epi_out_scale, epi_res_inv_scale, eps);
    else
#else
    if constexpr (OH2) ls_epilogue_h2<TM, TN, WN, M16, H1>(p, acc, smem, wave, lane, m0, n0, wave_m0, wave_n0, post, epi_out_scale, epi_res_inv_scale);
    else
#endif
    ls_epilogue<TM, TN, WN, M16>(p, acc, smem, wave, lane, m0, n0, wave_m0, wave_n0, post);
#ifdef DGP_DIAG
    DIAG_STAMP(e2);
    if (p.dbg && threadIdx.x == 0) {
        unsigned long long* d = p.dbg + 10ull * blockIdx.x;
        d[0] = t_pro; d[1] = acc_mf + acc_ba; d[2] = e2 - e1; d[4] = acc_mf; d[7] = acc_ba;
        d[9] = e1 - e_loop_end;                       // staging of the accumulators (16x16x32 loops) between the K loop and the epilogue
        if constexpr (MODE != 3) { d[3] = eps[0]; d[5] = eps[1]; d[6] = eps[2]; d[8] = eps[3]; }     // (MODE 3: slots 3 / 5 / 6 hold the ring wave's stamps)
        else d[8] = eps[3];
        unsigned long long rt1;
        unsigned hwid, xcc;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* tl = p.dbg + 10ull * 65536 + 3ull * blockIdx.x;
        tl[0] = rt0; tl[1] = rt1; tl[2] = ((unsigned long long)(xcc & 15u) << 16) | (hwid & 0x7F00u);
    }
#endif
}

// Second half of the tail K-split: out = epilogue(sum over the K-slices of a tail tile, in fixed order).
template <int BM, int BN, bool F16>
__global__ __launch_bounds__(256) void tail_fixup_kernel(const ConvArgs p) {
    const int lane = threadIdx.x & 63;
    const float post = F16 ? 1.f / ((p.in_fmt ? h2_in_scale(p, lane) : pow2_scale_for(p.in_absmax, lane, p.in2_absmax)) * pow2_scale_for(p.w_absmax, lane)) : 1.f;
    constexpr int C4 = BN / 4;
    const int per_tile = BM * C4;
    const int ntail = (int)gridDim.y;
    const int HoWo = p.Ho * p.Wo;
    const float sh_scale = p.shadow ? shadow_scale_for(p.shadow_prev, lane) : 0.f;
    float amax = 0.f;
    for (int tt = blockIdx.y; tt < ntail; tt += gridDim.y) {
        const int tile = p.n_main + tt, mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < per_tile; e += gridDim.x * blockDim.x) {
            const int row = e / C4, c4 = e - row * C4;
            const int m = mt * BM + row, co = nt * BN + 4 * c4;
            if (m >= p.M || co >= p.Cout) continue;
            const float* sl = p.slab + ((size_t)tt * p.tail_ksplit) * (BM * BN) + (size_t)row * BN + 4 * c4;
            float4 a = *reinterpret_cast<const float4*>(sl);
            for (int s = 1; s < p.tail_ksplit; ++s) {
                const float4 b = *reinterpret_cast<const float4*>(sl + (size_t)s * (BM * BN));
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            float4 sc = make_float4(post, post, post, post), bi = make_float4(0.f, 0.f, 0.f, 0.f), rr = bi;
            if (p.scale) { const float4 t = *reinterpret_cast<const float4*>(p.scale + co); sc.x *= t.x; sc.y *= t.y; sc.z *= t.z; sc.w *= t.w; }
            if (p.bias) bi = *reinterpret_cast<const float4*>(p.bias + co);
            if (p.res) {
                long long roff = -1;
                if (p.res_s == 1) roff = (long long)m * p.Cout + co;
                else {
                    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    if (p.res_s == -2) { if (!((ho | wo) & 1)) roff = (((long long)n * p.res_H + (ho >> 1)) * p.res_W + (wo >> 1)) * p.Cout + co; }
                    else roff = (((long long)n * p.res_H + ho * p.res_s) * p.res_W + wo * p.res_s) * p.Cout + co;
                }
                if (roff >= 0) rr = *reinterpret_cast<const float4*>(p.res + roff);
            }
            float4 o;
            o.x = a.x * sc.x + bi.x + rr.x; o.y = a.y * sc.y + bi.y + rr.y; o.z = a.z * sc.z + bi.z + rr.z; o.w = a.w * sc.w + bi.w + rr.w;
            if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (p.mask) {
                float4 g;
                if (p.mask_fmt == 2) {
                    const uint2 mh = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(p.mask) + ((long long)m * p.Cout + co) * 2);
                    g = h2_gate4(u32x2{mh.x, mh.y}, u32x2{0u, 0u});
                } else if (p.mask_fmt) {
                    const char* cell = reinterpret_cast<const char*>(p.mask + (long long)m * p.Cout + (co & ~7)) + ((co & 4) ? 8 : 0);
                    const uint2 mh = *reinterpret_cast<const uint2*>(cell), ml = *reinterpret_cast<const uint2*>(cell + 16);
                    g = h2_gate4(u32x2{mh.x, mh.y}, u32x2{ml.x, ml.y});
                } else g = *reinterpret_cast<const float4*>(p.mask + (long long)m * p.Cout + co);
                o.x = g.x > 0.f ? o.x : 0.f; o.y = g.y > 0.f ? o.y : 0.f; o.z = g.z > 0.f ? o.z : 0.f; o.w = g.w > 0.f ? o.w : 0.f;
            }
            *reinterpret_cast<float4*>(p.out + (long long)m * p.Cout + co) = o;
            if (sh_scale > 0.f) {
                uint2 sh, sl;
                split2_f16(o, sh_scale, sh, sl);
                if (p.shadow_fmt == 2) {
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.shadow) + ((long long)m * p.Cout + co) * 2) = sh;
                } else {
                char* cell = reinterpret_cast<char*>(p.shadow + (long long)m * p.Cout + (co & ~7)) + ((co & 4) ? 8 : 0);
                *reinterpret_cast<uint2*>(cell) = sh;
                *reinterpret_cast<uint2*>(cell + 16) = sl;
                }
            }
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    }
    if (p.out_absmax) track_absmax(p.out_absmax, amax, lane, (int)((blockIdx.y * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 6)));
}

// tail_fixup_kernel for an H2 output tensor (and an fp32 or H2 residual): a thread owns 8 channels of a row
template <int BM, int BN>
__global__ __launch_bounds__(256) void tail_fixup_h2_kernel(const ConvArgs p) {
    const int lane = threadIdx.x & 63;
    const float post = 1.f / ((p.in_fmt ? h2_in_scale(p, lane) : pow2_scale_for(p.in_absmax, lane, p.in2_absmax)) * pow2_scale_for(p.w_absmax, lane));
    constexpr int C8 = BN / 8;
    const int per_tile = BM * C8;
    const int ntail = (int)gridDim.y;
    const int HoWo = p.Ho * p.Wo;
    const float out_scale = h2_out_scale(p, lane), res_inv_scale = p.res_fmt ? h2_res_inv_scale(p, lane) : 1.f;
    float amax = 0.f;
    for (int tt = blockIdx.y; tt < ntail; tt += gridDim.y) {
        const int tile = p.n_main + tt, mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < per_tile; e += gridDim.x * blockDim.x) {
            const int row = e / C8, c8 = e - row * C8;
            const int m = mt * BM + row, co = nt * BN + 8 * c8;
            if (m >= p.M || co >= p.Cout) continue;
            const float* sl = p.slab + ((size_t)tt * p.tail_ksplit) * (BM * BN) + (size_t)row * BN + 8 * c8;
            float o[8];
            {
                const float4 a = *reinterpret_cast<const float4*>(sl), b = *reinterpret_cast<const float4*>(sl + 4);
                o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
            }
            for (int sidx = 1; sidx < p.tail_ksplit; ++sidx) {
                const float4 a = *reinterpret_cast<const float4*>(sl + (size_t)sidx * (BM * BN));
                const float4 b = *reinterpret_cast<const float4*>(sl + (size_t)sidx * (BM * BN) + 4);
                o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; o[4] += b.x; o[5] += b.y; o[6] += b.z; o[7] += b.w;
            }
            float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.res) {
                long long roff = -1;
                if (p.res_s == 1) roff = (long long)m * p.Cout + co;
                else {
                    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    roff = (((long long)n * p.res_H + ho * p.res_s) * p.res_W + wo * p.res_s) * p.Cout + co;
                }
                const uint4 r0 = *reinterpret_cast<const uint4*>(p.res + roff), r1 = *reinterpret_cast<const uint4*>(p.res + roff + 4);
                if (p.res_fmt) h2_unpack8(r0, r1, res_inv_scale, r);
                else {
                    const float4 f0 = __builtin_bit_cast(float4, r0), f1 = __builtin_bit_cast(float4, r1);
                    r[0] = f0.x; r[1] = f0.y; r[2] = f0.z; r[3] = f0.w; r[4] = f1.x; r[5] = f1.y; r[6] = f1.z; r[7] = f1.w;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float scv = p.scale ? p.scale[co + k] * post : post;
                o[k] = o[k] * scv + (p.bias ? p.bias[co + k] : 0.f) + r[k];
                if (p.relu) o[k] = fmaxf(o[k], 0.f);
                amax = fmaxf(amax, fabsf(o[k]));
            }
            uint4 hi, lo;
            h2_pack8(o, out_scale, hi, lo);
            uint4* dst = reinterpret_cast<uint4*>(p.out + (long long)m * p.Cout + co);
            dst[0] = hi; dst[1] = lo;
        }
    }
    if (p.out_absmax) track_absmax(p.out_absmax, amax, lane, (int)((blockIdx.y * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 6)));
}

template <int BM, int BN, int NT, int BK, int CW = 4>
static hipError_t launch_conv_split(ConvArgs a, hipStream_t s) {
    constexpr int NP = NT == 6 ? 3 : 2;
    constexpr int KG = BK / 8;
    if (NT == 2 && ((!a.in_absmax && !a.in_fmt) || !a.w_absmax)) return hipErrorInvalidValue;      // fp16 split needs both ranges (H2 input carries its scale)
    if ((a.in_fmt || a.out_fmt) && !(NT == 2 && BK == 32 && CW == 4 && a.wh3 && (a.Cout % 8) == 0 && (a.Cin % 8) == 0 && !a.up && !a.stem &&
                                     (!a.mask || (a.in_fmt == 2 && a.out_fmt == 2 && a.mask_fmt == 2))))
        return hipErrorInvalidValue;                                                 // H2 tensors: fp16-split cell kernels only (an H1 gate: H1 -> H1 launches)
    // 16-bit tier (in_fmt 2: H1 cells, already in 4-byte units here -- launch_conv): H1 or fp32 output, H1 residual, no predicted scales
    const bool h1 = a.in_fmt == 2;
    if (h1 != (a.out_fmt == 2) && a.out_fmt != 0) return hipErrorInvalidValue;
    if (a.out_fmt == 2 && !h1) return hipErrorInvalidValue;
    if (a.res && (a.res_fmt == 2) != (a.out_fmt == 2)) return hipErrorInvalidValue;
    if (h1 && a.shadow) return hipErrorInvalidValue;
    if (a.res && a.res_s == -2 && !(h1 && a.out_fmt == 2)) { if (a.in_fmt) return hipErrorInvalidValue; }
    // non-temporal residual loads / output stores in the H2 epilogue (A/B switch DGP_EPI_NT; 0: off, 1: every layer, 2 (default): only
    // layers with >= 8 column tiles (N >= 1024: conv3 of block3 / block4), 3 / 4: their loads / stores only).  There the 128 KB a tile
    // streams through the epilogue evict the A rows the other column tiles of the row block still read: PMC FETCH_SIZE 1061 MB per
    // launch against 393 MB of operands on block4's conv3.  Measured (same box, ms per launch, nt 0 / 2): block3 conv3 0.105 -> 0.088,
    // the conv1 that re-reads the tensor 0.070 -> 0.076, block4 conv3 0.268 -> 0.253; everywhere (1) loses: a small output written nt
    // (R2: 39 MB) is no longer cache-resident for its consumer
    static const int epi_nt_env = dgp_tune("DGP_EPI_NT", 2);
    a.epi_nt = epi_nt_env == 1 ? 3 : (a.CoutP / BN >= 8 ? (epi_nt_env == 2 ? 3 : epi_nt_env == 3 ? 1 : epi_nt_env == 4 ? 2 : 0) : 0);   // bit 0: loads, bit 1: stores
    static const int tap_minor = dgp_tune("DGP_TAP_MINOR", 1);
    a.tap_minor = (tap_minor && !a.stem && a.ntaps > 1 && a.nk * 32 == a.ntaps * a.Cin) ? 1 : 0;
    const size_t smem_loop = (size_t)2 * (NP * KG * (BM + (BK == 32 ? 4 : 8)) + NP * KG * (BN + 4)) * 16;
    const size_t smem_epi = (size_t)CW * 32 * (BN + 4) * 4;        // (upper bound: 4 x 1 compute-wave layout of the CS kernels)
    size_t smem = smem_loop > smem_epi ? smem_loop : smem_epi;
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.CoutP + BN - 1) / BN;
    if (a.CoutP % BN != 0 || (a.Cin < 32 && !a.stem) || a.out_mode != 0) return hipErrorInvalidValue;
    if (a.tap_rows == 0) a.tap_rows = a.Cin >> 2;
    if (a.in2 && (a.ntaps != 1 || a.stride != 1 || a.up || a.cin_split % 32 || (a.Cin - a.cin_split) % 32 || a.cin_split <= 0 ||
                  a.cin_split >= a.Cin || a.H != a.Ho || a.W != a.Wo)) return hipErrorInvalidValue;
    // loader specialisation (template MODE): pointwise / plain / everything-at-run-time
    const bool pointwise = a.ntaps == 1 && a.stride == 1 && a.pad_t == 0 && a.pad_l == 0 && !a.up && !a.stem && a.H == a.Ho && a.W == a.Wo;
    const int mode = (pointwise && (unsigned long long)a.M * (a.in2 ? a.cin_split : a.Cin) * 4ull == a.in_bytes &&
                      (!a.in2 || (unsigned long long)a.M * (a.Cin - a.cin_split) * 4ull == a.in2_bytes) && a.in_bytes < 4200000000u)
                         ? 2 : ((a.up || a.in2 || a.stem) ? 0 : 1);
    // halo walk (MODE 3, see the kernel): 3x3 / stride 1 convs of the H2 engine on 128 x 128 tiles whose tap shifts fit the pixel ring.
    // A/B switch DGP_HALO=0
    static const int halo_env = dgp_env("DGP_HALO", 1);
    const bool halo = halo_env && BM == 128 && BN == 128 && CW == 4 && NT == 2 && BK == 32 && mode == 1 && a.in_fmt && a.out_fmt && a.wh3 &&
                      a.KH == 3 && a.KW == 3 && a.ntaps == 9 && a.stride == 1 && a.dil >= 1 && a.pad_t == a.dil && a.pad_l == a.dil && a.H == a.Ho &&
                      a.W == a.Wo && a.W >= 2 && (a.Cin % 32) == 0 && a.nk * 32 == 9 * a.Cin && a.tap_rows == (a.Cin >> 2) && !a.in_scale_dev &&
                      (unsigned long long)a.M * a.Cin * 4ull == a.in_bytes && a.in_bytes < 4000000000u &&
                      a.dil * (a.W - 2) + 128 <= HALO_C - 8 && (unsigned long long)a.dil * (a.W + 1) * a.Cin * 4ull < 200000000ull &&
                      // a strip longer than the ring still works (the ring slides), but the loaders can then never run a chunk ahead and
                      // the walk measured 0.4 % behind the per-tap loaders (ResNet-101 1280 x 720: W = 80 at dilation 2); DGP_HALO=2 forces it
                      (halo_env >= 2 || 128 + 2 * a.dil * (a.W + 1) <= HALO_C);
    constexpr bool CAN_PB = NT == 2 && CW == 4;
    if (!(CAN_PB && a.wh3)) a.wh3 = nullptr;
    auto kern = conv_igemm_split_ls<BM, BN, NT, BK, CW, false, 0>;
    if (a.wh3) kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_PB, 2>
                    : mode == 1 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_PB, 1> : conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_PB, 0>;
    constexpr bool CAN_CS = CAN_PB && BK == 32;
    // 2 (default): every fp16 kernel with pre-split weights; 1: 128 x 128 tiles only; 0: loaders split (A/B switch)
    static const int cs_env = dgp_tune("DGP_COMPUTE_SPLIT", 2);
    const bool cs = CAN_CS && a.wh3 && (cs_env >= 2 || (cs_env == 1 && BN == 128));
    if (cs) kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_CS, 2, CAN_CS>
                 : mode == 1 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_CS, 1, CAN_CS> : conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_CS, 0, CAN_CS>;
    else kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, false, 2>
              : mode == 1 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, false, 1> : conv_igemm_split_ls<BM, BN, NT, BK, CW, false, 0>;
    // LDS-DMA loaders (A/B switch DGP_DMA=0): 128 x 128 CS kernels with the plain or the pointwise walk
    constexpr bool CAN_DMA = CAN_CS && BM == 128 && (BN == 128 || BN == 64);
    static const int dma_env = dgp_tune("DGP_DMA", 1);
    const bool dma = CAN_DMA && cs && mode != 0 && dma_env;
    // deep DMA ring (see the kernel) for grids of at most one tile per CU; A/B switch DGP_DEEP_RING=0, =2: every DMA launch
    constexpr bool CAN_DEEP = CAN_DMA && BN == 128;
    static const int deep_env = dgp_tune("DGP_DEEP_RING", 1);
    bool deep = false, use_halo = false;
    if (dma) {
        kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DMA, 2, CAN_DMA, CAN_DMA>
                         : conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DMA, 1, CAN_DMA, CAN_DMA>;
        smem = (size_t)(3 * BM * 8 + 2 * NP * KG * BN) * 16;           // 3 A stages + 2 B stages = 80 KB
        static int n_cu_d = 0;
        if (!n_cu_d) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu_d, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu_d <= 0) n_cu_d = 256; }
        const long long ntile = (long long)((a.M + BM - 1) / BM) * ((a.CoutP + BN - 1) / BN);
        deep = CAN_DEEP && deep_env && (deep_env == 2 || ntile <= (long long)n_cu_d) && a.nk >= 6 && !h1;
        if (deep) {
            if constexpr (CAN_DEEP)
                kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DEEP, 2, CAN_DEEP, CAN_DEEP, false, false, CAN_DEEP>
                                 : conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DEEP, 1, CAN_DEEP, CAN_DEEP, false, false, CAN_DEEP>;
            smem = (size_t)(5 * BM * 8 + 4 * NP * KG * BN) * 16;       // 5 A stages + 4 B stages = 144 KB
        }
        if (smem < smem_epi) smem = smem_epi;
    }
    if (h1) {              // H1 input (16-bit tier): LDS-DMA kernels only; H1 output (O1 epilogue) or fp32 output (the heads' pointwise GEMM)
        if (!cs || !dma) return hipErrorInvalidValue;
        if constexpr (CAN_DMA) {
            if (a.out_fmt) {
                kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DMA, 2, CAN_DMA, CAN_DMA, CAN_DMA, CAN_DMA, false, CAN_DMA>
                                 : conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DMA, 1, CAN_DMA, CAN_DMA, CAN_DMA, CAN_DMA, false, CAN_DMA>;
                if constexpr (CAN_DEEP) {
                    if (halo) {
                        kern = conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DEEP, CAN_DEEP ? 3 : 1, CAN_DEEP, CAN_DEEP, CAN_DEEP, CAN_DEEP, false, CAN_DEEP>;
                        smem = (size_t)HALO_B0 + 2 * (size_t)(NP * KG * BN) * 16;
                        if (smem < smem_epi) smem = smem_epi;
                        use_halo = true;
                    }
                }
            } else {
                if (mode != 2 || a.res) return hipErrorInvalidValue;
                kern = conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DMA, 2, CAN_DMA, CAN_DMA, CAN_DMA, false, false, CAN_DMA>;
            }
        }
    } else if (a.in_fmt) {        // H2 input: the same kernels without the split (AH2); H2 output: epilogue variant (OH2)
        if (!cs) return hipErrorInvalidValue;
        if constexpr (CAN_CS) {
            if (a.out_fmt) {
                if (dma) {
                    if constexpr (CAN_DMA)
                        kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DMA, 2, CAN_DMA, CAN_DMA, CAN_DMA, CAN_DMA>
                                         : conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DMA, 1, CAN_DMA, CAN_DMA, CAN_DMA, CAN_DMA>;
                    if constexpr (CAN_DEEP) {
                        if (halo && !deep) {
                            kern = conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DEEP, CAN_DEEP ? 3 : 1, CAN_DEEP, CAN_DEEP, CAN_DEEP, CAN_DEEP>;
                            smem = (size_t)HALO_B0 + 2 * (size_t)(NP * KG * BN) * 16;
                            if (smem < smem_epi) smem = smem_epi;
                            use_halo = true;
                        }
                    }
                    if constexpr (CAN_DEEP) {
                        if (deep) kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DEEP, 2, CAN_DEEP, CAN_DEEP, CAN_DEEP, CAN_DEEP, CAN_DEEP>
                                                   : conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DEEP, 1, CAN_DEEP, CAN_DEEP, CAN_DEEP, CAN_DEEP, CAN_DEEP>;
                    }
                } else {
                    kern = mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_CS, 2, CAN_CS, false, CAN_CS, CAN_CS>
                         : mode == 1 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_CS, 1, CAN_CS, false, CAN_CS, CAN_CS>
                                     : conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_CS, 0, CAN_CS, false, CAN_CS, CAN_CS>;
                }
            } else {           // fp32 output: pointwise GEMMs only (the heads, layer tests)
                if (mode != 2) return hipErrorInvalidValue;
                if (a.res && a.res_fmt) return hipErrorInvalidValue;      // (its epilogue adds an fp32 residual: H2 cells would be read as floats)
                if (dma) {
                    if constexpr (CAN_DMA) kern = conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DMA, 2, CAN_DMA, CAN_DMA, CAN_DMA, false>;
                    if constexpr (CAN_DEEP) if (deep) kern = conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_DEEP, 2, CAN_DEEP, CAN_DEEP, CAN_DEEP, false, CAN_DEEP>;
                } else kern = conv_igemm_split_ls<BM, BN, NT, BK, CW, CAN_CS, 2, CAN_CS, false, CAN_CS, false>;
            }
        }
    } else if (a.out_fmt) return hipErrorInvalidValue;
    static bool attr_done_dev[16][5][5][4] = {};
    auto& attr_done = attr_done_dev[dgp_device_slot()];
    const int mode_slot = use_halo ? 3 : mode;
    const int fmt_slot = h1 ? (a.out_fmt ? 4 : 3) : a.in_fmt ? (a.out_fmt ? 2 : 1) : 0, path_slot = deep ? 4 : dma ? 3 : cs ? 2 : (a.wh3 ? 1 : 0);
    if (!attr_done[fmt_slot][path_slot][mode_slot]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done[fmt_slot][path_slot][mode_slot] = true;
    }
    long long nwg = (long long)a.mtiles * a.ntiles;
    // Grid tail: with `slots` workgroups resident, the last tiles % slots tiles run on a mostly idle chip.  Split their K range
    // over up to 4 blocks each (raw slabs + a fixup pass that sums them in fixed order: deterministic) so the last round is full.
    a.tail_ksplit = 0; a.n_main = (int)nwg;
    static const int tail_env = dgp_env("DGP_TAIL_SPLIT", 1);      // A/B switch
    // (H2 tensors: with the split gone from the K loop the K-split of the tail + its fix-up launch no longer pays -- same-box A/B,
    //  block3 conv1 -7..-24 %, block4 conv1 -12 %, block4 conv2 -5 % without it, +2.5 % end to end -- so tail_env == 2 is needed to force it)
    if (tail_env && (!a.in_fmt || tail_env == 2) && !h1 && a.slab && CW == 4 && BN == 128 && !use_halo) {      // (128 x 64 tiles fit three per CU and gain nothing)
        static int n_cu = 0;
        if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
        const long long slots = 2LL * n_cu;            // two workgroups of these kernels fit a CU
        const long long rem = nwg % slots;
        const int nks = a.nk * (32 / BK);
        // (a grid smaller than one round is left alone: at the trainer's 11-frame batches splitting it measured 1.5 % slower per step)
        if (rem > 0 && nwg >= slots && rem * 4 <= slots * 3) {
            int ks = (int)(slots / rem);
            if (ks > 4) ks = 4;
            while (ks > 1 && (nks % ks != 0 || nks / ks < 4)) --ks;
            // the fixup pass costs ~10-15 us: with two or more full rounds in front of the tail it only pays for deep-K layers
            // (measured per layer with DGP_TAIL_SPLIT=0/1: block3 -17 %, block4 N=512 -3..-9 %, block2 +-5 % -> left alone)
            if (nwg / slots >= 2 && nks < 48) ks = 1;
            if (ks > 1 && (unsigned long long)rem * ks * BM * BN * 4ull <= a.slab_bytes) {
                a.tail_ksplit = ks; a.n_main = (int)(nwg - rem);
                nwg = a.n_main + rem * ks;
            }
        }
    }
#ifdef DGP_DIAG
    static unsigned long long* dbg_buf = nullptr;
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, 13 * 8 * 65536);
    a.dbg = nwg <= 65536 ? dbg_buf : nullptr;
#endif
    // supertile order (see the kernel): H2 engine, 128-wide tiles, >= 8 column tiles and a weight panel that cannot stay in an XCD's L2
    a.st_gn = 0;
    {
        static const int st_env = dgp_tune("DGP_SUPERTILE", 1);       // A/B switch; > 1: force the row-chunk size
        const long long panel_bytes = (long long)a.nk * 32 * a.CoutP * 4;
        if (st_env && a.in_fmt && a.out_fmt && a.tail_ksplit <= 1 && BN == 128 && a.ntiles >= 8 && panel_bytes > (3LL << 20) && a.mtiles >= 64) {
            const long long per_ntile = (long long)a.nk * 32 * BN * 4;
            int gn = 1;
            static const long long st_cap = (long long)dgp_tune("DGP_ST_CAP_KB", 1536) << 10;
            while (gn * 2 <= a.ntiles && (long long)(gn * 2) * per_ntile <= st_cap && a.ntiles % (gn * 2) == 0) gn *= 2;      // <= 1.5 MB of weight cells per group
            a.st_gn = gn; a.st_mc = st_env > 1 ? st_env : 8; a.st_mb = (a.mtiles + 7) / 8;
            nwg = 8LL * a.st_mb * a.ntiles;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * (CW + 4)), smem, s, a);
    if (a.tail_ksplit > 1) {
        const int ntail = (int)((long long)a.mtiles * a.ntiles - a.n_main);
        if (a.out_fmt) hipLaunchKernelGGL((tail_fixup_h2_kernel<BM, BN>), dim3(BM * BN / 8 / 256, ntail), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((tail_fixup_kernel<BM, BN, NT == 2>), dim3(BM * BN / 4 / 256, ntail), dim3(256), 0, s, a);
    }
#ifdef DGP_DIAG
    if (a.dbg) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h(10 * nwg);
        (void)hipMemcpy(h.data(), a.dbg, 80 * nwg, hipMemcpyDeviceToHost);
        double v[8] = {0};
        for (long long b = 0; b < nwg; ++b) for (int k = 0; k < 8; ++k) v[k] += (double)h[10 * b + k];
        for (int k = 0; k < 8; ++k) v[k] /= (double)nwg;
        const int nks = a.nk * (32 / BK);
        int occ = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * (CW + 4), smem);
        hipFuncAttributes fa{};
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
        printf("[diag occupancy] blocks/CU %d (smem %zu B, regs %d, static smem %zu, local %zu)\n", occ, smem, fa.numRegs,
               fa.sharedSizeBytes, fa.localSizeBytes);
        printf("[diag split %dx%d NT%d BK%d CW%d] tiles %lld K-steps %d | compute wave 0: first barrier %.0f cyc, epilogue %.0f | per K-step: "
               "ldsread+mfma %.0f barrier-wait %.0f || loader wave: wait+split+ds_write %.0f load-issue %.0f barrier-wait %.0f\n",
               BM, BN, NT, BK, CW, nwg, nks, v[0], v[2], v[4] / nks, v[7] / nks, v[3] / nks, v[5] / nks, v[6] / nks);
        double v8 = 0; for (long long b = 0; b < nwg; ++b) v8 += (double)h[10 * b + 8];
        double v9 = 0; for (long long b = 0; b < nwg; ++b) v9 += (double)h[10 * b + 9];
        {   // per-CU timeline of the launch: how many workgroups a CU holds over the launch's span, and how long a freed slot stays empty
            std::vector<unsigned long long> tl(3 * nwg);
            (void)hipMemcpy(tl.data(), a.dbg + 10ull * 65536, 24 * nwg, hipMemcpyDeviceToHost);
            if (const char* dump = getenv("DGP_DIAG_DUMP")) {      // raw (start, end, CU) of every workgroup of every launch, for scripts/cu_phase.py
                if (FILE* f = fopen(dump, "a")) {
                    fprintf(f, "# launch %dx%d tiles %lld K-steps %d\n", BM, BN, nwg, nks);
                    for (long long b = 0; b < nwg; ++b) fprintf(f, "%llu %llu %llu\n", tl[3 * b], tl[3 * b + 1], tl[3 * b + 2]);
                    fclose(f);
                }
            }
            unsigned long long t_lo = ~0ull, t_hi = 0;
            std::vector<std::pair<unsigned long long, std::pair<unsigned long long, int>>> ev;      // (cu key, (time, +1 / -1))
            double dur = 0; long long nz = 0;
            for (long long b = 0; b < nwg; ++b) {
                const unsigned long long s0 = tl[3 * b], s1 = tl[3 * b + 1], k = tl[3 * b + 2];
                if (!s1 || s1 < s0) continue;
                ++nz; dur += (double)(s1 - s0);
                if (s0 < t_lo) t_lo = s0;
                if (s1 > t_hi) t_hi = s1;
                ev.push_back({k, {s0, +1}}); ev.push_back({k, {s1, -1}});
            }
            std::sort(ev.begin(), ev.end(), [](const auto& x, const auto& y) {
                return x.first != y.first ? x.first < y.first : x.second.first != y.second.first ? x.second.first < y.second.first : x.second.second < y.second.second; });
            double res_t[4] = {0, 0, 0, 0}, gap_sum = 0; long long gaps = 0; int ncu = 0;
            for (size_t i = 0; i < ev.size();) {
                size_t j = i; while (j < ev.size() && ev[j].first == ev[i].first) ++j;
                ++ncu;
                int cnt = 0; unsigned long long tprev = t_lo, last_end = 0; bool pending = false;
                for (size_t e = i; e < j; ++e) {
                    const unsigned long long tt = ev[e].second.first;
                    res_t[cnt > 3 ? 3 : cnt] += (double)(tt - tprev); tprev = tt;
                    if (ev[e].second.second > 0) { if (pending) { gap_sum += (double)(tt - last_end); ++gaps; pending = false; } ++cnt; }
                    else { --cnt; last_end = tt; pending = true; }
                }
                res_t[0] += (double)(t_hi - tprev);
                i = j;
            }
            const double span = (double)(t_hi - t_lo), tot = span * ncu;
            if (nz && span > 0)
                printf("[diag timeline] span %.2f us on %d CUs | workgroup life %.2f us (%.0f shader cycles -> %.2f GHz) | CU time holding 0 / 1 / 2 / 3+ workgroups: %.3f %.3f %.3f %.3f | "
                       "a freed slot waits %.2f us for its next workgroup (%lld hand-overs)\n",
                       span / 100.0, ncu, dur / nz / 100.0, v[0] + v[1] + v[2] + v9 / nwg, (v[0] + v[1] + v[2] + v9 / nwg) / (dur / nz * 10.0),
                       res_t[0] / tot, res_t[1] / tot, res_t[2] / tot, res_t[3] / tot, gaps ? gap_sum / gaps / 100.0 : 0.0, gaps);
        }
        printf("[diag epilogue, DMA kernels] last MFMAs + staging %.0f | set-up (scale / bias, residual requests) %.0f | chunk 0 %.0f | chunks 1.. %.0f | absmax %.0f (res %d)\n",
               v9 / nwg, v[3], v[5], v[6], v8 / nwg, a.res ? 1 : 0);
    }
#endif
    return hipGetLastError();
}

// The 256 x 128 tile (see conv_igemm_split_ls, "BM = 256"): cell -> cell convolutions of either tier (H1 -> H1, H2 -> H2), pointwise (incl. the
// K-concatenated shortcut) or plain taps.  `a` holds REAL channels here (before to_h1_units): conv_kernel_name and launch_conv ask the same
// question.  DGP_W64: 0 (default) never -- it measured SLOWER than the 128 x 128 tile on every layer of the bench workload in BOTH tiers
// (one stream, profiles/r6_w64_ab_layers.txt / r6_w64_h2_ab_layers.txt: 16-bit tier block4 +6..+26 %, block3 +14..+28 %; parity tier
// block4 +8..+19 %, block3 +12..+34 %; EXPERIMENTS.md R6 (1) has the stamps and the micro-benchmarks that say why); any other value:
// wherever the kernel can run (the layer tests, the bit-identity tests of both tiers).
static int w64_env() { static const int v = dgp_env("DGP_W64", 0); return v; }
static bool conv_w64_eligible(const ConvArgs& a) {
    const int fmt = a.in_fmt;
    if (!w64_env() || (fmt != 1 && fmt != 2) || a.out_fmt != fmt || !a.wh3 || a.up || a.stem || a.out_mode != 0 || a.shadow) return false;
    if (a.res && a.res_fmt != fmt && !(fmt == 1 && a.res_fmt == 0)) return false;
    if (a.in_scale_dev || a.out_scale_dev || a.res_scale_dev) { if (fmt == 1) return false; }      // (the parity trainer's predicted scales: 128-row kernels)
    if (a.mask && !(fmt == 2 && a.mask_fmt == 2)) return false;
    const int cq = fmt == 2 ? 63 : 31;                                 // channels per K-step - 1
    if (a.CoutP % 128 != 0 || (a.Cout % 8) || (a.Cin & cq) || (fmt == 2 && (a.nk & 1))) return false;
    const bool pointwise = a.ntaps == 1 && a.stride == 1 && a.pad_t == 0 && a.pad_l == 0 && a.H == a.Ho && a.W == a.Wo;
    if (a.in2 && (!pointwise || (a.cin_split & cq) || ((a.Cin - a.cin_split) & cq))) return false;
    if (pointwise && ((unsigned long long)a.M * (a.in2 ? a.cin_split : a.Cin) * 4ull != a.in_bytes ||
                      (a.in2 && (unsigned long long)a.M * (a.Cin - a.cin_split) * 4ull != a.in2_bytes) || a.in_bytes >= 4200000000u)) return false;
    if (!pointwise && a.nk * 32 != a.ntaps * a.Cin) return false;      // (plain taps: whole channel chunks per tap)
    return true;
}

static hipError_t launch_conv_w64(ConvArgs a, hipStream_t s) {       // (`a` already in H1 units where the tensors are H1)
    constexpr int BM = 256, BN = 128, NT = 2, BK = 32, CW = 4, NP = 2, KG = 4;
    static const int tap_minor = dgp_tune("DGP_TAP_MINOR", 1);
    a.tap_minor = (tap_minor && a.ntaps > 1 && a.nk * 32 == a.ntaps * a.Cin) ? 1 : 0;
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = a.CoutP / BN;
    if (a.tap_rows == 0) a.tap_rows = a.Cin >> 2;
    static const int epi_nt_env = dgp_tune("DGP_EPI_NT", 2);
    a.epi_nt = epi_nt_env == 1 ? 3 : (a.ntiles >= 8 ? (epi_nt_env == 2 ? 3 : epi_nt_env == 3 ? 1 : epi_nt_env == 4 ? 2 : 0) : 0);
    a.tail_ksplit = 0; a.st_gn = 0;
    const bool pointwise = a.ntaps == 1 && a.stride == 1 && a.pad_t == 0 && a.pad_l == 0 && a.H == a.Ho && a.W == a.Wo;
    const int mode = pointwise ? 2 : 1;
    const bool h1 = a.in_fmt == 2;
    auto kern = h1 ? (mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, true, 2, true, true, true, true, false, true>
                                : conv_igemm_split_ls<BM, BN, NT, BK, CW, true, 1, true, true, true, true, false, true>)
                   : (mode == 2 ? conv_igemm_split_ls<BM, BN, NT, BK, CW, true, 2, true, true, true, true, false, false>
                                : conv_igemm_split_ls<BM, BN, NT, BK, CW, true, 1, true, true, true, true, false, false>);
    size_t smem = (size_t)(3 * BM * 8 + 2 * NP * KG * BN) * 16;                // 3 A stages + 2 B stages = 128 KB
    const size_t smem_epi = (size_t)CW * 64 * (BN + 4) * 4;                    // 132 KB: the four 64-row wave tiles of the epilogue
    if (smem < smem_epi) smem = smem_epi;
    static bool attr_dev[16][2][3] = {};
    bool& attr = attr_dev[dgp_device_slot()][h1 ? 1 : 0][mode];
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    a.n_main = a.mtiles * a.ntiles;
#ifdef DGP_DIAG
    static unsigned long long* dbg_buf = nullptr;
    if (!dbg_buf) (void)hipMalloc(&dbg_buf, 10 * 8 * 65536);
    a.dbg = a.n_main <= 65536 ? dbg_buf : nullptr;
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)a.n_main), dim3(64 * (CW + 4)), smem, s, a);
#ifdef DGP_DIAG
    if (a.dbg) {
        (void)hipStreamSynchronize(s);
        const long long nwg = a.n_main;
        std::vector<unsigned long long> h(10 * nwg);
        (void)hipMemcpy(h.data(), a.dbg, 80 * nwg, hipMemcpyDeviceToHost);
        double v[10] = {0};
        for (long long b = 0; b < nwg; ++b) for (int k = 0; k < 10; ++k) v[k] += (double)h[10 * b + k];
        for (int k = 0; k < 10; ++k) v[k] /= (double)nwg;
        printf("[diag w64 256x128 mode %d] tiles %lld K-steps %d | compute wave 0: first barrier %.0f cyc, staging %.0f, epilogue %.0f | per K-step: "
               "ldsread+mfma %.0f barrier-wait %.0f\n", mode, nwg, a.nk, v[0], v[9], v[2], v[4] / a.nk, v[7] / a.nk);
    }
#endif
    return hipGetLastError();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool WIDE>
static hipError_t launch_conv_t(ConvArgs a, hipStream_t s) {
    size_t smem = (size_t)(2 * 8 * (BM + 1) + 2 * 8 * BN) * 16;
    a.mtiles = (a.M + BM - 1) / BM;
    a.ntiles = (a.CoutP + BN - 1) / BN;
#ifdef DGP_DIAG
    static unsigned long long* dbg_buf = nullptr;
    {
        const long long nb = (long long)a.mtiles * a.ntiles;
        if (!dbg_buf) (void)hipMalloc(&dbg_buf, 10 * 8 * 65536);
        a.dbg = nb <= 65536 ? dbg_buf : nullptr;
    }
#endif
    if (a.CoutP % BN != 0) return hipErrorInvalidValue;       // weight panels are padded to the tile
    auto kern = conv_igemm_f32<BM, BN, WAVES_M, WAVES_N, WIDE>;
    static bool attr_done_dev[16] = {};   // per instantiation and device
    bool& attr_done = attr_done_dev[dgp_device_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (a.ksplit > 1 && (a.out_mode != 1 || a.nk % a.ksplit != 0)) return hipErrorInvalidValue;
    const long long nwg = (long long)a.mtiles * a.ntiles * (a.ksplit > 1 ? a.ksplit : 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * WAVES_M * WAVES_N), smem, s, a);
#ifdef DGP_DIAG
    if (a.dbg) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h(10 * nwg);
        (void)hipMemcpy(h.data(), a.dbg, 80 * nwg, hipMemcpyDeviceToHost);
        double v[8] = {0};
        for (long long b = 0; b < nwg; ++b) for (int k = 0; k < 8; ++k) v[k] += (double)h[10 * b + k];
        for (int k = 0; k < 8; ++k) v[k] /= (double)nwg;
        printf("[diag %dx%d] tiles %lld nk %d | cycles/tile: prologue %.0f loop %.0f epilogue %.0f | per K-step (wave 0): "
               "gload-issue %.0f mfma+ldsread %.0f vmcnt-wait %.0f lds-store %.0f barrier %.0f\n", BM, BN, nwg, a.nk,
               v[0], v[1], v[2], v[3] / a.nk, v[4] / a.nk, v[5] / a.nk, v[6] / a.nk, v[7] / a.nk);
    }
#endif
    return hipGetLastError();
}

// Tile choice, from scripts/conv_sweep.py on MI355X (ResNet-50 640x480 batch-32 shapes, TFLOP/s for
// 128x128 / 128x64 / 64x64): the small 64x64 tile (4 workgroups per CU, finest tail) wins for the
// K-heavy 3x3 convs and for 1x1 convs with a deep K and few output channels; 128x64 wins for the
// shallow-K, wide-N 1x1 convs (fewer re-reads of the activation rows); a 128x128 tile shared by 8 waves (same
// 16 waves per CU as the 64x64 tile, half the L2 traffic per FLOP) wins when Cout >= 512; 128x128 with 4 waves never won.
int pick_tile(int M, int CoutP, int K, bool have_absmax) {
    (void)M;
    if (CoutP <= 32) return TILE_128x32;
    if (const char* f = getenv("DGP_FORCE_TILE")) {     // tuning experiments only
        const int v = atoi(f);
        if ((v == TILE_128x128 || v == TILE_128x128_W8 || v == TILE_128x128_LS || v == TILE_128x128_S6 ||
             v == TILE_128x128_S3 || v == TILE_128x128_S6K16 || v == TILE_128x128_S3K16 || v == TILE_128x128_S6K16W8 ||
             ((v == TILE_128x128_H3K16 || v == TILE_128x128_H3K16W8 || v == TILE_128x128_H3K32) && have_absmax)) &&
            CoutP % 128 == 0) return v;
        if ((v == TILE_128x64_LS || v == TILE_128x64_S6 || (v == TILE_128x64_H3 && have_absmax)) && CoutP % 64 == 0) return v;
        if ((v == TILE_128x64 || v == TILE_64x64) && CoutP % 64 == 0) return v;
    }
    // rule 4 (default): fp16-split kernels where the caller tracks operand ranges (3 fp16 MFMAs per fp32-class
    //         product), bf16-split elsewhere;
    // rule 3 (DGP_CONV_MODE=bf16x6): bf16-split kernels (6 bf16 MFMAs per product, no range requirement);
    // rule 2 (DGP_CONV_MODE=f32): fp32 MFMA everywhere (bitwise fmaf chains)
    static const int rule = dgp_tune("DGP_TILE_RULE", 0) ? dgp_tune("DGP_TILE_RULE", 0)
                            : !getenv("DGP_CONV_MODE") ? 4
                            : !strcmp(getenv("DGP_CONV_MODE"), "f16") ? 4
                            : !strcmp(getenv("DGP_CONV_MODE"), "f32") ? 2
                            : !strcmp(getenv("DGP_CONV_MODE"), "bf16x6") ? 3 : 4;
    if (rule >= 4 && have_absmax) {
        // fp16 high/low split (3 MFMAs per product; needs the operand ranges).  With the MFMA work cut to a quarter
        // of the fp32 pipe's, the L2 -> L1 operand path sets the pace: the BK = 32 staging (a lane group fetches a
        // whole 128-byte line per pixel; BK = 16 fetches half lines) won every shape in scripts/split_sweep.py, and two
        // planes of fp16 leave room for two such workgroups per CU
        if (CoutP % 128 == 0) return TILE_128x128_H3K32;
        if (CoutP % 64 == 0) return TILE_128x64_H3;
    }
    if (rule >= 3) {
        // 128x128 / BK 16 (two workgroups per CU) won or tied on every shape with Cout % 128 == 0 in
        // scripts/split_sweep.py; Cout = 64 layers take the 128x64 tile
        if (CoutP % 128 == 0) return TILE_128x128_S6K16;
        if (CoutP % 64 == 0) return TILE_128x64_S6;
    }
    if (rule >= 2) {
        // loader-specialised 128x64 (4 compute + 4 loader waves, 3 workgroups per CU, loads two K-steps ahead) won or
        // tied on every Cin >= 32 layer except the very wide 1x1 convs, where the 8-wave 128x128 tile is ahead
        if (CoutP >= 1024) return TILE_128x128_W8;
        if (K >= 576) return TILE_128x64_LS;     // 3x3 convs and deep-K 1x1 reductions (MFMA-bound)
        // shallow-K 1x1 convs of block1/2 are HBM/L2-bound: all 256 threads loading beats 4 loader waves
        return CoutP >= 512 ? TILE_128x128_W8 : TILE_128x64;
    }
    if (rule >= 1 && CoutP >= 512 && !(K >= 4096)) return TILE_128x128_W8;   // wide-N 1x1 convs: 8 waves share a 128x128 tile
    if (K >= 1024 && CoutP <= 512) return TILE_64x64;      // 3x3 convs, deep-K 1x1 reductions
    if (K >= 576 && CoutP <= 256) return TILE_64x64;       // 3x3 convs of block1/2
    return TILE_128x64;
}

// Name of the kernel launch_conv will run for (args, tile) -- for the profile table / roofline accounting.
const char* conv_kernel_name(const ConvArgs& a, int tile_cfg) {
    if ((a.Cin >= 32 || a.stem) && a.out_mode == 0) {
        switch (tile_cfg) {
            case TILE_128x128_S6K16: return "split6_128x128_k16";
            case TILE_128x128_S6K16W8: return "split6_128x128_k16w8";
            case TILE_128x128_H3K16:   return "splith3_128x128_k16";
            case TILE_128x128_H3K16W8: return "splith3_128x128_k16w8";
            case TILE_128x128_H3K32:   return a.in_fmt == 2 ? (conv_w64_eligible(a) ? "h1_256x128_k64" : "h1_128x128_k64")
                                              : (conv_w64_eligible(a) ? "splith3_256x128_k32" : "splith3_128x128_k32");
            case TILE_128x64_H3:       return a.in_fmt == 2 ? "h1_128x64_k64" : "splith3_128x64_k32";
            case TILE_128x128_S6:    return "split6_128x128_k32";
            case TILE_128x64_S6:     return "split6_128x64_k32";
            case TILE_128x128_S3K16: return "split3_128x128_k16";
            case TILE_128x128_S3:    return "split3_128x128_k32";
            default: break;
        }
    }
    return "f32";
}

// 16-bit tier: ConvArgs describe the layer in REAL channels and fp32-sized byte extents; the cell kernels address the H1 input in 4-byte
// units (two halves each), so every quantity the loaders use is halved here, once: channels per pixel, K-steps, panel rows per tap,
// second-source split, byte extents.  The GEMM's columns (Cout, CoutP) are not touched: the O1 epilogue halves its own byte offsets.
static bool to_h1_units(ConvArgs& a) {
    if ((a.Cin & 63) || (a.in2 && (a.cin_split & 63)) || (a.nk & 1) || a.up || a.stem || !a.wh3) return false;
    a.Cin >>= 1; a.cin_split >>= 1; a.nk >>= 1; a.tap_rows >>= 1;
    a.log2cin4 = ilog2(a.Cin / 4);
    a.in_bytes >>= 1; a.in2_bytes >>= 1; a.w_bytes >>= 1; a.wh3_bytes >>= 1;
    if (a.out_fmt == 2) a.out_bytes >>= 1;
    if (a.res && a.res_fmt == 2) a.res_bytes >>= 1;
    return true;
}

hipError_t launch_conv(const ConvArgs& a_in, int tile_cfg, hipStream_t s) {
    ConvArgs a = a_in;
    if (a.in_fmt == 2 && !(a.out_mode == 0 && to_h1_units(a))) return hipErrorInvalidValue;
    if (a.in_fmt != 2 && (a.out_fmt == 2 || a.res_fmt == 2)) return hipErrorInvalidValue;
    if (a.Cin < 32 && !a.stem) {       // generic per-lane tap path (stem / small test shapes)
        if (tile_cfg == TILE_128x32) return launch_conv_t<128, 32, 4, 1, false>(a, s);
        if (tile_cfg == TILE_128x128_W8 || tile_cfg == TILE_128x128_LS || tile_cfg == TILE_128x128_S6 ||
            tile_cfg == TILE_128x128_S3 || tile_cfg == TILE_128x128_S6K16 || tile_cfg == TILE_128x128_S3K16 ||
            tile_cfg == TILE_128x128_S6K16W8 || tile_cfg == TILE_128x128_H3K16 || tile_cfg == TILE_128x128_H3K16W8 ||
            tile_cfg == TILE_128x128_H3K32) tile_cfg = TILE_128x128;
        if (a.CoutP % 128 == 0 && tile_cfg == TILE_128x128) return launch_conv_t<128, 128, 2, 2, false>(a, s);
        return launch_conv_t<128, 64, 2, 2, false>(a, s);
    }
    switch (tile_cfg) {
        case TILE_128x32: return launch_conv_t<128, 32, 4, 1, true>(a, s);
        case TILE_128x64: return launch_conv_t<128, 64, 2, 2, true>(a, s);
        case TILE_64x64:  return launch_conv_t<64, 64, 2, 2, true>(a, s);
        case TILE_128x128_W8: return launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x128_LS: return a.out_mode == 0 ? launch_conv_ls<128, 128>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x64_LS:  return a.out_mode == 0 ? launch_conv_ls<128, 64>(a, s) : launch_conv_t<128, 64, 2, 2, true>(a, s);
        case TILE_128x128_S6: return a.out_mode == 0 ? launch_conv_split<128, 128, 6, 32>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x128_S3: return a.out_mode == 0 ? launch_conv_split<128, 128, 3, 32>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x128_S6K16: return a.out_mode == 0 ? launch_conv_split<128, 128, 6, 16>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x128_S3K16: return a.out_mode == 0 ? launch_conv_split<128, 128, 3, 16>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x128_S6K16W8: return a.out_mode == 0 ? launch_conv_split<128, 128, 6, 16, 8>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x128_H3K16:   return a.out_mode == 0 ? launch_conv_split<128, 128, 2, 16>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x128_H3K16W8: return a.out_mode == 0 ? launch_conv_split<128, 128, 2, 16, 8>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x128_H3K32:
            if (conv_w64_eligible(a_in)) return launch_conv_w64(a, s);
            return a.out_mode == 0 ? launch_conv_split<128, 128, 2, 32>(a, s) : launch_conv_t<128, 128, 2, 4, true>(a, s);
        case TILE_128x64_H3:       return a.out_mode == 0 ? launch_conv_split<128, 64, 2, 32>(a, s) : launch_conv_t<128, 64, 2, 2, true>(a, s);
        case TILE_128x64_S6:  return a.out_mode == 0 ? launch_conv_split<128, 64, 6, 32>(a, s) : launch_conv_t<128, 64, 2, 2, true>(a, s);
        default:          return launch_conv_t<128, 128, 2, 2, true>(a, s);
    }
}

}  // namespace dgp
