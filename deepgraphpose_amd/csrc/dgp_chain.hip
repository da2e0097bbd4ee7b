// gfx950 (MI355X / CDNA4): the bottleneck "chain" kernel of the H2 inference engine.
//
// One launch computes, for a tile of output pixels, the END of bottleneck unit k and the BEGINNING of unit k + 1
// (PET/nnet/pose_net.py:46-52 -> slim resnet_v1 `bottleneck`; conv3 has no activation, the ReLU follows the add):
//
//     X'  = relu( bn3(R2 . W3) + shortcut )        shortcut = X (identity), X[::2, ::2] (subsample), or (X . Wsc) K-concatenated
//     R1' = relu( bn1'(X' . W1') )                 conv1 of the next unit (1x1, stride 1)
//
// X' is written to HBM once (the next unit's shortcut needs it) and is consumed by conv1 of the next unit STRAIGHT FROM THE
// ACCUMULATOR REGISTERS: the 4C-wide tensor is read from HBM once per unit (as the residual) instead of twice, and conv1 costs no
// launch of its own.  Layer by layer the stride-1 units of block1 moved 78.7 MB per frame, of which 19.7 MB is this second read.
//
// Why no LDS tile is needed between the two GEMMs.  v_mfma_f32_16x16x32_f16 computes D = A B with A[i = lane & 15][k-group lane >> 4],
// B[k-group lane >> 4][j = lane & 15] and D[i = 4 (lane >> 4) + r][j = lane & 15].  With the WEIGHTS as the A operand and the PIXELS
// as the B operand (the transposed product), a lane ends up with 4 consecutive output channels of ONE pixel per 16-channel block.
// The weight columns are permuted at pack time so that blocks (2q, 2q + 1) hold channels 32 q + 8 g + 4 b + r (g = lane >> 4,
// b = block parity): two blocks give the lane 8 consecutive channels of pixel lane & 15 -- exactly one H2 cell (hi 16 B | lo 16 B,
// written with two 16-byte stores) AND exactly the B-operand fragment (pixel lane & 15, k-group g) of the next GEMM's K-step q.
// So the epilogue of conv3 (BN affine, residual, ReLU, range tracking, fp16 high / low split) produces the operand of conv1 in place.
// Pointwise convs have no halo: a wave owns its 16 or 32 pixel rows through both GEMMs and never exchanges activations.
//
// What goes through LDS: only the weights.  Per 32 output channels `jp` of conv3 a chunk holds conv3's fragments for those columns
// (all K-steps), conv1's fragments for those 32 K rows (all columns) and the BN affine of those channels, in FRAGMENT ORDER (1 KiB =
// 64 lanes x 16 B per MFMA operand, so LDS-DMA copies linearly and ds_read_b128 is conflict-free).  Loader waves stream the chunks
// through a ring of NS slots (L2-resident after the first tile), one s_barrier per chunk; workgroups are persistent over tiles and
// the chunk sequence simply wraps around.
//
// Arithmetic is the engine's: fp16 high / low operand pairs, three MFMAs per product (h h + h l + l h), fp32 accumulation, the same
// epilogue formula as ls_epilogue_h2 -- so outputs agree with the layer-by-layer path to accumulation-order rounding.
#include "dgp_internal.h"
#include "dgp_device.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace dgp {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ floatx4 mma16(const uint4& a, const uint4& b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint4 ld16nt(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {      // streamed once: evict-first
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 2));
}
__device__ __forceinline__ void st16nt(__amdgpu_buffer_rsrc_t r, const uint4& v, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, 2);
}
__device__ __forceinline__ uint4 ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    // voffset carries the per-lane offset (OOB for rows past M: the hardware range check zero-fills / drops), soffset the
    // wave-uniform part -- the SGPR offset takes no part in the range check, so OOB stays OOB whatever is added here
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, const uint4& v, unsigned voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, soff, 0);
}

#ifdef DGP_DIAG
// diagnostic build only (scripts/diag_unit.sh): s_memtime stamps of compute wave 0 and loader wave 0 of every workgroup, summed over the grid.
// The stamps fence the schedule and drain the LDS queue: read SHARES, not totals.
//   [0] conv2 tap work  [1] conv2 tap barrier  [2] conv2 epilogue  [3] tile prologue  [4] conv3 MFMAs  [5] conv3 epilogue  [6] conv1 MFMAs
//   [7] chunk barrier   [8] conv1 epilogue     [9] tiles           [10] tile loop     [12] loader issue in conv2 steps, [11] in pointwise steps
//   [13] loader vmcnt wait [14] loader barrier [15] loader steps
__device__ unsigned long long g_unit_diag[16];
#define UD_STAMP(x)                                                                     \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#define UD_PARAM , unsigned long long (&ud)[12]
#define UD_ARG , ud
#else
#define UD_STAMP(x)
#define UD_PARAM
#define UD_ARG
#endif

// residual cells of chunk jp (32 channels of X) for the wave's RB row blocks -> res; chain_prefetch: the first PD - 1 chunks of a tile
template <int RES, int RB, bool H1 = false>
__device__ __forceinline__ void chain_fetch_res(const ChainArgs& p, uint4 (&res)[RB][2], const unsigned (&roff)[RB],
                                                const __amdgpu_buffer_rsrc_t rs_s2, int jp) {
#if defined(DGP_UX) && (DGP_UX & 1)      // timing-only ablation (scripts/diag_unit.sh): no residual loads
    for (int rb = 0; rb < RB; ++rb) { res[rb][0] = make_uint4(0, 0, 0, 0); res[rb][1] = make_uint4(0, 0, 0, 0); }
    return;
#endif
    if constexpr (RES != 0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if constexpr (H1) {                    // one 16-byte cell per 8 channels: chunk jp's 32 channels are 64 bytes of the pixel row
                res[rb][0] = (p.nt & 1) ? ld16nt(rs_s2, roff[rb], jp * 64) : ld16(rs_s2, roff[rb], jp * 64);
            } else if (p.nt & 1) {
                res[rb][0] = ld16nt(rs_s2, roff[rb], jp * 128);
                res[rb][1] = ld16nt(rs_s2, roff[rb], jp * 128 + 16);
            } else {
                res[rb][0] = ld16(rs_s2, roff[rb], jp * 128);
                res[rb][1] = ld16(rs_s2, roff[rb], jp * 128 + 16);
            }
        }
    }
}
template <int RES, int RB, int PD, bool H1 = false>
__device__ __forceinline__ void chain_prefetch(const ChainArgs& p, uint4 (&res)[PD][RB][2], const unsigned (&roff)[RB],
                                               const __amdgpu_buffer_rsrc_t rs_s2) {
#pragma unroll
    for (int u = 0; u < PD - 1; ++u) chain_fetch_res<RES, RB, H1>(p, res[u], roff, rs_s2, u);
}

// conv3 (+ shortcut, ReLU) -> X' -> conv1 -> R1' for the RB row blocks of one wave, one weight chunk per 32 channels of X'
// (the ring protocol is the caller's: `slot` is the chunk to read first, one barrier per chunk).  ph / pl: the B-operand fragments
// (pixel lane & 15, k-group lane >> 4) of conv3's K-steps -- R2 first, then the K-concatenated source.
// H1 (the 16-bit tier): every tensor is an H1 tensor (one 16-byte cell of 8 halves per 8 channels), the weight chunks are the SAME
// chunks -- only their high fragments are read -- and a product is one MFMA: pl is not used, X' / R1' leave as single cells.
template <int C, int C1, int CIN2, int RES, int RB, int NS, int PD, int CHUNK, int WR, bool H1 = false>
__device__ __forceinline__ void chain_tail(const ChainArgs& p, const uint4* ring, int& slot, const uint4 (&ph)[RB][(C + CIN2) / 32],
                                           const uint4 (&pl)[RB][(C + CIN2) / 32], const unsigned (&xoff)[RB], const unsigned (&roff)[RB],
                                           const unsigned (&r1off)[RB], const __amdgpu_buffer_rsrc_t rs_s2, const __amdgpu_buffer_rsrc_t rs_xo,
                                           const __amdgpu_buffer_rsrc_t rs_r1, float& amax_x, float& amax_r1, const int lane,
                                           uint4 (&res)[PD][RB][2] UD_PARAM) {
#ifdef DGP_DIAG
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
#endif
    constexpr int C4 = 4 * C, KS1 = (C + CIN2) / 32, NJP = C4 / 32, NCB = C1 / 16;
    constexpr int PL = H1 ? 1 : 2;                 // fragments per weight block: [hi][lo], or the high fragment alone (the 16-bit tier's chunks)
    constexpr int F1 = KS1 * 2 * PL, F2 = NCB * PL;
    static_assert(WR >= 2 && WR % 2 == 0, "weight fragments in flight: pairs share the ring");
    static_assert(NJP % PD == 0 && PD >= 2, "chunk loop is unrolled PD times");
    const int g = lane >> 4;
    floatx4 a2[RB][NCB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) a2[rb][cb] = floatx4{0.f, 0.f, 0.f, 0.f};
    // (the caller has issued the residual loads of chunks 0 .. PD - 2 into res[0 .. PD - 2]: chain_prefetch)
    auto fetch_res = [&](int jp, int buf) { chain_fetch_res<RES, RB, H1>(p, res[buf], roff, rs_s2, jp); };
    constexpr int CB = H1 ? 64 : 128;              // bytes of 32 channels of one pixel
    for (int jp0 = 0; jp0 < NJP; jp0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int jp = jp0 + u;
            if (jp + PD - 1 < NJP) fetch_res(jp + PD - 1, (u + PD - 1) % PD);
            UD_STAMP(c0);
            const uint4* Wc = ring + slot * (CHUNK / 16);          // the chunk; W: this lane's 16 bytes of fragment 0
            const uint4* W = Wc + lane;
            // ---- conv3 for channels [32 jp, 32 jp + 32): two 16-channel blocks, transposed product (weights = A operand)
            floatx4 a1[RB][2];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) { a1[rb][0] = floatx4{0.f, 0.f, 0.f, 0.f}; a1[rb][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }
            // The weight fragments of the step flow through a ring of WR registers, WR fragments ahead of the MFMAs that use them
            // (hipcc's own schedule was ds_read -> s_waitcnt 0 -> 2-3 MFMAs: one exposed LDS round trip per fragment pair).  The
            // fences keep the program order; the waits are then counted by the compiler.
            uint4 wq[WR];
#pragma unroll
            for (int f = 0; f < WR && f < F1 + F2; ++f) wq[f] = W[f * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int f0 = (s * 2 + b) * PL;
                    const uint4 wh = wq[f0 % WR], wl = wq[(f0 + PL - 1) % WR];
                    if constexpr (!H1) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a1[rb][b] = mma16(wl, ph[rb][s], a1[rb][b]);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a1[rb][b] = mma16(wh, pl[rb][s], a1[rb][b]);
                    }
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a1[rb][b] = mma16(wh, ph[rb][s], a1[rb][b]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (f0 + WR < F1 + F2) wq[f0 % WR] = W[(f0 + WR) * 64];
                    if (PL == 2 && f0 + 1 + WR < F1 + F2) wq[(f0 + 1) % WR] = W[(f0 + 1 + WR) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
            // ---- epilogue of conv3 = operand of conv1: BN affine (x the power of two that undoes the operand scales), shortcut,
            // ReLU, range, split
            UD_STAMP(c1);
            const float4 sa = __builtin_bit_cast(float4, Wc[(F1 + F2) * 64 + 2 * g]), sb = __builtin_bit_cast(float4, Wc[(F1 + F2) * 64 + 2 * g + 1]);
            const float4 ba = __builtin_bit_cast(float4, Wc[(F1 + F2) * 64 + 8 + 2 * g]), bb = __builtin_bit_cast(float4, Wc[(F1 + F2) * 64 + 8 + 2 * g + 1]);
            const float sc[8] = {sa.x * p.post1, sa.y * p.post1, sa.z * p.post1, sa.w * p.post1,
                                 sb.x * p.post1, sb.y * p.post1, sb.z * p.post1, sb.w * p.post1};
            const float bi[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
            uint4 xh[RB], xl[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float o[8], r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (RES != 0) {
                    if constexpr (H1) h1_unpack8(res[u][rb][0], p.res_inv_scale, r);
                    else h2_unpack8(res[u][rb][0], res[u][rb][1], p.res_inv_scale, r);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    o[k] = a1[rb][k >> 2][k & 3] * sc[k] + bi[k] + r[k];
                    o[k] = fmaxf(o[k], 0.f);
                }
                if constexpr (H1) { xh[rb] = h1_pack8(o, p.xout_scale); xl[rb] = xh[rb]; }
                else h2_pack8(o, p.xout_scale, xh[rb], xl[rb]);
#if defined(DGP_UX) && (DGP_UX & 2)      // timing-only ablation: no X' stores
                if (p.nt == 77) {
#else
                if (p.nt & 2) {
#endif
                    st16nt(rs_xo, xh[rb], xoff[rb], jp * CB);
                    if constexpr (!H1) st16nt(rs_xo, xl[rb], xoff[rb], jp * CB + 16);
                } else {
#if !(defined(DGP_UX) && (DGP_UX & 2))
                    st16(rs_xo, xh[rb], xoff[rb], jp * CB);
                    if constexpr (!H1) st16(rs_xo, xl[rb], xoff[rb], jp * CB + 16);
#endif
                }
                if (xoff[rb] != OOB) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) amax_x = fmaxf(amax_x, o[k]);
                }
            }
            // ---- conv1 of the next unit: K-step jp, all C1 columns
            UD_STAMP(c2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int f0 = F1 + cb * PL;
                const uint4 wh = wq[f0 % WR], wl = wq[(f0 + PL - 1) % WR];
                if constexpr (!H1) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a2[rb][cb] = mma16(wl, xh[rb], a2[rb][cb]);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a2[rb][cb] = mma16(wh, xl[rb], a2[rb][cb]);
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a2[rb][cb] = mma16(wh, xh[rb], a2[rb][cb]);
                __builtin_amdgcn_sched_barrier(0);
                if (f0 + WR < F1 + F2) wq[f0 % WR] = W[(f0 + WR) * 64];
                if (PL == 2 && f0 + 1 + WR < F1 + F2) wq[(f0 + 1) % WR] = W[(f0 + 1 + WR) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = slot + 1 == NS ? 0 : slot + 1;
            UD_STAMP(c3);
            // (not __syncthreads(): its fences would drain the residual prefetch and the stores; the ring only needs this
            // wave's LDS reads of the slot to have completed)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                          // B(it): this slot may be refilled
            UD_STAMP(c4);
#ifdef DGP_DIAG
            ud[4] += c1 - c0; ud[5] += c2 - c1; ud[6] += c3 - c2; ud[7] += c4 - c3;
#endif
        }
    }
    UD_STAMP(c0);
    // ---- epilogue of conv1: BN affine, ReLU, range, split, store R1' (block pair q = 32 channels, 8 per lane)
#pragma unroll
    for (int q = 0; q < NCB / 2; ++q) {
        const float4 s0 = *reinterpret_cast<const float4*>(p.sc1 + 32 * q + 8 * g), s1 = *reinterpret_cast<const float4*>(p.sc1 + 32 * q + 8 * g + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(p.bi1 + 32 * q + 8 * g), b1 = *reinterpret_cast<const float4*>(p.bi1 + 32 * q + 8 * g + 4);
        const float sc[8] = {s0.x * p.post2, s0.y * p.post2, s0.z * p.post2, s0.w * p.post2,
                             s1.x * p.post2, s1.y * p.post2, s1.z * p.post2, s1.w * p.post2};
        const float bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fmaxf(a2[rb][2 * q + (k >> 2)][k & 3] * sc[k] + bi[k], 0.f);
            if constexpr (H1) {
                st16(rs_r1, h1_pack8(o, p.r1_scale), r1off[rb], q * CB);
            } else {
            uint4 hi, lo;
            h2_pack8(o, p.r1_scale, hi, lo);
            st16(rs_r1, hi, r1off[rb], q * 128);
            st16(rs_r1, lo, r1off[rb], q * 128 + 16);
            }
            if (r1off[rb] != OOB) {
#pragma unroll
                for (int k = 0; k < 8; ++k) amax_r1 = fmaxf(amax_r1, o[k]);
            }
        }
    }
    UD_STAMP(c1);
#ifdef DGP_DIAG
    ud[8] += c1 - c0;
#endif
}

// C     channels of R2 (conv3's K), C4 = 4 C channels of X'
// C1    output channels of the next unit's conv1 (C inside a block, 2 C across a block boundary)
// CIN2  channels of the K-concatenated shortcut source (0: none)
// RES   0 none (CIN2 > 0), 1 residual on the same pixel grid, 2 residual[n, 2 ho, 2 wo] (subsample of a stride-2 unit)
// RB    16-row blocks per wave (1 or 2); NCW compute waves, NLW loader waves, NS ring slots, PD residual buffers (PD - 1 chunks ahead),
//       WR weight fragments in flight per wave
template <int C, int C1, int CIN2, int RES, int RB, int NCW, int NLW, int NS, int PD, int WR, bool H1 = false>
__global__ __launch_bounds__(64 * (NCW + NLW)) void chain_kernel(const ChainArgs p) {
    constexpr int C4 = 4 * C;
    constexpr int EB = H1 ? 2 : 4;                                     // bytes per channel of the activation tensors
    constexpr int KSA = C / 32, KSB = CIN2 / 32, KS1 = KSA + KSB;      // K-steps of conv3: R2, then the second source
    constexpr int NJP = C4 / 32;                                       // chunks per tile
    constexpr int NCB = C1 / 16;                                       // 16-channel blocks of conv1's output
    constexpr int PL = H1 ? 1 : 2;                                     // planes per weight block in the chunks (H1: the high fragments alone)
    constexpr int F1 = KS1 * 2 * PL, F2 = NCB * PL, NF = F1 + F2 + 1;  // 1-KiB fragments per chunk: conv3 [s][b][plane], conv1 [cb][plane], affine
    constexpr int CHUNK = NF * 1024;
    constexpr int TILE = 16 * RB * NCW;
    static_assert(NJP % PD == 0 && PD >= 2, "chunk loop is unrolled PD times");
    static_assert(RES == 0 || CIN2 == 0, "residual or K-concatenated shortcut, not both");
    static_assert(NS == 2 || NS == 3, "ring protocols of the loader below");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int my_tiles = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * NJP;                                  // chunks this workgroup consumes

    if (wave >= NCW) {
        // ================================ loader waves: weight chunks -> LDS ring ================================
        const int lw = wave - NCW;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, (int)p.w_bytes, 0x00020000);
        int jp = 0, slot = 0;
        // wave lw copies fragments lw, lw + NLW, ...: NIW instructions per chunk (a compile-time count: the waits below are counted)
        auto run = [&](auto niw) {
            constexpr int NIW = decltype(niw)::value;
            auto issue = [&]() {
                char* dst = smem + slot * CHUNK;
                const int gbase = jp * CHUNK;
#pragma unroll
                for (int i = 0; i < NIW; ++i) {
                    const int f = lw + i * NLW;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(dst + f * 1024), 16, lane * 16, gbase + f * 1024, 0, 0);
                }
                jp = jp + 1 == NJP ? 0 : jp + 1;
                slot = slot + 1 == NS ? 0 : slot + 1;
            };
            if constexpr (NS == 3) {
                // chunk it + 2 is issued after barrier it - 1 (every wave has left slot (it - 1) % 3); chunk it + 1 has landed before barrier it
                if (total > 0) issue();
                if (total > 1) { issue(); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                          // B(-1): chunk 0 has landed
                for (int it = 0; it < total; ++it) {
                    if (it + 2 < total) { issue(); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory"); }
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                      // B(it)
                }
            } else {
                // two slots (large chunks): chunk it + 1 is issued after barrier it - 1 and must land during step it
                if (total > 0) issue();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                          // B(-1)
                for (int it = 0; it < total; ++it) {
                    if (it + 1 < total) issue();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                      // B(it)
                }
            }
        };
        constexpr int NI_HI = (NF + NLW - 1) / NLW, NI_LO = NF / NLW;
        if (lw < NF % NLW || NI_HI == NI_LO) run(std::integral_constant<int, NI_HI>());
        else run(std::integral_constant<int, NI_LO>());
        return;
    }

    // ================================== compute waves ==================================
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.r2), 0, (int)p.r2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_s2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src2 ? p.src2 : p.r2), 0,
                                                                            p.src2 ? (int)p.src2_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xo = __builtin_amdgcn_make_buffer_rsrc(p.xout, 0, (int)p.xout_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc(p.r1out, 0, (int)p.r1_bytes, 0x00020000);
    const uint4* ring = reinterpret_cast<const uint4*>(smem);
    int slot = 0;
    float amax_x = 0.f, amax_r1 = 0.f;
    __builtin_amdgcn_s_barrier();                                      // B(-1)
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        unsigned xoff[RB], roff[RB], r1off[RB];
        uint4 ph[RB][KS1], pl[RB][KS1];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int m = t * TILE + (wave * RB + rb) * 16 + l15;
            const bool ok = m < p.M;
            xoff[rb] = ok ? (unsigned)m * (unsigned)(C4 * EB) + (unsigned)(g * 8 * EB) : OOB;
            r1off[rb] = ok ? (unsigned)m * (unsigned)(C1 * EB) + (unsigned)(g * 8 * EB) : OOB;
            if (RES == 2) {
                const int n = m / p.HoWo, rem = m - n * p.HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                roff[rb] = ok ? (unsigned)((n * p.res_H + 2 * ho) * p.res_W + 2 * wo) * (unsigned)(C4 * EB) + (unsigned)(g * 8 * EB) : OOB;
            } else roff[rb] = xoff[rb];
            const unsigned aoff = ok ? (unsigned)m * (unsigned)(C * EB) + (unsigned)(g * 8 * EB) : OOB;
#pragma unroll
            for (int s = 0; s < KSA; ++s) {
                ph[rb][s] = ld16(rs_r2, aoff, s * 32 * EB);
                if constexpr (H1) pl[rb][s] = ph[rb][s]; else pl[rb][s] = ld16(rs_r2, aoff, s * 128 + 16);
            }
            if constexpr (KSB > 0) {
                const unsigned boff = ok ? (unsigned)m * (unsigned)(CIN2 * EB) + (unsigned)(g * 8 * EB) : OOB;
#pragma unroll
                for (int s = 0; s < KSB; ++s) {
                    ph[rb][KSA + s] = ld16(rs_s2, boff, s * 32 * EB);
                    if constexpr (H1) pl[rb][KSA + s] = ph[rb][KSA + s]; else pl[rb][KSA + s] = ld16(rs_s2, boff, s * 128 + 16);
                }
            }
        }
        uint4 res[PD][RB][2];
        chain_prefetch<RES, RB, PD, H1>(p, res, roff, rs_s2);
#ifdef DGP_DIAG
        unsigned long long ud[12] = {0};
#endif
        chain_tail<C, C1, CIN2, RES, RB, NS, PD, CHUNK, WR, H1>(p, ring, slot, ph, pl, xoff, roff, r1off, rs_s2, rs_xo, rs_r1, amax_x, amax_r1, lane, res UD_ARG);
#ifdef DGP_DIAG
        if (wave == 0 && lane == 0) {
            for (int i = 4; i < 9; ++i) atomicAdd(&g_unit_diag[i], ud[i]);
            atomicAdd(&g_unit_diag[9], 1ull);
        }
#endif
    }
    // both tensors are post-ReLU: max = max |.|
    if (p.xout_absmax) track_absmax(p.xout_absmax, amax_x, lane, (int)(blockIdx.x * 8u + (unsigned)wave));
    if (p.r1_absmax) track_absmax(p.r1_absmax, amax_r1, lane, (int)(blockIdx.x * 8u + (unsigned)wave) + 97);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Unit kernel (block1, C = 64): conv2 of unit k (3x3, stride 1, SAME) IN FRONT of the chain, so that one launch runs
//     R1 --conv2 + BN + ReLU--> R2 --conv3 + BN (+ shortcut) + ReLU--> X' --conv1' + BN + ReLU--> R1'
// with R2 never leaving the registers: conv2's accumulators come out of the transposed product in the same lane layout as conv3's,
// and two 16-channel blocks are again one B-operand fragment of the next GEMM.  Layer by layer a stride-1 unit of block1 moved
// 78.7 MB per frame; here it reads R1 (4.9, + halo) and X (19.7) and writes X' (19.7) and R1' (4.9).
//
// conv2 needs neighbours: a workgroup owns a TH x 16 pixel tile of ONE frame (wave w = row w, lane & 15 = column) and its
// (TH + 2) x 18 halo tile of R1 lives in LDS -- filled by the loader wave with LDS-DMA (out-of-image pixels: the hardware range check
// writes the zeros of SAME padding) while the previous tile runs its pointwise stage, which no longer reads the halo buffer.  Pixel
// hp of the halo tile is 256 B = 16 slots of 16 B (cell c: slot 2c high, 2c + 1 low) = one row of LDS banks, so slot s is stored
// at s ^ (hp & 15): the 16 lanes of a ds_read_b128 service group (16 different columns) then touch (nearly) all 16 bank groups.
// The swizzle is applied on the SOURCE side (the DMA destination is lane-linear): lane L of instruction i fills slot L & 15 of
// pixel 4 i + (L >> 4) with global slot (L & 15) ^ (hp & 15).
// Weights: 9 chunks for conv2 (one tap each: [ks][cb][plane] fragments + conv2's BN affine) followed by the chain's chunks, through
// the same ring; 9 + 8 barriers per tile.
// Tiles are dealt so that workgroups on one XCD (blockIdx & 7) work on neighbouring tiles: halo rows are re-read from that L2.
// H1 (the 16-bit tier): all tensors are H1, a halo pixel is 128 B = 8 slots (cell c in slot c), two pixels per row of LDS banks, so slot s
// of pixel hp is stored at s ^ ((hp >> 1) & 7): 16 consecutive pixels of one cell then cover the 16 bank groups once.  One DMA instruction
// fills 8 pixels (lane L: slot L & 7 of pixel 8 i + (L >> 3)).  Weight chunks as in the parity tier, high fragments only.
template <int C, int C1, int CIN2, int RES, int NCW, int NLW, int PD, int WR, int HWV = 1, bool H1 = false>
__global__ __launch_bounds__(64 * (NCW + NLW + HWV)) void unit_kernel(const ChainArgs p) {
    // NLW weight-loader waves + HWV (0 / 1) halo wave.  Round 4: with the halo pieces in the weight loaders' queues (HWV = 0) the weight
    // stream waits behind them -- a wave's vector-memory operations return in order, the halo pieces come from HBM / the Infinity Cache,
    // the weight chunks from L2: a timing-only build without the halo DMA ran unit_c64_n64_sc 21 % faster, unit_c64_n64_id 7 %
    // (scripts/ablate_unit.sh).  The halo wave gets a part of that back where the launch is not HBM-bound (launch_unit).
    constexpr int NS = 3;
    constexpr int TH = NCW, TW = 16, HW = TW + 2, HPIX = (TH + 2) * HW;
    constexpr int EB = H1 ? 2 : 4;
    constexpr int PIXB = C * EB;
    constexpr int PPI = 1024 / PIXB;                                // halo pixels per DMA instruction (4; H1: 8)
    static_assert(C == 64, "one pixel of R1 = 16 (H1: 8) slots of 16 bytes (the bank swizzle)");
    constexpr int NHI = (HPIX * PIXB + 1023) / 1024;               // DMA instructions per halo tile (4 pixels each; the last may run past
    constexpr int HALO_BYTES = NHI * 1024;                         // the tile: those lanes are out of range and write zeros into the pad)
    static_assert(NLW == 1 || NLW == 2 || NLW == 4, "one, two or four weight-loader waves");
    static_assert(HWV == 0 || HWV == 1, "at most one halo wave");
    constexpr int NHW = HWV ? 1 : NLW;                              // waves that share a halo tile's instructions
    constexpr int C4 = 4 * C, KS1 = (C + CIN2) / 32, NJP = C4 / 32, NCB = C1 / 16;
    constexpr int PL = H1 ? 1 : 2;                                  // planes per weight block in the chunks (H1: the high fragments alone)
    constexpr int NFJ = (KS1 * 2 + NCB) * PL + 1;                   // fragments of a chain chunk
    constexpr int KS2 = C / 32, NCB2 = C / 16, NF2 = KS2 * NCB2 * PL + 1;     // conv2: fragments per tap + the affine fragment
    constexpr int NFMAX = NFJ > NF2 ? NFJ : NF2, CHUNK = NFMAX * 1024;
    constexpr int STEPS = 9 + NJP;
    // halo instructions issued per pointwise step; the halo wave is done one step before the tile ends, so that its wait at the tile's
    // last barrier finds the pieces landed instead of exposing their latency to every wave
    constexpr int HSTEPS = HWV ? NJP - 1 : NJP;
    constexpr int HPP = (NHI + HSTEPS - 1) / HSTEPS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;
    char* ringb = smem + HALO_BYTES;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int vwg = (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;           // workgroups of one XCD take consecutive tiles
    const int my_tiles = vwg < p.ntiles ? (p.ntiles - vwg + G - 1) / G : 0;
    const int total = my_tiles * STEPS;
    const int tpf = p.TY * p.TX;

    if (wave >= NCW) {
        // ================================ loader waves: weight chunks (+ halo tiles) ================================
        // Weight wave lw copies the fragments i with i % NLW == lw; lw is a compile-time constant inside (which fragments a wave copies
        // and how many is then straight-line code: with a run-time lw every LDS-DMA instruction sat behind its own conditional branch,
        // ~200 cycles per instruction instead of ~70).
        // halo instructions [i0, i1) of a tile, every NHW-th from hl on
        auto halo_issue = [&](int tile, int i0, int i1, int hl) {
            const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.r1in), 0, (int)p.r1in_bytes, 0x00020000);
            const int f = tile / tpf, rem = tile - f * tpf, ty = rem / p.TX, tx = rem - ty * p.TX;
            for (int i = i0 + hl; i < i1; i += NHW) {
                const int hp = PPI * i + (H1 ? (lane >> 3) : (lane >> 4));
                const int hy = (hp * 57) >> 10, hx = hp - HW * hy;                 // hp / 18 (exact for hp < 400)
                const int gy = ty * TH - 1 + hy, gx = tx * TW - 1 + hx;
                const bool ok = hp < HPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                const unsigned cell = H1 ? (unsigned)((lane & 7) ^ ((hp >> 1) & 7)) : (unsigned)((lane & 15) ^ (hp & 15));
                const unsigned off = ok ? (unsigned)((f * p.H + gy) * p.W + gx) * (unsigned)PIXB + (cell << 4) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(halo + i * 1024), 16, (int)off, 0, 0, 0);
            }
        };
        // the pointwise stage of a tile no longer reads the halo buffer (every wave passed barrier 8): the next tile's is fetched then
        auto halo_step = [&](int cur_st, int tile, int hl) {
#if !(defined(DGP_UX) && (DGP_UX & 32))     // (timing-only ablation: no halo tiles after the first)
            if (cur_st >= 9 && tile + G < p.ntiles) {
                const int i0 = (cur_st - 9) * HPP, i1 = i0 + HPP < NHI ? i0 + HPP : NHI;
                halo_issue(tile + G, i0, i1, hl);
            }
#endif
        };
        const int lw_rt = wave - NCW;
        if (HWV && lw_rt == NLW) {
            // ---- the halo wave: nothing but halo pieces in its queue; they must have landed at a tile's last barrier
            if (total > 0) halo_issue(vwg, 0, NHI, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // B(-1): the first halo tile has landed
            int cur_st = 0, tile = vwg;
            for (int it = 0; it < total; ++it) {
                halo_step(cur_st, tile, 0);
                if (cur_st == STEPS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                          // B(it)
                if (++cur_st == STEPS) { cur_st = 0; tile += G; }
            }
            return;
        }
        auto loader = [&](auto lw_c) {
        constexpr int lw = decltype(lw_c)::value;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wfrag), 0, (int)p.w_bytes, 0x00020000);
        int st = 0, slot = 0;                                          // step and ring slot of the NEXT chunk to issue
        // instructions of THIS wave per chunk: compile-time counts for the counted waits
        // (wave lw issues ceil or floor of the chunk's fragments / NLW: the first `fragments % NLW` waves one more)
        constexpr int N2 = lw < NF2 % NLW ? (NF2 + NLW - 1) / NLW : NF2 / NLW, NJ = lw < NFJ % NLW ? (NFJ + NLW - 1) / NLW : NFJ / NLW;
        auto issue = [&]() {
            char* dst = ringb + slot * CHUNK;
#if defined(DGP_UX) && (DGP_UX & 4)      // timing-only ablation: every workgroup walks the chunks in its own rotation (results are garbage)
            const int st2 = (st + (int)blockIdx.x) % 9, stj = (st - 9 + (int)blockIdx.x) % NJP;
#else
            const int st2 = st, stj = st - 9;
#endif
            if (st < 9) {
#pragma unroll
                for (int k = 0; k < N2; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(dst + (lw + k * NLW) * 1024), 16, lane * 16, st2 * (NF2 * 1024) + (lw + k * NLW) * 1024, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < NJ; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(dst + (lw + k * NLW) * 1024), 16, lane * 16,
                                                             9 * (NF2 * 1024) + stj * (NFJ * 1024) + (lw + k * NLW) * 1024, 0, 0);
            }
            const bool was_conv2 = st < 9;
            st = st + 1 == STEPS ? 0 : st + 1;
            slot = slot + 1 == NS ? 0 : slot + 1;
            return was_conv2;
        };
        auto wait_all_but_newest = [&](bool conv2_chunk) {             // everything older than the chunk just issued has landed
            if (conv2_chunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ) : "memory");
        };
        if (total > 0) {
            if (!HWV) halo_issue(vwg, 0, NHI, lw);
            issue();
            wait_all_but_newest(issue());                              // (STEPS >= 2: both are conv2 chunks)
        }
        __builtin_amdgcn_s_barrier();                                  // B(-1): (the first halo tile and) chunk 0 have landed
        int cur_st = 0, tile = vwg;
#ifdef DGP_DIAG
        unsigned long long l0 = 0, l1 = 0, l2 = 0, l3 = 0, ul[3] = {0, 0, 0}, ul3 = 0;
#endif
        for (int it = 0; it < total; ++it) {
            UD_STAMP(l0);
            if (!HWV) halo_step(cur_st, tile, lw);                     // (older than the chunk issued below: landed when its wait returns)
#ifdef DGP_DIAG
            bool c2_ = false;
            if (it + 2 < total) c2_ = issue();
            UD_STAMP(l1);
            if (it + 2 < total) wait_all_but_newest(c2_);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            UD_STAMP(l2);
#else
            if (it + 2 < total) wait_all_but_newest(issue());
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __builtin_amdgcn_s_barrier();                              // B(it)
            UD_STAMP(l3);
#ifdef DGP_DIAG
            if (cur_st < 9) ul[0] += l1 - l0; else ul3 += l1 - l0;
            ul[1] += l2 - l1; ul[2] += l3 - l2;
#endif
            if (++cur_st == STEPS) { cur_st = 0; tile += G; }
        }
#ifdef DGP_DIAG
        if (lw == 0 && lane == 0) {
            atomicAdd(&g_unit_diag[12], ul[0]); atomicAdd(&g_unit_diag[13], ul[1]); atomicAdd(&g_unit_diag[14], ul[2]); atomicAdd(&g_unit_diag[11], ul3);
            atomicAdd(&g_unit_diag[15], (unsigned long long)total);
        }
#endif
        };
        if (lw_rt == 0) loader(std::integral_constant<int, 0>());
        else if (NLW > 1 && lw_rt == 1) loader(std::integral_constant<int, (NLW > 1 ? 1 : 0)>());
        else if (NLW > 2 && lw_rt == 2) loader(std::integral_constant<int, (NLW > 2 ? 2 : 0)>());
        else if (NLW > 3) loader(std::integral_constant<int, (NLW > 3 ? 3 : 0)>());
        return;
    }

    // ================================== compute waves ==================================
    const int l15 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_s2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src2), 0, (int)p.src2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xo = __builtin_amdgcn_make_buffer_rsrc(p.xout, 0, (int)p.xout_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r1 = __builtin_amdgcn_make_buffer_rsrc(p.r1out, 0, (int)p.r1_bytes, 0x00020000);
    const uint4* ring = reinterpret_cast<const uint4*>(ringb);
    int slot = 0;
    float amax_x = 0.f, amax_r1 = 0.f, amax_r2 = 0.f;
    __builtin_amdgcn_s_barrier();                                      // B(-1)
#ifdef DGP_DIAG
    unsigned long long ud[12] = {0}, u0 = 0, u1 = 0, u2 = 0, ubeg = 0, uend = 0;
    UD_STAMP(ubeg);
#endif
    for (int tile = vwg; tile < p.ntiles; tile += G) {
        UD_STAMP(u0);
        const int f = tile / tpf, rem = tile - f * tpf, ty = rem / p.TX, tx = rem - ty * p.TX;
        const int y = ty * TH + wave, x = tx * TW + l15;
        const bool ok = y < p.H && x < p.W;
        const int m = (f * p.H + y) * p.W + x;
        unsigned xoff[1], roff[1], r1off[1];
        xoff[0] = ok ? (unsigned)m * (unsigned)(C4 * EB) + (unsigned)(g * 8 * EB) : OOB;
        roff[0] = xoff[0];
        r1off[0] = ok ? (unsigned)m * (unsigned)(C1 * EB) + (unsigned)(g * 8 * EB) : OOB;
        uint4 ph[1][KS1], pl[1][KS1];
        if constexpr (CIN2 > 0) {                                      // the K-concatenated source at this wave's pixels (used after conv2)
            const unsigned boff = ok ? (unsigned)m * (unsigned)(CIN2 * EB) + (unsigned)(g * 8 * EB) : OOB;
#pragma unroll
            for (int s = 0; s < CIN2 / 32; ++s) {
                ph[0][KS2 + s] = ld16(rs_s2, boff, s * 32 * EB);
                if constexpr (H1) pl[0][KS2 + s] = ph[0][KS2 + s]; else pl[0][KS2 + s] = ld16(rs_s2, boff, s * 128 + 16);
            }
        }
        // the residual of the first pointwise chunks is requested NOW: conv2 below touches no global memory, so HBM would sit idle
        // for nine steps and the pointwise stage would start by waiting for it
        uint4 res[PD][1][2];
        chain_prefetch<RES, 1, PD, H1>(p, res, roff, rs_s2);
        // ---- conv2: 9 taps x KS2 K-steps, the pixel fragment comes from the halo tile (shifted by the tap), weights from the ring
        floatx4 acc[NCB2];
#pragma unroll
        for (int cb = 0; cb < NCB2; ++cb) acc[cb] = floatx4{0.f, 0.f, 0.f, 0.f};
        uint4 aff[KS2][4];
        UD_STAMP(u1);
#ifdef DGP_DIAG
        ud[3] += u1 - u0;
#endif
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            UD_STAMP(u0);
            const uint4* Wc = ring + slot * (CHUNK / 16);
            const uint4* W = Wc + lane;
            const int hp = (wave + t / 3) * HW + l15 + t % 3;
            const char* hrow = halo + hp * PIXB;
            const int sw = H1 ? ((hp >> 1) & 7) : (hp & 15);
            // pixel fragments of the tap (all K-steps) first, then the weight fragments through the ring (see chain_tail)
            uint4 ah[KS2], al[KS2];
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                if constexpr (H1) {
                    ah[ks] = *reinterpret_cast<const uint4*>(hrow + (((4 * ks + g) ^ sw) << 4));
                    al[ks] = ah[ks];
                } else {
                const int sl = 2 * (4 * ks + g);
                ah[ks] = *reinterpret_cast<const uint4*>(hrow + ((sl ^ sw) << 4));
                al[ks] = *reinterpret_cast<const uint4*>(hrow + (((sl + 1) ^ sw) << 4));
                }
            }
            constexpr int NW2 = KS2 * NCB2 * PL;
            uint4 wq[WR];
#if defined(DGP_UX) && (DGP_UX & 8)      // timing-only ablation: the weight fragments are not read from LDS
#define UX_W(i) make_uint4(lane, i, slot, t)
#else
#define UX_W(i) W[(i) * 64]
#endif
#pragma unroll
            for (int f = 0; f < WR && f < NW2; ++f) wq[f] = UX_W(f);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
                for (int cb = 0; cb < NCB2; ++cb) {
                    const int f0 = (ks * NCB2 + cb) * PL;
                    const uint4 wh = wq[f0 % WR], wl = wq[(f0 + PL - 1) % WR];
#if defined(DGP_UX) && (DGP_UX & 16)     // timing-only ablation: no MFMAs
                    acc[cb][0] += __builtin_bit_cast(float, wh.x ^ wl.y ^ ah[ks].x ^ al[ks].y);
#else
                    if constexpr (!H1) {
                    acc[cb] = mma16(wl, ah[ks], acc[cb]);
                    acc[cb] = mma16(wh, al[ks], acc[cb]);
                    }
                    acc[cb] = mma16(wh, ah[ks], acc[cb]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    if (f0 + WR < NW2) wq[f0 % WR] = UX_W(f0 + WR);
                    if (PL == 2 && f0 + 1 + WR < NW2) wq[(f0 + 1) % WR] = UX_W(f0 + 1 + WR);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef UX_W
            if (t == 8) {                                              // conv2's BN affine rides in the last tap's chunk
#pragma unroll
                for (int q = 0; q < KS2; ++q) {
                    aff[q][0] = Wc[(NF2 - 1) * 64 + 8 * q + 2 * g]; aff[q][1] = Wc[(NF2 - 1) * 64 + 8 * q + 2 * g + 1];
                    aff[q][2] = Wc[(NF2 - 1) * 64 + 16 + 8 * q + 2 * g]; aff[q][3] = Wc[(NF2 - 1) * 64 + 16 + 8 * q + 2 * g + 1];
                }
            }
            slot = slot + 1 == NS ? 0 : slot + 1;
            UD_STAMP(u1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            UD_STAMP(u2);
#ifdef DGP_DIAG
            ud[0] += u1 - u0; ud[1] += u2 - u1;
#endif
        }
        UD_STAMP(u0);
        // ---- epilogue of conv2 = operand of conv3: BN affine, ReLU, range, split with R2's scale
#pragma unroll
        for (int q = 0; q < KS2; ++q) {
            const float4 sa = __builtin_bit_cast(float4, aff[q][0]), sb = __builtin_bit_cast(float4, aff[q][1]);
            const float4 ba = __builtin_bit_cast(float4, aff[q][2]), bb = __builtin_bit_cast(float4, aff[q][3]);
            const float sc[8] = {sa.x * p.post0, sa.y * p.post0, sa.z * p.post0, sa.w * p.post0, sb.x * p.post0, sb.y * p.post0, sb.z * p.post0, sb.w * p.post0};
            const float bi[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fmaxf(acc[2 * q + (k >> 2)][k & 3] * sc[k] + bi[k], 0.f);
            if constexpr (H1) { ph[0][q] = h1_pack8(o, p.r2_scale); pl[0][q] = ph[0][q]; }
            else h2_pack8(o, p.r2_scale, ph[0][q], pl[0][q]);
            if (ok) {
#pragma unroll
                for (int k = 0; k < 8; ++k) amax_r2 = fmaxf(amax_r2, o[k]);
            }
        }
        UD_STAMP(u1);
#ifdef DGP_DIAG
        ud[2] += u1 - u0; ud[9] += 1;
#endif
        chain_tail<C, C1, CIN2, RES, 1, NS, PD, CHUNK, WR, H1>(p, ring, slot, ph, pl, xoff, roff, r1off, rs_s2, rs_xo, rs_r1, amax_x, amax_r1, lane, res UD_ARG);
    }
#ifdef DGP_DIAG
    UD_STAMP(uend);
    ud[10] = uend - ubeg;
    if (wave == 0 && lane == 0)
        for (int i = 0; i < 11; ++i) atomicAdd(&g_unit_diag[i], ud[i]);
#endif
    if (p.xout_absmax) track_absmax(p.xout_absmax, amax_x, lane, (int)(blockIdx.x * 8u + (unsigned)wave));
    if (p.r1_absmax) track_absmax(p.r1_absmax, amax_r1, lane, (int)(blockIdx.x * 8u + (unsigned)wave) + 97);
    if (p.r2_absmax) track_absmax(p.r2_absmax, amax_r2, lane, (int)(blockIdx.x * 8u + (unsigned)wave) + 41);
}

// fp32 fragments -> fp16 high / low fragment pairs in chunk order.  src per chunk: pairs x [64 lanes][8 floats] then the affine
// fragment [256 floats]; out per chunk: 2 pairs + 1 fragments of 1 KiB (pairs as [hi][lo], affine copied).
// s1 / s2: the powers of two conv3's / conv1's weights are stored with.
__global__ __launch_bounds__(256) void chain_pack_kernel(const float4* __restrict__ src, int n_chunks, int f1_pairs, int f2_pairs,
                                                         float s1, float s2, uint4* __restrict__ out, int planes) {
    const int pairs = f1_pairs + f2_pairs;
    const int nf = planes * pairs + 1;
    const long long src_chunk = (long long)pairs * 128 + 64;      // float4s
    const long long total = (long long)n_chunks * (pairs + 1) * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ln = (int)(i & 63);
        const long long fp = i >> 6;
        const int ch = (int)(fp / (pairs + 1)), pr = (int)(fp % (pairs + 1));
        uint4* o = out + ((long long)ch * nf + planes * pr) * 64 + ln;
        if (pr == pairs) {
            o[0] = __builtin_bit_cast(uint4, src[ch * src_chunk + (long long)pairs * 128 + ln]);
            continue;
        }
        const float4* sp = src + ch * src_chunk + (long long)pr * 128 + ln * 2;
        uint2 h0, l0, h1, l1;
        const float sc = pr < f1_pairs ? s1 : s2;
        split2_f16(sp[0], sc, h0, l0);
        split2_f16(sp[1], sc, h1, l1);
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        if (planes == 2) o[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

hipError_t launch_chain_pack(const float* src, int n_chunks, int f1_pairs, int f2_pairs, float s1, float s2, void* out, hipStream_t s, int planes) {
    const long long total = (long long)n_chunks * (f1_pairs + f2_pairs + 1) * 64;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(chain_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(src), n_chunks, f1_pairs,
                       f2_pairs, s1, s2, reinterpret_cast<uint4*>(out), planes);
    return hipGetLastError();
}

namespace {

template <int C, int C1, int CIN2, int RES, int RB, int NCW, int NLW, int PD, int NS = 3, int WR = 4, bool H1 = false>
hipError_t launch_chain_t(const ChainArgs& a0, hipStream_t s) {
    constexpr int NF = ((C + CIN2) / 32 * 2 + C1 / 16) * (H1 ? 1 : 2) + 1;
    constexpr int LDS = NS * NF * 1024;
    static_assert(LDS <= 160 * 1024, "ring does not fit the LDS");
    auto kern = chain_kernel<C, C1, CIN2, RES, RB, NCW, NLW, NS, PD, WR, H1>;
    static bool attr_done[16] = {};
    static int wgs_per_cu[16] = {};
    const int dev = dgp_device_slot();
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * (NCW + NLW), LDS);
        if (e != hipSuccess) return e;
        static const int force = dgp_tune("DGP_CHAIN_WGS", 0);      // tuning: resident workgroups per CU
        wgs_per_cu[dev] = force > 0 ? force : (occ < 1 ? 1 : occ);
        attr_done[dev] = true;
    }
    ChainArgs a = a0;
    static const int nt_env = dgp_tune("DGP_CHAIN_NT", 0);      // bit 0: residual loads, bit 1: X' stores non-temporal
    a.nt = nt_env;
    constexpr int TILE = 16 * RB * NCW;
    a.ntiles = (a.M + TILE - 1) / TILE;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = ncu * wgs_per_cu[dev];
    if (grid > a.ntiles) grid = a.ntiles;
    if (grid < 1) return hipSuccess;
#ifdef DGP_DIAG
    unsigned long long hz[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_unit_diag), hz, sizeof(hz));
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * (NCW + NLW)), LDS, s, a);
#ifdef DGP_DIAG
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(hz, HIP_SYMBOL(g_unit_diag), sizeof(hz));
    const double nt = (double)hz[9] > 0 ? (double)hz[9] : 1.0, ns = (double)hz[15] > 0 ? (double)hz[15] : 1.0;
    printf("[diag chain C %d C1 %d CIN2 %d] grid %d tiles %d (%.1f per workgroup) | compute wave 0, cycles per TILE: prologue %.0f | conv2: 9 x (work %.0f + barrier %.0f) "
           "epilogue %.0f | chain: %d x (conv3 %.0f + epilogue %.0f + conv1 %.0f + barrier %.0f) conv1 epilogue %.0f | tile loop %.0f per tile | "
           "loader wave 0, cycles per STEP: issue %.0f (conv2 steps) / %.0f (pointwise steps), vmcnt wait %.0f, barrier %.0f\n", C, C1, CIN2, grid, a.ntiles, (double)a.ntiles / grid,
           hz[3] / nt, hz[0] / nt / 9.0, hz[1] / nt / 9.0, hz[2] / nt, 4 * C / 32, hz[4] / nt / (4 * C / 32), hz[5] / nt / (4 * C / 32), hz[6] / nt / (4 * C / 32),
           hz[7] / nt / (4 * C / 32), hz[8] / nt, hz[10] * (double)grid / nt / grid, hz[12] / (ns * 9.0 / (9.0 + 4 * C / 32)), hz[11] / (ns * (4 * C / 32) / (9.0 + 4 * C / 32)), hz[13] / ns, hz[14] / ns);
#endif
    return hipGetLastError();
}

template <int C, int C1, int CIN2, int RES, int NCW, int NLW, int PD, int WR = 4, int HWV = 1, bool H1 = false>
hipError_t launch_unit_t(const ChainArgs& a0, int N, hipStream_t s) {
    constexpr int NFJ = ((C + CIN2) / 32 * 2 + C1 / 16) * (H1 ? 1 : 2) + 1, NF2 = (C / 32) * (C / 16) * (H1 ? 1 : 2) + 1, NFMAX = NFJ > NF2 ? NFJ : NF2;
    constexpr int LDS = ((NCW + 2) * 18 * C * (H1 ? 2 : 4) + 1023) / 1024 * 1024 + 3 * NFMAX * 1024;
    static_assert(LDS <= 160 * 1024, "halo tile + ring do not fit the LDS");
    auto kern = unit_kernel<C, C1, CIN2, RES, NCW, NLW, PD, WR, HWV, H1>;
    static bool attr_done[16] = {};
    static int wgs_per_cu[16] = {};
    const int dev = dgp_device_slot();
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * (NCW + NLW + HWV), LDS);
        if (e != hipSuccess) return e;
        static const int force = dgp_tune("DGP_CHAIN_WGS", 0);
        wgs_per_cu[dev] = force > 0 ? force : (occ < 1 ? 1 : occ);
        attr_done[dev] = true;
    }
    ChainArgs a = a0;
    static const int nt_env = dgp_tune("DGP_CHAIN_NT", 0);
    a.nt = nt_env;
    a.TY = (a.H + NCW - 1) / NCW; a.TX = (a.W + 15) / 16;
    a.ntiles = N * a.TY * a.TX;
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    int grid = ncu * wgs_per_cu[dev];
    if (grid > a.ntiles) grid = a.ntiles;
    if (grid < 1) return hipSuccess;
#ifdef DGP_DIAG
    unsigned long long hz[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_unit_diag), hz, sizeof(hz));
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * (NCW + NLW + HWV)), LDS, s, a);
#ifdef DGP_DIAG
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(hz, HIP_SYMBOL(g_unit_diag), sizeof(hz));
    const double nt = (double)hz[9] > 0 ? (double)hz[9] : 1.0, ns = (double)hz[15] > 0 ? (double)hz[15] : 1.0;
    printf("[diag unit  C %d C1 %d CIN2 %d] grid %d tiles %d (%.1f per workgroup) | compute wave 0, cycles per TILE: prologue %.0f | conv2: 9 x (work %.0f + barrier %.0f) "
           "epilogue %.0f | chain: %d x (conv3 %.0f + epilogue %.0f + conv1 %.0f + barrier %.0f) conv1 epilogue %.0f | tile loop %.0f per tile | "
           "loader wave 0, cycles per STEP: issue %.0f (conv2 steps) / %.0f (pointwise steps), vmcnt wait %.0f, barrier %.0f\n", C, C1, CIN2, grid, a.ntiles, (double)a.ntiles / grid,
           hz[3] / nt, hz[0] / nt / 9.0, hz[1] / nt / 9.0, hz[2] / nt, 4 * C / 32, hz[4] / nt / (4 * C / 32), hz[5] / nt / (4 * C / 32), hz[6] / nt / (4 * C / 32),
           hz[7] / nt / (4 * C / 32), hz[8] / nt, hz[10] * (double)grid / nt / grid, hz[12] / (ns * 9.0 / (9.0 + 4 * C / 32)), hz[11] / (ns * (4 * C / 32) / (9.0 + 4 * C / 32)), hz[13] / ns, hz[14] / ns);
#endif
    return hipGetLastError();
}

}  // namespace

bool chain_supported(int C, int C1, int CIN2, int res) {
    if (C == 64 && C1 == 64 && CIN2 == 0 && res == 1) return true;
    if (C == 64 && C1 == 64 && CIN2 == 64 && res == 0) return true;
    if (C == 64 && C1 == 128 && CIN2 == 0 && res == 2) return true;
    if (C == 128 && C1 == 128 && CIN2 == 0 && res == 1) return true;
    if (C == 128 && C1 == 256 && CIN2 == 0 && res == 2) return true;
    if (C == 256 && C1 == 256 && CIN2 == 0 && res == 1) return true;
    return false;
}

bool unit_supported(int C, int C1, int CIN2, int res) {
    return C == 64 && C1 == 64 && ((CIN2 == 0 && res == 1) || (CIN2 == 64 && res == 0));
}

hipError_t launch_unit(const ChainArgs& a, int N, int C, int C1, int CIN2, int res, hipStream_t s) {
    static const int cfg = dgp_tune("DGP_UNIT_CFG", 0);      // tuning: tile rows / loader waves
    if (C == 64 && C1 == 64 && CIN2 == 0 && res == 1) {
        // (measured on the batch-32 640x480 shape, ms per launch: 8 rows + 2 loader waves 0.41-0.42; 10 rows 0.42-0.43; 8 rows + 4 loader
        //  waves 0.42-0.43 -- the weight stream is not the pace; 4 rows + 1 loader, two workgroups per CU 0.56-0.67: 4.5 KB of weight
        //  fragments per pixel; four residual buffers (three chunks ahead, requested before conv2) 0.57: spills at the 168-register cap)
        if (cfg == 1) return launch_unit_t<64, 64, 0, 1, 8, 4, 2, 4, 0>(a, N, s);          // (12 waves: a 13th would cap the registers at 128)
        if (cfg == 2) return launch_unit_t<64, 64, 0, 1, 10, 2, 2, 4, 0>(a, N, s);
        if (cfg == 3) return launch_unit_t<64, 64, 0, 1, 8, 2, 2, 4, 1>(a, N, s);          // with the halo wave
        // (identity units move 1.57 GB per launch at 3.6-3.7 TB/s: the halo wave measured -2 % on one box, +1 / +1.5 % on two others)
        // (16-bit tier, ms per launch on one box: 2 loader waves 0.257, 4 loader waves 0.272 (4 residual buffers), 4 loader waves + halo wave 0.240 -- with a third of
        //  the MFMAs the loader waves' issue rate is what a pointwise step waits for)
        if (a.h1) return launch_unit_t<64, 64, 0, 1, 8, 4, 2, 4, 1, true>(a, N, s);
        return launch_unit_t<64, 64, 0, 1, 8, 2, 2, 4, 0>(a, N, s);
    }
    if (C == 64 && C1 == 64 && CIN2 == 64 && res == 0) {
        if (cfg == 1) return launch_unit_t<64, 64, 64, 0, 8, 4, 2, 8, 0>(a, N, s);
        if (cfg == 2) return launch_unit_t<64, 64, 64, 0, 10, 2, 2, 8, 0>(a, N, s);
        if (cfg == 3) return launch_unit_t<64, 64, 64, 0, 8, 2, 2, 8, 0>(a, N, s);          // no halo wave: the halo pieces in the weight loaders' queues
        // (halo wave: 0.402 / 0.402 -> 0.367 / 0.378 ms alternating on one box, 0.406-0.417 -> 0.391-0.411 on two others)
        if (a.h1) return launch_unit_t<64, 64, 64, 0, 8, 2, 2, 8, 1, true>(a, N, s);
        return launch_unit_t<64, 64, 64, 0, 8, 2, 2, 8, 1>(a, N, s);
    }
    return hipErrorInvalidValue;
}

int chain_frags_per_chunk(int C, int C1, int CIN2, int planes) { return ((C + CIN2) / 32 * 2 + C1 / 16) * planes + 1; }

const char* chain_kernel_name(int C, int C1, int CIN2, int res) {
    static thread_local char buf[64];
    snprintf(buf, sizeof buf, "chain_c%d_n%d_%s", C, C1, res == 0 ? "sc" : (res == 2 ? "s2" : "id"));
    (void)CIN2;
    return buf;
}

hipError_t launch_chain(const ChainArgs& a, int C, int C1, int CIN2, int res, hipStream_t s) {
    static const int cfg = dgp_tune("DGP_CHAIN_CFG", 0);      // tuning: alternative workgroup shapes
    //                                                                        C   C1  CIN2 RES RB NCW NLW PD
    if (C == 64 && C1 == 64 && CIN2 == 0 && res == 1) {
        if (cfg == 1) return launch_chain_t<64, 64, 0, 1, 2, 5, 1, 4>(a, s);
        if (cfg == 2) return launch_chain_t<64, 64, 0, 1, 2, 4, 1, 2>(a, s);
        if (cfg == 3) return launch_chain_t<64, 64, 0, 1, 1, 11, 1, 4>(a, s);
        if (cfg == 4) return launch_chain_t<64, 64, 0, 1, 1, 8, 2, 4>(a, s);          // more loader waves (4 / 5: two / four)
        if (cfg == 5) return launch_chain_t<64, 64, 0, 1, 1, 8, 4, 4>(a, s);
        if (a.h1) return launch_chain_t<64, 64, 0, 1, 1, 8, 1, 4, 3, 4, true>(a, s);
        return launch_chain_t<64, 64, 0, 1, 1, 8, 1, 4>(a, s);
    }
    if (C == 64 && C1 == 64 && CIN2 == 64 && res == 0) {
        if (cfg == 1) return launch_chain_t<64, 64, 64, 0, 2, 5, 1, 2>(a, s);
        if (cfg == 4) return launch_chain_t<64, 64, 64, 0, 1, 8, 2, 2>(a, s);
        if (cfg == 5) return launch_chain_t<64, 64, 64, 0, 1, 8, 4, 2>(a, s);
        if (a.h1) return launch_chain_t<64, 64, 64, 0, 1, 8, 1, 2, 3, 4, true>(a, s);
        return launch_chain_t<64, 64, 64, 0, 1, 8, 1, 2>(a, s);
    }
    if (C == 64 && C1 == 128 && CIN2 == 0 && res == 2) {
        if (cfg == 1) return launch_chain_t<64, 128, 0, 2, 2, 5, 1, 4>(a, s);
        if (cfg == 4) return launch_chain_t<64, 128, 0, 2, 1, 8, 2, 4>(a, s);
        if (cfg == 5) return launch_chain_t<64, 128, 0, 2, 1, 8, 4, 4>(a, s);
        if (a.h1) return launch_chain_t<64, 128, 0, 2, 1, 8, 1, 4, 3, 4, true>(a, s);
        return launch_chain_t<64, 128, 0, 2, 1, 8, 1, 4>(a, s);
    }
    if (C == 128 && C1 == 128 && CIN2 == 0 && res == 1) {
        if (cfg == 1) return launch_chain_t<128, 128, 0, 1, 1, 8, 2, 4, 3, 4>(a, s);
        if (cfg == 2) return launch_chain_t<128, 128, 0, 1, 1, 8, 2, 2, 3, 8>(a, s);
        if (cfg == 3) return launch_chain_t<128, 128, 0, 1, 1, 6, 2, 2, 3, 8>(a, s);
        if (cfg == 4 || cfg == 5) return launch_chain_t<128, 128, 0, 1, 1, 8, 4, 2, 3, 4>(a, s);
        if (a.h1) return launch_chain_t<128, 128, 0, 1, 1, 8, 2, 2, 3, 4, true>(a, s);
        return launch_chain_t<128, 128, 0, 1, 1, 8, 2, 2, 3, 4>(a, s);
    }
    if (C == 128 && C1 == 256 && CIN2 == 0 && res == 2) {
        if (a.h1) return launch_chain_t<128, 256, 0, 2, 1, 6, 2, 4, 3, 4, true>(a, s);
        return launch_chain_t<128, 256, 0, 2, 1, 6, 2, 4>(a, s);
    }
    if (C == 256 && C1 == 256 && CIN2 == 0 && res == 1) {          // 65-KiB chunks: two ring slots, one workgroup per CU
        if (cfg == 1) return launch_chain_t<256, 256, 0, 1, 1, 4, 2, 2, 2, 8>(a, s);
        if (cfg == 2) return launch_chain_t<256, 256, 0, 1, 1, 6, 2, 2, 2, 8>(a, s);
        if (cfg == 3) return launch_chain_t<256, 256, 0, 1, 1, 5, 2, 2, 2, 4>(a, s);
        if (a.h1) return launch_chain_t<256, 256, 0, 1, 1, 5, 2, 2, 2, 8, true>(a, s);
        return launch_chain_t<256, 256, 0, 1, 1, 5, 2, 2, 2, 8>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dgp
