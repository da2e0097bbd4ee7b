// Device-side helpers shared by the kernel files: raw buffer loads, range tracking, the fp16 high / low split and the H2 cell
// format.  Not part of the ABI.
#pragma once
#include "dgp_internal.h"

namespace dgp {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    // raw buffer load: an offset >= num_records returns zeros (hardware range check), which is
    // how zero padding, ragged tile edges and padded taps are produced without data selects
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

constexpr unsigned OOB = 0xFFFFFFF0u;

// Wave-wide maximum of NON-NEGATIVE floats (maxima of |x|: they order like their bit patterns) by DPP -- quad butterfly, half-row and
// row mirror: 4 VALU; then one v_readlane per row -- instead of six ds_bpermute round trips through the LDS pipe the co-resident
// workgroup's K loop keeps busy.  Wave-uniform result (lives in an SGPR).
__device__ __forceinline__ unsigned wave_max_bits(float x) {
    int v = (int)__float_as_uint(x);
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));       // quad_perm [1, 0, 3, 2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));       // quad_perm [2, 3, 0, 1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false));      // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));      // row_mirror
    return (unsigned)max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                         max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// Range tracking for the fp16-split kernels.  A tensor's max |x| lives in ABSMAX_SLOTS device floats (the maximum of
// the slots is the value): producers spread their atomics over the slots -- tens of thousands of waves maxing into
// ONE address serialise at the memory side (measured: a 0.3 ms layer became 0.9 ms) -- and skip the atomic when the
// slot already holds a value at least as large.  Non-negative floats order like their bit patterns.
__device__ __forceinline__ void track_absmax(float* slots, float amax, int lane, int salt) {
    const unsigned bits = wave_max_bits(amax);
    if (lane == 0) {
        unsigned* s = reinterpret_cast<unsigned*>(slots) + (salt & (ABSMAX_SLOTS - 1));
        if (bits > __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(s, bits);
    }
}

// track_absmax with the slot's value already in hand (requested earlier, so that no memory round trip stands at the end of the wave: a
// stale value costs at most one redundant atomic).
__device__ __forceinline__ void track_absmax_known(float* slots, float amax, int lane, int salt, unsigned known_bits) {
    const unsigned m = wave_max_bits(amax);
    if (lane == 0 && m > known_bits) atomicMax(reinterpret_cast<unsigned*>(slots) + (salt & (ABSMAX_SLOTS - 1)), m);
}

// max over the slots; every lane of the calling wave gets the value
__device__ __forceinline__ float read_absmax(const float* slots, int lane) {
    static_assert(ABSMAX_SLOTS == 256, "one float4 per lane");
    const float4 v = reinterpret_cast<const float4*>(slots)[lane];
    return __uint_as_float(wave_max_bits(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))));      // (slots hold maxima of |x|: non-negative, never NaN)
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float pow2_scale_for(const float* absmax, int lane, const float* absmax2 = nullptr) {
    // 2^(14 - E) for max = m 2^E (1 <= m < 2); 1 for an all-zero (or untracked) tensor.  absmax2: a second tensor that
    // shares the scale (K-concatenated sources)
    if (!absmax) return 1.f;
    float mx = read_absmax(absmax, lane);
    if (absmax2) mx = fmaxf(mx, read_absmax(absmax2, lane));
    const unsigned mb = __float_as_uint(mx);
    const int be = (int)((mb >> 23) & 0xFF);            // biased exponent
    if (be == 0 || be == 0xFF) return 1.f;
    int se = 127 + 14 - (be - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}

// Predicted-range fp16 copies (ConvArgs::shadow).  The copy of a tensor is written while its range is still being measured, so its
// scale comes from the range the tensor had one training step earlier: max = m 2^E maps to [2^10, 2^11), five bits below the fp16
// overflow.  0: no usable previous range (first step, all-zero or non-finite tensor) -- no copy.
__device__ __forceinline__ float shadow_scale_for(const float* prev, int lane) {
    if (!prev) return 0.f;
    const unsigned mb = __float_as_uint(read_absmax(prev, lane));
    const int be = (int)((mb >> 23) & 0xFF);
    if (be == 0 || be == 0xFF) return 0.f;
    const int se = 127 + 10 - (be - 127);
    if (se < 1 || se > 254) return 0.f;
    return __uint_as_float((unsigned)se << 23);
}
// A copy written with `scale` is usable when this step's measured maximum stayed inside [2^4, 65000) after scaling: no overflow,
// and the largest values keep both fp16 pieces normal (22 bits).  NaN compares false: not usable.
__device__ __forceinline__ bool shadow_usable(float scale, const float* cur, int lane) {
    if (!(scale > 0.f) || !cur) return false;
    const float v = read_absmax(cur, lane) * scale;
    return v >= 16.f && v < 65000.f;
}

// scales of H2 tensors inside a kernel: the host-known float, or the prediction from the previous step's range slots (ConvArgs::*_scale_dev)
__device__ __forceinline__ float h2_in_scale(const ConvArgs& p, int lane) { return p.in_scale_dev ? shadow_scale_for(p.in_scale_dev, lane) : p.in_scale; }
__device__ __forceinline__ float h2_out_scale(const ConvArgs& p, int lane) { return p.out_scale_dev ? shadow_scale_for(p.out_scale_dev, lane) : p.out_scale; }
__device__ __forceinline__ float h2_res_inv_scale(const ConvArgs& p, int lane) {
    return p.res_scale_dev ? 1.f / shadow_scale_for(p.res_scale_dev, lane) : p.res_inv_scale;
}

__device__ __forceinline__ void split2_f16(const float4 v, const float s, uint2& ph, uint2& pl) {
#if defined(DGP_SPLIT_PK)
    const float2v x01 = {v.x * s, v.y * s}, x23 = {v.z * s, v.w * s};
    const half2v h01 = __builtin_convertvector(x01, half2v), h23 = __builtin_convertvector(x23, half2v);
    const float2v r01 = x01 - __builtin_convertvector(h01, float2v), r23 = x23 - __builtin_convertvector(h23, float2v);   // exact
    const half2v l01 = __builtin_convertvector(r01, half2v), l23 = __builtin_convertvector(r23, half2v);
    ph.x = __builtin_bit_cast(unsigned, h01); ph.y = __builtin_bit_cast(unsigned, h23);
    pl.x = __builtin_bit_cast(unsigned, l01); pl.y = __builtin_bit_cast(unsigned, l23);
#else
    // 10 VALU per 4 values, written out because hipcc computes the high parts twice (16): h = f16(s x) straight into its half of
    // the packed register (v_fma_mixlo/hi_f16), r = s x - h exactly in fp32 with h read as an fp16 operand (v_fma_mix_f32),
    // l = f16(r) packed (v_cvt_pk_f16_f32).  Scalar fp32 arithmetic on purpose: packed fp32 VALU ops are slow beside MFMAs.
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
    const float2v r01 = {r0, r1}, r23 = {r2, r3};
    const half2v l01 = __builtin_convertvector(r01, half2v), l23 = __builtin_convertvector(r23, half2v);
    ph.x = h01; ph.y = h23;
    pl.x = __builtin_bit_cast(unsigned, l01); pl.y = __builtin_bit_cast(unsigned, l23);
#endif
}

// ---- H2 activation format helpers -------------------------------------------------------------------------------------------
// 8 consecutive channels of a pixel = one 32-byte cell pair [8 halves high | 8 halves low] holding x * scale (ConvArgs::in_fmt)
__device__ __forceinline__ void h2_pack8(const float (&v)[8], float scale, uint4& hi, uint4& lo) {
    uint2 h0, l0, h1, l1;
    split2_f16(make_float4(v[0], v[1], v[2], v[3]), scale, h0, l0);
    split2_f16(make_float4(v[4], v[5], v[6], v[7]), scale, h1, l1);
    hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
    lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
__device__ __forceinline__ void h2_unpack8(const uint4 hi, const uint4 lo, float inv_scale, float (&v)[8]) {
    const half8 h = __builtin_bit_cast(half8, hi), l = __builtin_bit_cast(half8, lo);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = ((float)h[k] + (float)l[k]) * inv_scale;      // hi + lo is exact in fp32 (22 bits)
}

// ---- H1 activation format (the 16-bit tier): 8 consecutive channels of a pixel = ONE 16-byte cell of 8 halves holding
// fp16(x * scale) -- bit for bit the HIGH cell of the H2 pair (same v_fma_mixlo/hi_f16 rounding), i.e. plain NHWC fp16 with a
// per-tensor power-of-two scale.  Half the bytes of H2 / fp32 in HBM, L2 and LDS; 11 significant bits.
__device__ __forceinline__ uint4 h1_pack8(const float (&v)[8], float scale) {
    unsigned h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[k]) : "v"(scale), "v"(v[2 * k]));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[k]) : "v"(scale), "v"(v[2 * k + 1]));
    }
    return make_uint4(h[0], h[1], h[2], h[3]);
}
__device__ __forceinline__ void h1_unpack8(const uint4 c, float inv_scale, float (&v)[8]) {
    const half8 h = __builtin_bit_cast(half8, c);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)h[k] * inv_scale;
}

// ReLU gate read from an H2 tensor: 4 channels' high halves (h) and low halves (l); the stored value hi + lo is > 0 exactly when
// hi > 0, or hi == 0 and lo > 0.  Returned as 4 floats that compare > 0 like the value.
__device__ __forceinline__ float h2_half_bits(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)v); }
__device__ __forceinline__ float4 h2_gate4(const u32x2 h, const u32x2 l) {
    // (written on the 16-bit patterns: a bit_cast of h.y to a two-half vector was compiled as a second read of h.x)
    const unsigned h0 = h[0], h1 = h[1], l0 = l[0], l1 = l[1];
    return make_float4(h2_half_bits(h0 & 0xFFFFu) + h2_half_bits(l0 & 0xFFFFu), h2_half_bits(h0 >> 16) + h2_half_bits(l0 >> 16),
                       h2_half_bits(h1 & 0xFFFFu) + h2_half_bits(l1 & 0xFFFFu), h2_half_bits(h1 >> 16) + h2_half_bits(l1 >> 16));
}

}  // namespace dgp
