// Why does ONE compute wave per SIMD run the 256-row tile's MFMA stream at ~25 cycles per v_mfma_f32_16x16x32_f16 (round 6)?
// The stream of the 64 x 128 wave tile without any memory: 32 accumulators (4 row blocks x 8 column blocks), 8 A quads, NB weight quads,
// fragment F feeds 4 MFMAs (row blocks 0-3) into column block F / 2.  Variants: PARK = a second wave per SIMD parked at a barrier (the
// loader waves of the real kernel), SPIN = that wave spinning on SALU / VALU instead.   hipcc -O3 --offload-arch=gfx950 mfma_w64.hip -o mfma_w64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0: 4 waves; 1: 8 waves, waves 4-7 parked at the barrier each step; 2: waves 4-7 issue VALU; 3: ONE accumulator pattern (c[i][0] only)
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, int iters, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6;
    if (wave >= 4) {
        if (MODE == 1) { for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier(); }
        if (MODE == 2) { float x = in[threadIdx.x & 511]; for (int i = 0; i < iters * 200; ++i) x = x * 1.0001f + 0.5f; out[blockIdx.x * 512 + threadIdx.x] = x; }
        return;
    }
    floatx4 c[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) c[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 al[4], ah[4], bq[8];
    for (int q = 0; q < 4; ++q) for (int t = 0; t < 8; ++t) { al[q][t] = (_Float16)in[(threadIdx.x + t + q) & 511]; ah[q][t] = (_Float16)in[(threadIdx.x * 5 + t + q) & 511]; }
    for (int q = 0; q < 8; ++q) for (int t = 0; t < 8; ++t) bq[q][t] = (_Float16)in[(threadIdx.x * 3 + t + 7 * q) & 511];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int F = 0; F < 16; ++F) {
            const int j = MODE == 3 ? 0 : F >> 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16((F & 1) ? ah[i] : al[i], bq[F & 7], c[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(int nblocks, const float* in, float* out, unsigned long long* clk, const char* what) {
    const int iters = 3000, threads = MODE == 0 || MODE == 3 ? 256 : 512;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(threads), 0, 0, in, out, iters, clk); hipDeviceSynchronize(); }
    unsigned long long c[256];
    hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < nblocks; ++i) avg += (double)c[i]; avg /= nblocks;
    printf("%-58s %3d workgroups: %.2f cycles per MFMA (%.0f per 64-MFMA step)\n", what, nblocks, avg / (iters * 64.0), avg / iters);
}

int main() {
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int nb : {1, 256}) {
        run<0>(nb, in, out, clk, "one wave per SIMD, 32 accumulators");
        run<3>(nb, in, out, clk, "one wave per SIMD, 4 accumulators (dependent after 4)");
        run<1>(nb, in, out, clk, "+ a second wave per SIMD parked at the step's barrier");
        run<2>(nb, in, out, clk, "+ a second wave per SIMD issuing VALU");
    }
    return 0;
}
