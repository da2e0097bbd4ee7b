// Second step of scripts/micro/mfma_w64.hip: the 64 x 128 wave tile's K loop rebuilt piece by piece around the pure MFMA stream (1 053
// cycles per 64-MFMA step) -- which piece costs the ~550 cycles per step the real kernel loses?
//   1: + the 16 weight-fragment reads per step (ds_read_b128, ring of 8)          2: + the 8 A-cell reads and the ra -> ah / al copies
//   3: + one workgroup barrier per step (waves 4-7 parked at it)                   4: as 3, loop unrolled x 2 without the copies
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 128 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = in[i & 511];
    __syncthreads();
    if (wave >= 4) {
        if (MODE >= 3) { for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier(); }
        return;
    }
    const int l15 = lane & 15, g = lane >> 4;
    floatx4 c[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) c[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    uint4 ra[4][2], al[4], ah[4], bq[8];
    const uint4* B = reinterpret_cast<const uint4*>(smem + 96 * 1024) + l15 + g * 128;
    const unsigned a_row0 = (unsigned)((wave * 64 + l15) * 128 + (((2 * g) ^ (((wave * 64 + l15) >> 1) & 7)) << 4));
    unsigned a_cur = a_row0; int sa = 0, db = 1024;
    auto mma = [](const uint4& x, const uint4& y, floatx4 cc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, x), __builtin_bit_cast(half8, y), cc, 0, 0, 0);
    };
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define RA(I) do { ra[I][0] = *reinterpret_cast<const uint4*>(smem + a_cur + (I) * 2048); ra[I][1] = *reinterpret_cast<const uint4*>(smem + (a_cur ^ 16u) + (I) * 2048); } while (0)
#define RB(F) do { bq[(F) & 7] = B[((((F) & 1) ? 0 : 1) * 4) * 128 + 16 * (((F) % 16) >> 1)]; } while (0)
#define MM(F) do { constexpr int j_ = (F) >> 1; \
    if (((F) & 1) == 0) { c[0][j_] = mma(al[0], bq[(F) & 7], c[0][j_]); c[1][j_] = mma(al[1], bq[(F) & 7], c[1][j_]); c[2][j_] = mma(al[2], bq[(F) & 7], c[2][j_]); c[3][j_] = mma(al[3], bq[(F) & 7], c[3][j_]); } \
    else { c[0][j_] = mma(ah[0], bq[(F) & 7], c[0][j_]); c[1][j_] = mma(ah[1], bq[(F) & 7], c[1][j_]); c[2][j_] = mma(ah[2], bq[(F) & 7], c[2][j_]); c[3][j_] = mma(ah[3], bq[(F) & 7], c[3][j_]); } } while (0)
#define STEP(F) do { MM(F); FENCE(); RB((F) + 8); FENCE(); } while (0)
#define TAIL(F) do { MM(F); FENCE(); RB((F) - 8); FENCE(); } while (0)
    RA(0); RA(1); RA(2); RA(3);
    for (int q = 0; q < 8; ++q) RB(q);
    for (int i = 0; i < 4; ++i) { ah[i] = ra[i][0]; al[i] = ra[i][1]; }
    FENCE();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) { for (int i = 0; i < 4; ++i) { ah[i] = ra[i][0]; al[i] = ra[i][1]; } }
        FENCE();
        STEP(0); STEP(1); STEP(2); STEP(3); STEP(4); STEP(5); STEP(6); STEP(7);
        MM(8); FENCE(); MM(9); FENCE(); MM(10); FENCE(); MM(11); FENCE();
        if (MODE >= 3) __syncthreads();
        sa = sa == 2 ? 0 : sa + 1; a_cur = a_row0 + (unsigned)(sa * 32768);
        B += db; db = -db;
        if (MODE >= 2) { RA(0); RA(1); RA(2); RA(3); }
        RB(0); RB(1); RB(2); RB(3);
        FENCE();
        TAIL(12); TAIL(13); TAIL(14); TAIL(15);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(int nblocks, const float* in, float* out, unsigned long long* clk, const char* what) {
    const int iters = 3000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(512), 128 * 1024, 0, in, out, iters, clk); hipDeviceSynchronize(); }
    unsigned long long c[256];
    hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < nblocks; ++i) avg += (double)c[i]; avg /= nblocks;
    printf("%-64s %3d workgroups: %.2f cycles per MFMA (%.0f per 64-MFMA step)\n", what, nblocks, avg / (iters * 64.0), avg / iters);
}

int main() {
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int nb : {1, 256}) {
        run<1>(nb, in, out, clk, "MFMA stream + 16 weight-fragment reads per step");
        run<2>(nb, in, out, clk, "+ 8 A-cell reads and the ra -> ah / al copies");
        run<3>(nb, in, out, clk, "+ a workgroup barrier per step (waves 4-7 parked)");
    }
    return 0;
}
