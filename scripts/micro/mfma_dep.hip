// Does a chain of DEPENDENT v_mfma_f32_16x16x32_f16 (same accumulator, back to back) issue at the rate of independent ones?
// NACC accumulators used round-robin (1 = every MFMA depends on the previous one, 3 = the chain / unit kernels' product triple on one
// accumulator is 1, 4 = the conv kernel's spacing); 1 or 2 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 mfma_dep.hip -o mfma_dep
// Measured (MI355X): one wave per SIMD 17.0 / 16.3 / 16.7 / 16.3 / 16.3 cycles per MFMA for 1 / 2 / 3 / 4 / 8 accumulators -- a dependent chain
// costs 4 % (the accumulator is forwarded).  With two waves per SIMD only wave 0 is timed: it keeps the pipe to itself when its next MFMA is
// dependent (17 cycles) and shares it evenly when it is not (32), an arbitration effect, not throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters, unsigned long long* clk) {
    floatx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)in[(threadIdx.x + t) & 511]; b[t] = (_Float16)in[(threadIdx.x * 3 + t) & 511]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 24; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NACC>
static void run(int waves_per_simd, const float* in, float* out, unsigned long long* clk) {
    const int iters = 4000, threads = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, in, out, iters, clk);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, in, out, iters, clk);
    hipDeviceSynchronize();
    unsigned long long c[256];
    hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)c[i]; avg /= 256.0;
    // s_memtime ticks at 100 MHz; report ticks per MFMA of one wave and the implied SIMD occupancy relative to NACC = 8
    printf("accumulators %d, waves/SIMD %d: %.3f memtime ticks per MFMA issued by a wave (x %d waves share the pipe)\n", NACC, waves_per_simd,
           avg / (iters * 24.0), waves_per_simd);
}

int main() {
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) { run<1>(w, in, out, clk); run<2>(w, in, out, clk); run<3>(w, in, out, clk); run<4>(w, in, out, clk); run<8>(w, in, out, clk); }
    return 0;
}
