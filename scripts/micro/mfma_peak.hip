// Sustained fp32-MFMA rate and shader clock on this MI355X: pure v_mfma_f32_32x32x2_f32 loop.
// hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters, unsigned long long* clk) {
    floatx16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = in[threadIdx.x], y = in[threadIdx.x + 256];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    float *in, *out; unsigned long long* clk;
    const int blocks_per_cu[] = {1, 2, 4};
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 256 * 8 * 16);
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int bpc : blocks_per_cu) {
        const int grid = 256 * bpc, iters = 20000 / bpc;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, iters, clk);
        hipDeviceSynchronize();
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> c(2 * grid);
            hipMemcpy(c.data(), clk, 16 * grid, hipMemcpyDeviceToHost);
            double flops = 10.0 * grid * 4.0 * iters * 16.0 * 4096.0;
            double ghz = (double)c[0] / (double)c[1] * 0.1;
            printf("waves/SIMD %d: %.1f TFLOP/s over %.1f ms, in-kernel clock %.3f GHz (memtime/memrealtime), cycles/MFMA %.1f\n",
                   bpc, flops / ms / 1e9, ms, ghz, (double)c[0] / (iters * 16.0 * bpc));
        }
    }
    return 0;
}
