// fp32 MFMA loop fed from LDS exactly like conv_igemm_f32's inner loop (no global traffic, no barriers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float4 sA[8 * 129], sB[8 * 128];
    for (int i = threadIdx.x; i < 8 * 129; i += 256) sA[i] = make_float4(in[i & 511], in[(i * 3) & 511], in[(i * 5) & 511], in[(i * 7) & 511]);
    for (int i = threadIdx.x; i < 8 * 128; i += 256) sB[i] = make_float4(in[(i * 11) & 511], in[(i * 13) & 511], in[(i * 17) & 511], in[(i * 19) & 511]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const float4* a_base = sA + (wave >> 1) * 64 + l31;
    const float4* b_base = sB + (wave & 1) * 64 + l31;
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int chunk = 2 * kc + half;
            float4 af[2], bf[2];
            if (MODE == 0) {          // operands from registers (no LDS in the loop)
                af[0] = make_float4(in[0], in[1], in[2], in[3]); af[1] = af[0]; bf[0] = af[0]; bf[1] = af[0];
            } else {
                af[0] = a_base[chunk * 129]; af[1] = a_base[chunk * 129 + 32];
                bf[0] = b_base[chunk * 128]; bf[1] = b_base[chunk * 128 + 32];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const float* in, float* out, int bpc) {
    const int grid = 256 * bpc, iters = 4000 / bpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int j = 0; j < 10; ++j) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 10.0 * grid * 4.0 * iters * 64.0 * 4096.0;
    printf("mode %d (0 regs, 1 LDS-fed, 2 LDS-fed + barrier/step) waves/SIMD %d: %.1f TFLOP/s\n", MODE, bpc, flops / ms / 1e9);
}

int main() {
    float *in, *out;
    if (hipMalloc(&in, 512 * 4) != hipSuccess || hipMalloc(&out, 256 * 8 * 256 * 4) != hipSuccess) return 1;
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    (void)hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int bpc : {1, 2, 4}) { run<0>(in, out, bpc); run<1>(in, out, bpc); run<2>(in, out, bpc); }
    return 0;
}
