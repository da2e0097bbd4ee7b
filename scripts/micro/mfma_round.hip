// How does v_mfma_f32_16x16x32_f16 round?  (the accumulation error of the fp16-split convs.)
// Row 0 / column 0 of D = c + sum_k a[0][k] * b[k][0]; cases are laid out in the k dimension of row 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D) {   // A[16][32], B[32][16], C[16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    half8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)A[r * 32 + 8 * g + t]; b[t] = (_Float16)B[(8 * g + t) * 16 + r]; }
    floatx4 c;
    for (int v = 0; v < 4; ++v) c[v] = C[(4 * g + v) * 16 + r];
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * g + v) * 16 + r] = c[v];
}
static float run(float c0, const float* prods, int n) {
    float hA[512] = {0}, hB[512] = {0}, hC[256] = {0}, hD[256];
    for (int i = 0; i < n; ++i) { hA[i] = prods[i]; hB[i * 16] = 1.0f; }
    hC[0] = c0;
    float *dA, *dB, *dC, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    return hD[0];
}
int main() {
    const float T = 16777216.0f;   // 2^24: ulp 2
    float p[32];
    p[0] = 1.5f;                              printf("2^24 + 1.5            -> 2^24 + %g   (RN: 2, RZ: 0)\n", run(T, p, 1) - T);
    p[0] = -1.5f;                             printf("-2^24 - 1.5           -> -2^24 + %g  (RN: -2, RZ: 0)\n", run(-T, p, 1) + T);
    p[0] = 1.0f;                              printf("2^24 + 1 (tie)        -> 2^24 + %g   (RNE: 0, RNA: 2)\n", run(T, p, 1) - T);
    p[0] = 3.0f;                              printf("2^24 + 3 (tie)        -> 2^24 + %g   (RNE: 4, RZ: 2)\n", run(T, p, 1) - T);
    p[0] = 0.5f; p[1] = 0.5f; p[2] = 0.5f;    printf("2^24 + 3 x 0.5        -> 2^24 + %g   (sum first + RN: 2; one by one: 0)\n", run(T, p, 3) - T);
    for (int i = 0; i < 32; ++i) p[i] = 0.25f; printf("2^24 + 32 x 0.25      -> 2^24 + %g   (exact: 8)\n", run(T, p, 32) - T);
    for (int i = 0; i < 32; ++i) p[i] = 0.03125f; p[0] = 1.0f;
                                              printf("2^24 + 1 + 31 x 2^-5  -> 2^24 + %g   (exact sum 1.97 -> RN 2; few guard bits: 0)\n", run(T, p, 32) - T);
    // products that cancel: (2^15 * 2^0) - (2^15 * 2^0) + 2^-10: internal alignment width
    p[0] = 32768.0f; p[1] = -32768.0f; p[2] = 0.0009765625f;
                                              printf("0 + 2^15 - 2^15 + 2^-10 -> %g   (exact: 0.000976562)\n", run(0.0f, p, 3));
    p[0] = 32768.0f; p[1] = 0.0009765625f;    printf("-2^15 + (2^15 + 2^-10)  -> %g   (exact: 0.000976562; fp32-aligned products: 0.000976562 needs 25 bits)\n", run(-32768.0f, p, 2));
    p[0] = 1.0f; p[1] = 5.9604644775390625e-8f * 0.75f;   // 1 + 0.75 * 2^-24: RN over the pair -> 1 + 2^-23?  (0.75 ulp/2...) 
                                              printf("0 + 1 + 0.75*2^-24      -> 1 + %g  (RN of exact: 0; ulp 1.19e-07)\n", run(0.0f, p, 2) - 1.0f);
    p[0] = 1.0f; p[1] = 5.9604644775390625e-8f * 1.5f;    printf("0 + 1 + 1.5*2^-24       -> 1 + %g  (RN: 1.19e-07, RZ: 0)\n", run(0.0f, p, 2) - 1.0f);
    // fp16 subnormal inputs: flushed?
    p[0] = 5.9604644775390625e-8f;            printf("subnormal fp16 input 2^-24 x 1 -> %g (kept: 5.96e-08, flushed: 0)\n", run(0.0f, p, 1));
    return 0;
}
