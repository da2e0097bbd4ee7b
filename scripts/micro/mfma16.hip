#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {   // A[16][32], B[32][16] row-major, D[16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    half8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = (_Float16)A[r * 32 + 8 * g + t]; b[t] = (_Float16)B[(8 * g + t) * 16 + r]; }
    floatx4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * g + v) * 16 + r] = c[v];
}
int main() {
    float hA[512], hB[512], hD[256], ref[256];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 5 - 2); hB[i] = (float)((i * 3) % 7 - 3); }
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k2 = 0; k2 < 32; ++k2) s += hA[m * 32 + k2] * hB[k2 * 16 + n]; ref[m * 16 + n] = s; }
    float *dA, *dB, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) if (hD[i] != ref[i]) ++bad;
    printf("mfma 16x16x32 f16 layout: %d mismatches\n", bad);
    return bad != 0;
}
