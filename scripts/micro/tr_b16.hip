// ds_read_b64_tr_b16 lane map check (gfx950): image [R][C] of uint16 = row*256 + col; every group of 16 lanes reads the 4 x 16 block
// at (r0, c0): lane 4q+p supplies the address of row r0+q, columns c0+4p..c0+4p+3 and lane i should receive column c0+i of rows
// r0..r0+3.   hipcc --offload-arch=gfx950 -O2 tr_b16.hip -o tr_b16 && ./tr_b16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[16][64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) img[i / 64][i % 64] = (unsigned short)((i / 64) * 256 + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int r0 = 4 * (g & 1), c0 = 16 * (g >> 1);          // groups: (rows 0-3, cols 0-15), (rows 4-7, cols 0-15), (0-3, 16-31), (4-7, 16-31)
    const unsigned addr = (unsigned)(size_t)&img[r0 + q][c0 + 4 * p];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane] = v;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned long long h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15, r0 = 4 * (g & 1), c0 = 16 * (g >> 1);
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            const unsigned v = (unsigned)((h[l] >> (16 * e)) & 0xFFFF);
            printf(" (r%u,c%u)", v >> 8, v & 255);
            if (v != (unsigned)((r0 + e) * 256 + c0 + i)) ++bad;
        }
        printf("\n");
    }
    printf("%s (%d mismatches vs 'lane i gets column c0+i, element q = row r0+q')\n", bad ? "MISMATCH" : "OK", bad);
    return 0;
}
