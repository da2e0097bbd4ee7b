// LDS-DMA issue rate of WD loader waves while WR other waves of the same CU read the LDS (ds_read_b128 in a loop, optionally with MFMAs
// between the reads, like the compute waves of the chain / unit kernels).   hipcc -O3 --offload-arch=gfx950 dma_contend.hip -o dma_contend
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0: readers idle (exit), 1: ds_read_b128 only, 2: ds_read_b128 + 3 MFMAs per pair of reads, 3: MFMAs only
__global__ __launch_bounds__(1024) void k(const float* in, int nbytes, int iters, int wd, unsigned long long* clk, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave < wd) {
        char* dst = smem + 65536 + wave * 8192;
        int src = (wave * 8192 + (int)blockIdx.x * 1024) & (nbytes - 1);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + u * 1024), 16, lane * 16, src + u * 1024, 0, 0);
                asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            }
            src = (src + 8192 * 16) & (nbytes - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { clk[blockIdx.x * 16 + wave] = t1 - t0; atomicAdd(&done, 1); }
        return;
    }
    if (MODE == 0) return;
    // reader waves: run until every DMA wave has finished
    const uint4* W = reinterpret_cast<const uint4*>(smem) + lane;
    floatx4 acc[4] = {floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}, floatx4{0, 0, 0, 0}};
    uint4 s = make_uint4(0, 0, 0, 0);
    half8 b = __builtin_bit_cast(half8, make_uint4(lane, lane * 3, lane * 5, lane * 7));
    while (__atomic_load_n(&done, __ATOMIC_RELAXED) < wd) {
#pragma unroll
        for (int f = 0; f < 16; f += 2) {
            uint4 wh = make_uint4(1, 2, 3, 4), wl = make_uint4(5, 6, 7, 8);
            if (MODE != 3) { wh = W[f * 64]; wl = W[(f + 1) * 64]; }
            if (MODE >= 2) {
                acc[(f >> 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wl), b, acc[(f >> 1) & 3], 0, 0, 0);
                acc[(f >> 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wh), b, acc[(f >> 1) & 3], 0, 0, 0);
                acc[(f >> 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wh), b, acc[(f >> 1) & 3], 0, 0, 0);
            } else { s.x ^= wh.x ^ wl.y; s.y += wh.z + wl.w; }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)(s.x + s.y);
}

template <int MODE>
static void run(int wd, int wr, const float* in, int nbytes, unsigned long long* clk, float* out) {
    const int iters = 200;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * (wd + wr)), 65536 + wd * 8192, 0, in, nbytes, iters, wd, clk, out);
        (void)hipDeviceSynchronize();
    }
    static unsigned long long c[256 * 16];
    (void)hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < wd; ++w) avg += (double)c[b * 16 + w];
    avg /= 256.0 * wd;
    const char* names[] = {"idle", "ds_read_b128 only", "ds_read_b128 + MFMA", "MFMA only"};
    printf("%d DMA waves beside %d waves (%-20s): %6.1f cycles per 1-KiB instruction per wave, %5.1f B/clk per CU\n", wd, wr, names[MODE],
           avg / (iters * 8.0), wd * 1024.0 * iters * 8.0 / avg);
}

int main() {
    const int nbytes = 256 * 1024;
    float *in, *out; unsigned long long* clk;
    (void)hipMalloc(&in, nbytes); (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&clk, 256 * 16 * 8);
    (void)hipMemset(in, 0, nbytes);
    for (int wd : {1, 2, 4}) {
        run<0>(wd, 8, in, nbytes, clk, out); run<1>(wd, 8, in, nbytes, clk, out); run<2>(wd, 8, in, nbytes, clk, out); run<3>(wd, 8, in, nbytes, clk, out);
    }
    run<2>(2, 4, in, nbytes, clk, out);
    return 0;
}
