// LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) issue rate from an L2-resident buffer, as a function of the number of
// waves per CU that issue.  One workgroup per CU; every wave copies 1-KiB fragments of a 256-KB region into its own 8 KB of LDS, at most
// INFL instructions in flight per wave.   hipcc -O3 --offload-arch=gfx950 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;

template <int INFL>
__global__ __launch_bounds__(1024) void k(const float* in, int nbytes, int iters, unsigned long long* clk, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* dst = smem + wave * 8192;
    int src = (wave * 8192 + (int)blockIdx.x * 1024) & (nbytes - 1);
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + u * 1024), 16, lane * 16, src + u * 1024, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL - 1) : "memory");
        }
        src = (src + 8192 * 16) & (nbytes - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x * 16 + wave] = t1 - t0;
    if (threadIdx.x == 0) out[blockIdx.x] = reinterpret_cast<const float*>(smem)[iters & 63];
}

template <int INFL>
static void run(int waves, const float* in, int nbytes, unsigned long long* clk, float* out) {
    const int iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<INFL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<INFL>, dim3(256), dim3(64 * waves), waves * 8192, 0, in, nbytes, iters, clk, out);
        hipDeviceSynchronize();
    }
    static unsigned long long c[256 * 16];
    hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) avg += (double)c[b * 16 + w];
    avg /= 256.0 * waves;
    printf("waves/CU %2d, <= %2d in flight per wave: %6.1f cycles per 1-KiB instruction per wave, %5.1f B/clk per CU\n", waves, INFL,
           avg / (iters * 8.0), waves * 1024.0 * iters * 8.0 / avg);
}

// second experiment: the same loop over source regions of growing size (every instruction a fresh 1-KiB piece, pieces 33 KiB apart so that a
// region is walked many times before a line repeats): L2 hits -> Infinity-Cache hits -> HBM
template <int INFL>
__global__ __launch_bounds__(1024) void kbig(const float* in, unsigned mask, int iters, unsigned long long* clk, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)0x7fffffff, 0x00020000);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* dst = smem + wave * 8192;
    unsigned src = ((unsigned)(mask == 0x3fffffffu ? blockIdx.x : (blockIdx.x & 7)) * 4097u * 1024u + (unsigned)wave * 8192u) & mask;   // (up to 128 MB: the workgroups of an XCD walk the same pieces)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + u * 1024), 16, lane * 16, (int)src, 0, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL - 1) : "memory");
            src = (src + 33u * 1024u) & mask;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x * 16 + wave] = t1 - t0;
    if (threadIdx.x == 0) out[blockIdx.x] = reinterpret_cast<const float*>(smem)[iters & 63];
}
template <int INFL = 8>
static void run_big(int waves, const float* in, unsigned region, unsigned long long* clk, float* out) {
    const int iters = 400;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kbig<INFL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kbig<INFL>, dim3(256), dim3(64 * waves), waves * 8192, 0, in, region - 1, iters, clk, out);
        (void)hipDeviceSynchronize();
    }
    static unsigned long long c[256 * 16];
    (void)hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) avg += (double)c[b * 16 + w];
    avg /= 256.0 * waves;
    printf("source region %5u MB, waves/CU %d, <= %2d in flight: %6.1f cycles per 1-KiB instruction per wave, %5.1f B/clk per CU\n", region >> 20, waves, INFL,
           avg / (iters * 8.0), waves * 1024.0 * iters * 8.0 / avg);
}

int main() {
    if (1) {
        float* big; float* out2; unsigned long long* clk2;
        (void)hipMalloc(&big, 1u << 30); (void)hipMalloc(&out2, 4096); (void)hipMalloc(&clk2, 256 * 16 * 8);
        (void)hipMemset(big, 0, 1u << 30);
        for (unsigned mb : {1u, 2u, 8u, 32u, 128u, 1024u}) for (int w : {1, 2, 4}) run_big(w, big, mb << 20, clk2, out2);
        // in-flight depth against a long-latency source: every workgroup walks its OWN pieces of 1 GB (no sharing inside the XCD)
        run_big<1>(1, big, 1u << 30, clk2, out2); run_big<4>(1, big, 1u << 30, clk2, out2); run_big<8>(1, big, 1u << 30, clk2, out2);
        run_big<16>(1, big, 1u << 30, clk2, out2); run_big<32>(1, big, 1u << 30, clk2, out2); run_big<60>(1, big, 1u << 30, clk2, out2);
        (void)hipFree(big);
    }
    const int nbytes = 256 * 1024;
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, nbytes); hipMalloc(&out, 4096); hipMalloc(&clk, 256 * 16 * 8);
    hipMemset(in, 0, nbytes);
    for (int w : {1, 2, 4, 8, 16}) { run<1>(w, in, nbytes, clk, out); run<2>(w, in, nbytes, clk, out); run<4>(w, in, nbytes, clk, out); run<8>(w, in, nbytes, clk, out); run<16>(w, in, nbytes, clk, out); }
    return 0;
}
