// buffer_load ... lds (LDS-DMA) semantics check: 16 bytes per lane land lane-linearly at the wave's M0 base, an out-of-range lane writes zeros.
// hipcc --offload-arch=gfx950 -o dma_lds dma_lds.hip && ./dma_lds
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* in, int nbytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave: 1 KB into LDS at wave * 1024; lane's source = swizzled
    int voff = (wave * 64 + (lane ^ 5)) * 16;
    if (lane == 7) voff = 0x7fffff00;   // OOB -> expect zeros
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + __builtin_amdgcn_readfirstlane(wave * 1024)), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float4 v = reinterpret_cast<const float4*>(smem)[threadIdx.x];
    reinterpret_cast<float4*>(out)[threadIdx.x] = v;
}
int main() {
    const int n = 256 * 4;
    float *in, *out; hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    float h[n]; for (int i = 0; i < n; ++i) h[i] = i;
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    hipMemset(out, 0xff, n * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, in, n * 4, out);
    float o[n]; hipMemcpy(o, out, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const int lane = t & 63, wave = t >> 6;
        for (int e = 0; e < 4; ++e) {
            const float want = lane == 7 ? 0.f : (float)(((wave * 64 + (lane ^ 5)) * 4) + e);
            if (o[t * 4 + e] != want) { if (bad < 8) printf("t %d e %d got %g want %g\n", t, e, o[t * 4 + e], want); ++bad; }
        }
    }
    printf("dma test: %d mismatches\n", bad);
    return bad != 0;
}
