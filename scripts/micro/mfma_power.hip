// Sustained 16-bit MFMA rate, shader clock and board power for the two fp16 MFMA shapes, register operands only:
//   v_mfma_f32_16x16x32_f16 (8 MACs per operand value read)  vs  v_mfma_f32_32x32x16_f16 (16 MACs per operand value read)
// The bench workload runs at the board's power cap (scripts/power_probe.py): which shape buys more FLOPs per joule?
// hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <glob.h>
#include <chrono>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(threadIdx.x + 31 * i) & 511]; b[i] = (_Float16)in[(threadIdx.x * 3 + 17 * i) & 511]; }
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        floatx4 c[8];
        for (int j = 0; j < 8; ++j) c[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    } else {
        floatx16 c[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double read_num(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return -1;
    double v = -1; if (fscanf(f, "%lf", &v) != 1) v = -1; fclose(f); return v;
}
static std::vector<std::string> globv(const char* pat) {
    glob_t g; std::vector<std::string> r;
    if (glob(pat, 0, nullptr, &g) == 0) for (size_t i = 0; i < g.gl_pathc; ++i) r.push_back(g.gl_pathv[i]);
    globfree(&g); return r;
}
static int read_sclk(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r"); if (!f) return -1;
    char line[128]; int mhz = -1;
    while (fgets(line, sizeof line, f)) { std::string l(line); if (l.find('*') != std::string::npos) { size_t c = l.find(':'); mhz = atoi(l.c_str() + c + 1); } }
    fclose(f); return mhz;
}

template <int SHAPE>
void run(const float* in, float* out, const std::vector<std::string>& pw, const std::vector<std::string>& ck, double seconds) {
    const int grid = 256 * 2, iters = 4000;            // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    const double flops_per_launch = (double)grid * 4.0 * iters * 4.0 * (SHAPE == 16 ? 8 * 16.0 * 16 * 32 * 2 : 4 * 32.0 * 32 * 16 * 2);
    hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, in, out, iters); hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    double n = 0, psum = 0, pmax = 0, csum = 0; int ns = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int j = 0; j < 20; ++j) hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, in, out, iters);
        // sample while the queue is busy
        double p = 0; for (auto& f : pw) { double v = read_num(f); if (v > p) p = v; }
        int c = 0; for (auto& f : ck) { int v = read_sclk(f); if (v > 0) { c = v; if (p > 0) break; } }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds / 3) { psum += p / 1e6; if (p / 1e6 > pmax) pmax = p / 1e6; csum += c; ++ns; }
        hipDeviceSynchronize(); n += 20;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.0f TFLOP/s sustained over %.1f s, board power %.0f W (max %.0f), sclk %.0f MHz -> %.2f TFLOP/s per W\n",
           SHAPE == 16 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_32x32x16_f16", n * flops_per_launch / dt / 1e12, dt, psum / (ns ? ns : 1), pmax, csum / (ns ? ns : 1),
           n * flops_per_launch / dt / 1e12 / (psum / (ns ? ns : 1)));
}

int main() {
    float *in, *out;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 512 * 256 * 4);
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)(((unsigned)i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    auto pw = globv("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input");
    auto ck = globv("/sys/class/drm/card*/device/pp_dpm_sclk");
    for (int rep = 0; rep < 2; ++rep) { run<16>(in, out, pw, ck, 4.0); run<32>(in, out, pw, ck, 4.0); }
    return 0;
}
