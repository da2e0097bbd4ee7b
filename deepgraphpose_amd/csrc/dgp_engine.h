// Engine-internal types shared by dgp_net.hip (inference engine) and dgp_train.hip (training step).
#pragma once
#include "../../include/dgp_hip.h"
#include "dgp_internal.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace dgp {
int fail(int code, const std::string& msg);      // sets the thread-local error string
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return dgp::fail(DGP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    } while (0)

namespace dgp {

struct ConvLayer {
    std::string scope;        // TF scope of the conv (".../conv1"), weights = scope + "/weights"
    int Cin = 0, Cout = 0, CoutP = 0, KH = 1, KW = 1, stride = 1, rate = 1;
    int nk = 0, ntaps = 1;
    bool has_bn = true, relu = false;
    float *d_w = nullptr, *d_scale = nullptr, *d_bias = nullptr;
    void* d_wh3 = nullptr;    // fp16 high/low cells of the panel (fp16-split kernels), built at load
    void* d_wh1 = nullptr;    // 16-bit tier: high-only cells (launch_pack_h1), built at load for layers with K % 64 == 0
    // conv3 of a unit with a shortcut conv: both 1x1 convs as ONE K-concatenated GEMM [R2 | X] [W3 s3 ; Wsc ssc] + (b3 + bsc)
    float *d_w_fused = nullptr, *d_bias_fused = nullptr;
    void* d_wh3_fused = nullptr;
    void* d_wh1_fused = nullptr;
    int nk_fused = 0, cin2 = 0;
    float* d_w_rows = nullptr;   // stem only: panel for the row walk [7 rows x 8 pixels][CoutP][4] (pixel 7 and channel 3 are zero)
    // heads only: the transposed conv as ONE pointwise GEMM [pixels x 2048] x [2048 x (4 taps x 4 phases x nj)] + a gather of the
    // <= 4 tap contributions per output pixel (reads the 2048-channel map once instead of four shifted times)
    float* d_w_pw = nullptr;
    void* d_wh3_pw = nullptr;
    void* d_wh1_pw = nullptr;
    int coutp_pw = 0;
};

struct Unit {
    int sc = -1, c1 = -1, c2 = -1, c3 = -1;
    int stride = 1, rate = 1;
    int depth_in = 0, depth = 0, depth_bn = 0;
};

// conv3 of unit k (+ shortcut, ReLU) and conv1 of unit k + 1 as ONE launch (dgp_chain.hip): the weight fragments of both convs in
// chunk order, built at load.  Scales: the fragments hold w * 2^w3_exp / w * 2^w1_exp (max |w| 2^exp in [2^14, 2^15)).
struct ChainPlan {
    bool ok = false;
    int C = 0, C1 = 0, CIN2 = 0, res = 0;     // res: 0 K-concatenated shortcut conv, 1 identity, 2 subsample (stride-2 unit)
    void* d_frags = nullptr;
    unsigned frag_bytes = 0;
    float *d_sc1 = nullptr, *d_bi1 = nullptr; // conv1's BN affine
    int w3_exp = 0, w1_exp = 0;
    bool unit = false;                        // the fragments start with conv2's 9 tap chunks: the unit kernel runs conv2 too
    int w2_exp = 0;
    unsigned head_bytes = 0;                  // bytes of conv2's chunks in front of the chain's (0 when unit is false)
    // the 16-bit tier's copy: the same chunks with the HIGH fragments only (half the L2 -> LDS stream; ChainArgs::h1)
    void* d_frags_h1 = nullptr;
    unsigned frag_bytes_h1 = 0, head_bytes_h1 = 0;
};
// host: build the chunked fragments (see ChainArgs) and upload them.  w3cat [(C + CIN2)][4 C], w1 [4 C][C1] row-major; sc3 may be null (= 1)
// w2 (optional, HWIO [3][3][C][C] of a stride-1 conv2 with BN affine sc2 / bi2): build the unit kernel's fragments when an instance exists
int build_chain_plan(ChainPlan& pl, int C, int C1, int CIN2, int res, const float* w3cat, const float* sc3, const float* bi3,
                     const float* w1, const float* sc1, const float* bi1, const float* w2 = nullptr, const float* sc2 = nullptr,
                     const float* bi2 = nullptr);
void free_chain_plan(ChainPlan& pl);

constexpr size_t TAIL_SLAB_FLOATS = (size_t)512 * 128 * 128;      // K-split slabs of a conv grid's tail: <= 512 slices of a 128 x 128 tile (32 MB)

int coutp_for(int cout);
int head_pw_coutp(int cpw);
int nk_for(int kh, int kw, int cin);
void tf_same(int n, int k, int s, int d, int* out, int* pad_before);
int pad_before_for(int n, int k, int stride, int rate, bool conv2d_same_explicit);

}  // namespace dgp

struct dgp_net {
    dgp_net_desc desc{};
    int device = 0;
    std::vector<dgp::ConvLayer> layers;
    int conv1 = -1, head_part = -1, head_locref = -1;
    std::vector<dgp::Unit> units;
    std::vector<dgp::ChainPlan> chains;       // chains[ui]: conv3 of unit ui + conv1 of unit ui + 1 (ok = false: layer by layer)
    bool loaded = false;
    // a trainer that owns this net (dgp_trainer_create) re-packs its panels every step and may defer the ones only the parity path reads;
    // dgp_forward calls this first so that a forward on a trainer-owned net never reads a stale panel (dgp_train.hip, refresh_parity_panels)
    int (*owner_sync)(void* owner, void* stream) = nullptr;
    void* owner = nullptr;
    // precision tier (dgp_net_set_tier): 0 = parity tier (H2 cells, 22-bit operands as fp16 pairs, three MFMAs per product);
    // 1 = 16-bit tier (H1 cells: 2-byte activations end to end, fp16 operands, one MFMA per product, fp32 accumulation / epilogues /
    // heads / soft-argmax).  Scales, calibration and the range check are shared; switching tiers re-calibrates.
    int tier = 0;
    // geometry
    int h1 = 0, w1 = 0, hp = 0, wp = 0, fh = 0, fw = 0;
    // operand ranges of the fp16-split conv kernels: ABSMAX_SLOTS floats per tensor.  d_wmax[li]: weight panel of
    // layer li (filled at load); d_amax[li]: output of layer li (zeroed and re-tracked every forward);
    // d_inmax: the centred frame (|pixel - mean| < 256, constant)
    float *d_wmax = nullptr, *d_amax = nullptr, *d_inmax = nullptr;
    float* tail_slab = nullptr;   // workspace region for the K-split of a conv grid's tail (set by the running forward)
    unsigned tail_slab_bytes = 0;
    bool wmax_valid = false;      // false after a trainer re-packed the panels: everything derived from the weights at load time
                                  // (ranges, fp16 cells, stem row panel, fused shortcut panels) is stale and the forward avoids it
    // H2 activation format (fp16 high / low cells written by the producing epilogue, ConvArgs::in_fmt): scale exponent of every
    // layer's output tensor (scale = 2^exp, max |tensor| * scale in [2^10, 2^11): 4 bits of headroom under the fp16 limit), set by a
    // calibration pass -- the first forward after the weights were loaded (or after an overflow was reported) runs layer by layer,
    // reads each layer's tracked range and re-launches it with the final scale.  Afterwards the scales are frozen: results are a
    // deterministic function of (frames, scales); a range that outgrows its scale is caught by h2_range_check_kernel (d_flag).
    static constexpr int H2_NONE = (int)0x80000000;
    std::vector<int> act_exp;          // per layer; H2_NONE: output is not an H2 tensor.  conv1's entry holds the POOL output's exponent
    std::vector<char> unit_fuse_ok;    // conv3 + shortcut as one GEMM needs R2 and X on one scale; false: run the shortcut conv separately
    bool h2_calibrated = false;
    int h2_head = 4;                   // bits of headroom between a calibrated maximum and the fp16 limit; +3 after every reported overflow
    int h2_calibrations = 0;
    int *d_exps = nullptr, *d_flag = nullptr;
    const float* wmax(int li) const { return d_wmax ? d_wmax + (size_t)li * dgp::ABSMAX_SLOTS : nullptr; }     // li + n_layers: fused panel of layer li
    float* amax(int li) const { return d_amax ? d_amax + (size_t)li * dgp::ABSMAX_SLOTS : nullptr; }
    // optional per-launch timing (hipEvent pairs recorded on the caller's stream)
    bool prof_on = false, prof_in_infer = false;
    int prof_slots = 0, prof_used = 0, prof_launches = 0, prof_cursor = 0, prof_launches_used = 0;
    std::vector<hipEvent_t> prof_ev;          // [slot][launch][2]
    std::vector<std::string> prof_names;      // per launch of the last profiled forward
    std::vector<double> prof_flops;
    void prof_free() {
        for (auto e : prof_ev) (void)hipEventDestroy(e);
        prof_ev.clear();
    }
    ~dgp_net() {
        prof_free();
        for (auto& l : layers) {
            if (l.d_w) (void)hipFree(l.d_w);
            if (l.d_scale) (void)hipFree(l.d_scale);
            if (l.d_bias) (void)hipFree(l.d_bias);
            if (l.d_wh3) (void)hipFree(l.d_wh3);
            if (l.d_w_rows) (void)hipFree(l.d_w_rows);
            if (l.d_w_pw) (void)hipFree(l.d_w_pw);
            if (l.d_wh3_pw) (void)hipFree(l.d_wh3_pw);
            for (void* q : {l.d_wh1, l.d_wh1_fused, l.d_wh1_pw}) if (q) (void)hipFree(q);
            for (void* q : {(void*)l.d_w_fused, (void*)l.d_bias_fused, l.d_wh3_fused}) if (q) (void)hipFree(q);
        }
        for (auto& c : chains) dgp::free_chain_plan(c);
        for (float* q : {d_wmax, d_amax, d_inmax}) if (q) (void)hipFree(q);
        for (int* q : {d_exps, d_flag}) if (q) (void)hipFree(q);
    }
};
