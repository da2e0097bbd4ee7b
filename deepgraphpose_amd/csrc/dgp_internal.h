// Internal declarations shared by the kernel file and the engine.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

namespace dgp {

constexpr int BK = 32;            // K floats per main-loop step (8 chunks of 4)

// Arguments of the implicit-GEMM convolution kernel (see dgp_kernels.hip).
struct ConvArgs {
    const float* in;      // NHWC [N,H,W,Cin]
    const float* wpk;     // packed weight panels [nk*8][CoutP][4]
    const float* scale;   // [Cout] or nullptr (=1)
    const float* bias;    // [Cout] or nullptr (=0)
    const float* res;     // residual NHWC [N,res_H,res_W,Cout] or nullptr
    const float* mask;    // backward ReLU gate, same shape as out, or nullptr
    float*       out;
    int N, H, W, Cin, log2cin4;
    int Ho, Wo, Cout, CoutP;
    int KH, KW, stride, dil, pad_t, pad_l;
    int ntaps, nk;        // real taps, number of BK steps
    int M;                // N*Ho*Wo
    int res_s, res_H, res_W;   // res_s == 0: none; s >= 1: residual[n, ho*s, wo*s]; -2: residual on the 2x coarser grid
    int ksplit;                // > 1: split-K over workgroups into per-split slabs (out_mode 1 only)
    long long split_stride;    // floats between slabs
    int up;                    // 2: data-gradient gather of a stride-2 conv (input on the zero-stuffed grid)
    int relu;
    int out_mode;         // 0: NHWC [M][Cout]; 1: transposed-conv phase scatter
    int dc_nj;            // channels per phase for out_mode 1
    int mtiles, ntiles;
    int tap_minor;        // split kernels: walk K as (channel chunk, tap) instead of (tap, channel chunk)
    int stem;             // split kernels: 4-channel input walked one kernel ROW per K-step (8 pixels x 4 channels = 128 contiguous bytes);
                          // chunk c of a row is the pixel wi0 + c, so the width check is per chunk
    int tap_rows;         // weight-panel rows (of 4 k-values) per tap; 0 = Cin / 4
    // split kernels, 1x1 stride-1 convs only: K-concatenated second source on the same pixel grid (the shortcut conv fused into
    // conv3: out = [R2 | X] [W3' ; Wsc']).  Cin = cin_split + Cin2; channels >= cin_split come from in2 [M][Cin - cin_split]
    // split kernels: K-split of the grid's TAIL.  Blocks >= n_main each compute 1/tail_ksplit of the K range of one of the last
    // tiles into a slab (raw accumulators); launch_conv then runs tail_fixup to sum the parts in fixed order and apply the epilogue
    float* slab;
    unsigned slab_bytes;
    int n_main, tail_ksplit;
    int epi_nt;                // H2 epilogue: non-temporal residual loads and output stores
    int st_gn, st_mc, st_mb;   // supertile order of the grid (0: plain): column tiles per group, row tiles per chunk, row tiles per XCD band
    const float* in2;
    const float* in2_absmax;
    unsigned in2_bytes;
    int cin_split;
    unsigned long long* dbg;
    unsigned in_bytes, w_bytes, res_bytes, out_bytes;   // buffer-descriptor extents (< 4 GiB each)
    // dynamic range tracking for the fp16-split kernels (device arrays of ABSMAX_SLOTS floats, may be null; the
    // tensor's value is the maximum over the slots):
    const float* in_absmax;   // max |x| over the input tensor (an upper bound is fine), written by its producer
    const float* w_absmax;    // max |w| over the weight panel
    const void*  wh3;         // optional: the weight panel pre-split into fp16 high/low cells (launch_pack_h3), same scale as w_absmax gives
    unsigned     wh3_bytes;
    float*       out_absmax;  // the epilogue atomically maxes max |out| into this slot (zeroed by the caller)
    // "H2" activation format (inference engine): a tensor lives in HBM as fp16 high/low cell pairs -- per pixel and 8 channels
    // [8 halves hi | 8 halves lo] = 32 bytes, the same footprint and the same addresses as 8 fp32 channels -- holding x * scale with
    // scale = 2^e chosen per tensor by the engine (calibrated, with headroom; host-known, so it travels as a kernel argument).
    // Consumers copy the cells straight into the MFMA operand image (no split arithmetic in the K loop); producers split once in
    // the epilogue.  0 = fp32, 1 = H2, 2 = H1 (the 16-bit tier: ONE 16-byte cell of 8 halves per 8 channels = the H2 pair's high cell,
    // half the bytes; wh3 then points at high-only weight cells, launch_pack_h1; in with out 0 or 2, residual 2).
    int   in_fmt, out_fmt, res_fmt;
    float in_scale;           // in (and in2) cells hold x * in_scale
    float out_scale;          // cells written hold out * out_scale
    float res_inv_scale;      // residual value = (hi + lo) * res_inv_scale
    // Training step: an optional second copy of the fp32 output as fp16 high / low cells in the H2 layout (the weight-gradient
    // kernel's LDS-DMA operand).  Its power-of-two scale is PREDICTED from the range slots the same tensor had one step earlier
    // (shadow_scale_for, dgp_device.h); the consumer checks this step's range against it and falls back to the fp32 tensor when
    // the prediction failed, so a copy written with a stale scale is never read.  No previous range: no copy is written.
    float*       shadow;
    const float* shadow_prev;
    int          shadow_fmt;   // 0 / 1: H2 cells (fp16 high / low, the parity trainer's wgrad_dma operands); 2: H1 cells (high only: the 16-bit trainer)
    // Training step, H2 tensors with PREDICTED scales: where a pointer is set, the tensor's scale is not the host-known float above
    // but shadow_scale_for(pointer) -- the range slots the same tensor had one step earlier -- read on the device.  The host checks
    // after the pass that every such tensor stayed inside its predicted range (dgp_train.hip, h2_pred_check_kernel) and repeats the
    // step on fp32 tensors when one did not.
    const float* in_scale_dev;
    const float* out_scale_dev;
    const float* res_scale_dev;
    int mask_fmt;             // 1: the ReLU gate tensor (mask) is H2: gate = stored value > 0; 2: H1 (stored half > 0)
};

constexpr int ABSMAX_SLOTS = 256;

enum TileCfg { TILE_128x128 = 0, TILE_128x64 = 1, TILE_64x64 = 2, TILE_128x32 = 3, TILE_128x128_W8 = 4, TILE_128x128_LS = 5, TILE_128x64_LS = 6, TILE_128x128_S6 = 7, TILE_128x128_S3 = 8, TILE_128x64_S6 = 9, TILE_128x128_S6K16 = 10, TILE_128x128_S3K16 = 11, TILE_128x128_S6K16W8 = 12, TILE_128x128_H3K16 = 13, TILE_128x128_H3K16W8 = 14, TILE_128x64_H3 = 15, TILE_128x128_H3K32 = 16 };

hipError_t launch_conv(const ConvArgs& a, int tile_cfg, hipStream_t s);
int        pick_tile(int M, int CoutP, int K, bool have_absmax = false);
// fp32 panel [nk*8][CoutP][4] -> fp16 cells [nk*4 k-groups][2 planes][CoutP][8 halves] in the LDS order of the fp16-split kernels
hipError_t launch_pack_h3(const float* panel, int nk, int CoutP, const float* w_absmax, void* out, hipStream_t s);
// fp32 panel -> H1 cells [nk*4 k-groups of 8 channels][CoutP][8 halves]: the high plane of launch_pack_h3's cells, compacted (16-bit tier)
hipError_t launch_pack_h1(const float* panel, int nk, int CoutP, const float* w_absmax, void* out, hipStream_t s);
hipError_t launch_f32_to_h1(const float* x, long long n_groups8, float scale, void* out, hipStream_t s);
hipError_t launch_h1_to_f32(const void* x, long long n_groups8, float inv_scale, float* out, hipStream_t s);
struct PackH3Desc { const float* panel; int nkg; int CoutP; const float* rng; void* out; };      // nkg = nk * 4 k-groups
hipError_t launch_pack_h3_all(const PackH3Desc* table_dev, int n, hipStream_t s);
hipError_t launch_pack_h1_all(const PackH3Desc* table_dev, int n, hipStream_t s);      // same table layout, out = H1 cells (half the bytes)
hipError_t launch_absmax(const float* x, long long n, float* out_slots, hipStream_t s);   // slots = max(slots, max |x|)
const char* conv_kernel_name(const ConvArgs& a, int tile_cfg);
hipError_t launch_reduce_slabs(const float* slabs, long long n, long long stride, int nsplit, float* out, hipStream_t s);
hipError_t launch_head_gather(const float* T, const float* bias, int B, int h, int w, int njt, int ldt, float* out, hipStream_t s);
hipError_t launch_maxpool(const float* x, int N, int H, int W, int C, float* y, hipStream_t s, float h2_scale = 0.f);   // h2_scale > 0: y in H2 format
// fp32 NHWC [n_pix][C] <-> H2 cells (C % 8 == 0); scale = the power of two the cells are (to be) stored with
hipError_t launch_f32_to_h2(const float* x, long long n_groups8, float scale, void* out, hipStream_t s);
hipError_t launch_h2_to_f32(const void* x, long long n_groups8, float inv_scale, float* out, hipStream_t s);
// flag[0] |= 1 where a tracked range times the tensor's scale leaves the fp16 range (limit 60000), per layer; exps: scale exponents
hipError_t launch_h2_range_check(const float* amax_slots, const int* exps, int n_layers, int* flag, hipStream_t s);
// root block fused: uint8 frames -> conv1 (7x7/2) + BN + ReLU -> 3x3/2 max-pool -> H2 cells [B, HP, WP, 64] with scale out_scale
hipError_t launch_stem_pool_fused(const unsigned char* frames, int B, int H, int W, const void* wcells, const float* w_absmax,
                                  const float* bn_scale, const float* bn_bias, float m0, float m1, float m2, float out_scale,
                                  float* out, float* out_absmax, hipStream_t s, int out_h1 = 0, const float* out_prev = nullptr,
                                  unsigned char* idx = nullptr);
hipError_t launch_preprocess(const uint8_t* f, long long npix, float m0, float m1, float m2,
                             float* out, hipStream_t s);
hipError_t launch_motion_energy(const uint8_t* frames, long long frame_bytes, int n_frames, const uint8_t* prev_frame,
                                unsigned long long* sums, hipStream_t s);
constexpr size_t SOFT_ARGMAX_LDS_LIMIT = 150 * 1024;      // bytes of one joint's map the LDS variant holds (38 400 cells); larger maps stream
hipError_t launch_soft_argmax(const float* scmap, int B, int H, int W, int C, float gamma,
                              int gauss_len, float* mu, float* conf, int* idx, float* pmap,
                              hipStream_t s, int record_stride = 0);
hipError_t launch_hard_argmax(const float* scmap, const float* locref, int B, int H, int W, int C,
                              int* idx, float* prob, float* offs, hipStream_t s);
hipError_t launch_pmap_threshold(float* pmap, int B, int H, int W, int C, float th, float* mu, hipStream_t s);

// Arguments of the DGP loss forward+backward kernels (dgp_loss.hip).
struct LossArgs {
    const float* pred;          // [nt,H,W,nj]
    const float* locref_pred;   // [nt,H,W,2nj]
    const float* mu;            // [nt*nj,2] soft-argmax of pred
    const float* targets;       // [n_vis_frames*nj,2] labels (NaN -> 0), scoremap units (row, col)
    const float* locref_map;    // [nt,H,W,2nj]
    const float* locref_mask;   // [nt,H,W,2nj]
    const int* visible_marker; const int* hidden_marker; const int* visible_in_targets;
    const float* S0; const float* ws; const float* ws_max;     // [nl,nj], [nl], [nl]
    float* dpred; float* dlocref;
    float* losses;              // [8]: visible, hidden, locref, ws, total, total_visible
    float* t_all; float* dLdt; int* kind; float* stats; float* norm;   // scratch
    int nt, H, W, nj, nl, n_v, n_h;
    int gm2, gm3, gauss_len, huber;
    float gamma, lengthscale, stride, hidden_scale, clique_scale, locref_weight;
    // temporal clique (wt > 0)
    const float* vector_field; const float* wt_batch; float* wt_w;
    int use_wt, Hin, Win;
    float wt_max, temporal_scale;
};
constexpr size_t LOSS_LDS_LIMIT = 150 * 1024;      // bytes of the TWO per-marker maps loss_ce_backward<false> keeps in LDS (19 200 cells); larger maps stream
hipError_t launch_loss(const LossArgs& a, hipStream_t s);

struct DlcLossArgs {                      // DLC step-0 loss (sigmoid CE on binary disks + locref Huber)
    const float *pred, *part_targets, *part_weights;          // part_weights may be null (all ones)
    const float *locref_pred, *locref_targets, *locref_mask;  // locref_pred null = no location refinement
    float *dpred, *dlocref, *losses;
    double* acc;
    long long n_part, n_loc;
    float locref_loss_weight;
    int huber;
};
hipError_t launch_dlc_loss(const DlcLossArgs& a, hipStream_t s);

// Arguments of the bottleneck chain kernel (dgp_chain.hip): conv3 of unit k (+ shortcut, ReLU) -> X' -> conv1 of unit k + 1 -> R1',
// all tensors H2.  Weight chunks: per 32 channels of X' the fp16 fragment pairs of conv3 (those columns), of conv1 (those K rows),
// and one fragment with the BN affine of those channels (32 scales | 32 biases).
struct ChainArgs {
    const void* r2;           // H2 [M][C], cells hold R2 * 2^e
    const void* src2;         // residual X (H2, C4 channels; res 1: [M], res 2: [N, res_H, res_W]) or K-concatenated source [M][CIN2]
    void* xout;               // H2 [M][C4]
    void* r1out;              // H2 [M][C1]
    const void* wfrag;        // C4 / 32 chunks of chain_frags_per_chunk() KiB
    const float* sc1;         // [C1] conv1's BN scale
    const float* bi1;         // [C1]
    float post1, post2;       // exact powers of two that undo the operand scales: 1 / (R2 scale x conv3 weight scale), 1 / (X' scale x conv1 weight scale)
    float res_inv_scale, xout_scale, r1_scale;
    float* xout_absmax; float* r1_absmax;
    int M, HoWo, Wo, res_H, res_W;
    int ntiles;
    unsigned r2_bytes, src2_bytes, xout_bytes, r1_bytes, w_bytes;
    // unit kernel (conv2 in front of the chain; r2 is not used: R2 stays in registers)
    const void* r1in;         // H2 [N, H, W, C]: conv2's input (the unit's conv1 output)
    unsigned r1in_bytes;
    int H, W, TY, TX;         // frame size (stride 1: output = input grid), tiles per frame
    float post0, r2_scale;    // 1 / (R1 scale x conv2 weight scale); scale R2's fragments are split with
    float* r2_absmax;
    int nt;                   // bit 0: residual loads, bit 1: X' stores with the non-temporal (evict-first) policy
    int h1;                   // 1: every tensor is an H1 tensor (the 16-bit tier): 2 bytes per channel, high weight fragments only, one MFMA per product
};
bool chain_supported(int C, int C1, int CIN2, int res);          // res: 0 K-concatenated shortcut, 1 identity, 2 subsample of a stride-2 unit
int  chain_frags_per_chunk(int C, int C1, int CIN2, int planes = 2);
hipError_t launch_chain(const ChainArgs& a, int C, int C1, int CIN2, int res, hipStream_t s);
const char* chain_kernel_name(int C, int C1, int CIN2, int res);
// the unit kernel: conv2 (3x3, stride 1) in front of the chain.  Weight chunks: 9 conv2 chunks (one tap each: (C / 32) (C / 16) fragment
// pairs + the affine fragment) followed by the chain's chunks.  a.H / a.W / a.r1in describe conv2's input; N frames
bool unit_supported(int C, int C1, int CIN2, int res);
hipError_t launch_unit(const ChainArgs& a, int N, int C, int C1, int CIN2, int res, hipStream_t s);
// fp32 fragments -> fp16 high / low fragment pairs.  src per chunk: (f1_pairs + f2_pairs) x [64 lanes][8 floats], then the affine
// fragment [64 floats... 256 floats]; out per chunk: 2 (f1_pairs + f2_pairs) + 1 KiB-fragments
// planes: 2 = [hi][lo] pairs; 1 = high fragments only (the 16-bit tier's chunks: f1_pairs + f2_pairs + 1 KiB-fragments per chunk)
hipError_t launch_chain_pack(const float* src, int n_chunks, int f1_pairs, int f2_pairs, float s1, float s2, void* out, hipStream_t s, int planes = 2);

// Run-time switches of the native code, two classes (README "Switches"):
//  * dgp_env(): the shipped ones -- the precision tiers (DGP_CONV_MODE, DGP_H2) and the A/B switches between paths that all ship
//    and that the tests exercise (<= 20 in all); read from the environment once per process;
//  * dgp_tune(): knobs of measured-and-settled choices and the opt-ins that measured SLOWER (the 256-row tile DGP_TALL, the C = 256
//    chain instance DGP_CHAIN_WIDE, the trainer's fast pass, the non-pipelined weight-gradient tile).  The product binary compiles
//    the default in; they read the environment only in a tuning build (DGP_BUILD_FLAGS=-DDGP_TUNING python -m deepgraphpose_amd.build),
//    which is also the only build that carries the kernels behind the opt-ins.
inline int dgp_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
inline int dgp_tune(const char* name, int dflt) {
#ifdef DGP_TUNING
    return dgp_env(name, dflt);
#else
    (void)name;
    return dflt;
#endif
}

// hipFuncSetAttribute applies to the CURRENT device: the "done once" flags of the launchers are kept per device
inline int dgp_device_slot() { int d = 0; (void)hipGetDevice(&d); return d & 15; }
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

}  // namespace dgp
