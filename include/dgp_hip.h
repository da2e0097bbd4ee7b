/*
 * dgp_hip.h -- C-ABI of libdgp_hip.so, the MI355X (gfx950) engine for the Deep Graph
 * Pose hot path.  Plain pointers and sizes only; every device pointer is memory the
 * CALLER owns (e.g. torch tensors), `stream` is a hipStream_t passed as void*.
 *
 * The reference (paninski-lab/deepgraphpose) has no FFI: its boundary for this path is
 * one TF-1.x session call.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root; PET = src/DeepLabCut/deeplabcut/
 * pose_estimation_tensorflow, DGP = src/deepgraphpose):
 *
 *   dgp_forward        sess.run(scmap[,locref], {inputs: frames})   DGP/models/eval.py:328
 *                      graph: PET/nnet/pose_net.py:36-54 (extract_features) +
 *                             DGP/models/fitdgp_util.py:18-74 (dgp_prediction_layer)
 *   dgp_soft_argmax    argmax_2d_from_cm                            DGP/models/fitdgp_util.py:342-402
 *                      + likelihood read-out                        DGP/models/eval.py:331-343
 *   dgp_hard_argmax    argmax_pose_predict                          PET/nnet/predict.py:62-77
 *   dgp_infer          sess.run([mu_n, scmap]) + read-out           DGP/models/eval.py:306-345
 *   dgp_net_create /   setup_dgp_eval_graph (graph build + Saver.restore)
 *   dgp_net_load_weights                                            DGP/models/eval.py:147-214
 *
 * All functions return 0 on success or a negative dgp_status; dgp_last_error() gives a
 * thread-local message.  Launches are asynchronous on `stream`.  Functions that synchronise say so at their
 * declaration; the ones on the hot path are: the FIRST dgp_forward / dgp_infer after dgp_net_load_weights,
 * dgp_net_set_tier, dgp_net_recalibrate or a reported range overflow (the calibration pass reads every
 * layer's tracked range: one stream sync + 1 KB copy per layer, see "H2" below), dgp_net_range_status,
 * dgp_net_load_weights and the layer-level test entry points dgp_chain_h2 / dgp_unit_h2.  Every other
 * forward / infer call enqueues and returns.
 * A dgp_net handle is bound to the device current at creation and is not thread-safe.
 */
#ifndef DGP_HIP_H
#define DGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGP_ABI_VERSION 1

typedef enum dgp_status {
    DGP_OK = 0,
    DGP_ERR_INVALID = -1,     /* bad argument / shape */
    DGP_ERR_HIP = -2,         /* HIP runtime error (see dgp_last_error) */
    DGP_ERR_MISSING = -3,     /* a required weight tensor was not supplied */
    DGP_ERR_STATE = -4        /* e.g. forward before load_weights */
} dgp_status;

typedef struct dgp_net dgp_net;   /* opaque */

typedef struct dgp_net_desc {
    int32_t depth;        /* 50 | 101 | 152 (net_type resnet_50 / resnet_101, PET/nnet/pose_net.py:14-16) */
    int32_t num_joints;   /* cfg.num_joints */
    int32_t in_h, in_w;   /* frame size */
    int32_t max_batch;
    int32_t with_locref;  /* build pose/locref_pred head too (cfg.location_refinement) */
    float   mean_pixel[3];/* cfg.mean_pixel, RGB (PET/default_config.py:23) */
    float   bn_eps;       /* slim resnet_arg_scope epsilon, 1e-5 */
} dgp_net_desc;

/* One named weight tensor in the TF variable layout (HWIO convs, [kh,kw,Cout,Cin]
 * transposed convs, [C] vectors), fp32, host memory. */
typedef struct dgp_tensor_view {
    const char*  name;    /* TF variable name, e.g. "resnet_v1_50/conv1/weights" */
    const float* data;
    int32_t      ndim;
    int64_t      shape[4];
} dgp_tensor_view;

int          dgp_version(void);
/* 1 when the library was built with -DDGP_TUNING (tuning knobs read from the environment, the opt-in kernels that measured slower
 * compiled in), 0 for the product build.  No reference counterpart (build introspection for the tests). */
int          dgp_tuning_build(void);
/* Host-only helper: CRC-32C (Castagnoli) of a byte range, continuing from `crc` (0 to start).  Used for the
 * per-tensor checksums of TF tensor bundles that tf.train.Saver writes/verifies (DGP/models/fitdgp.py:149-171). */
uint32_t     dgp_crc32c(const void* data, size_t n, uint32_t crc);
const char*  dgp_last_error(void);

int  dgp_net_create(const dgp_net_desc* desc, dgp_net** out);
void dgp_net_destroy(dgp_net* net);
/* Copies + repacks (k-chunked weight panels, BN folded to scale/bias).  Synchronous. */
int  dgp_net_load_weights(dgp_net* net, const dgp_tensor_view* tensors, int32_t n);
int  dgp_net_workspace_bytes(const dgp_net* net, int32_t batch, size_t* out_bytes);
/* Output geometry of the heads: scoremap is [batch, out_h, out_w, num_joints]. */
int  dgp_net_output_dims(const dgp_net* net, int32_t* out_h, int32_t* out_w,
                         int32_t* feat_h, int32_t* feat_w);
/* Number of conv-stack kernel launches of one forward and their algorithmic FLOPs. */
int  dgp_net_stats(const dgp_net* net, int32_t batch, int32_t* n_launches, double* conv_flops);

/* frames: device uint8 [batch, in_h, in_w, 3] NHWC (RGB).
 * scmap : device fp32 [batch, out_h, out_w, nj]      (raw logits, pose/part_pred)
 * locref: device fp32 [batch, out_h, out_w, 2*nj] or NULL
 * features: optional device fp32 [batch, feat_h, feat_w, 2048] copy-out, or NULL */
int  dgp_forward(dgp_net* net, const uint8_t* frames, int32_t batch, void* workspace,
                 size_t workspace_bytes, float* scmap, float* locref, float* features,
                 void* stream);

/* DGP soft-argmax + likelihood window.
 * scmap [B,H,W,C] fp32 -> mu [B,C,2] (row, col) fp32, conf [B,C] fp32 (sigmoid of the raw
 * logit at the window arg-max), idx [B,C,2] int32 (row, col), pmap optional [B,H,W,C].
 * Any H x W (the reference's placeholders are [None, None, None, nj]): maps up to 38 400 cells are held in LDS, larger ones
 * stream from global memory with the same arithmetic. */
int  dgp_soft_argmax(const float* scmap, int32_t B, int32_t H, int32_t W, int32_t C,
                     float gamma, int32_t gauss_len, float* mu, float* conf, int32_t* idx,
                     float* pmap, void* stream);

/* argmax_2d_from_cm's `th` branch (DGP/models/fitdgp_util.py:377-388; unused by the reference's drivers): on the pmap that
 * dgp_soft_argmax wrote, per (frame, joint) map: values below th * max become 0, the map is renormalised IN PLACE and
 * mu [B,C,2] (row, col) is its expectation. */
int  dgp_pmap_threshold(float* pmap, int32_t B, int32_t H, int32_t W, int32_t C, float th, float* mu, void* stream);

/* DLC hard arg-max over sigmoid(scmap).  idx [B,C,2] (row, col), prob [B,C],
 * offs [B,C,2] = locref[b,row,col,2c..2c+1] (dx, dy; NOT yet scaled by locref_stdev) or
 * zeros when locref == NULL. */
int  dgp_hard_argmax(const float* scmap, const float* locref, int32_t B, int32_t H, int32_t W,
                     int32_t C, int32_t* idx, float* prob, float* offs, void* stream);

/* Fused A1..A4: frames -> (mu, conf, idx); scoremap stays in the workspace unless
 * scmap_out != NULL. */
int  dgp_infer(dgp_net* net, const uint8_t* frames, int32_t batch, void* workspace,
               size_t workspace_bytes, float gamma, int32_t gauss_len, float* mu, float* conf,
               int32_t* idx, float* scmap_out, void* stream);

/* Same, written as ONE packed record per (frame, joint): traj [batch, nj, 5] fp32 lanes = (row, col, likelihood, iy, ix) with
 * the two window indices as int32 bit patterns in lanes 3..4.  This is the layout of the per-video trajectory that the
 * frame-sharded ranks exchange with a single RCCL all-gather; writing it here keeps torch's cat / copy
 * kernels out of the per-batch loop. */
int  dgp_infer_packed(dgp_net* net, const uint8_t* frames, int32_t batch, void* workspace,
                      size_t workspace_bytes, float gamma, int32_t gauss_len, float* traj,
                      float* scmap_out, void* stream);

/* ---- measurement: hipEvent pairs around every launch of dgp_forward / dgp_infer, recorded on
 * the caller's stream (no syncs until dgp_net_profile_launch reads them).  Used by bench.py for
 * the roofline object; the reference's only timing is time.time() around sess.run
 * (DGP/models/fitdgp.py:817-828). */
/* Change the frame size of an existing net (weights stay).  DLC's step-0 loader feeds a differently scaled /
 * cropped image every iteration (pose_defaultdataset.py:131-196: placeholders [1, None, None, 3]); workspaces
 * are sized per call from the current geometry. */
int  dgp_net_set_input_size(dgp_net* net, int32_t in_h, int32_t in_w);

int  dgp_net_profile_begin(dgp_net* net, int32_t max_steps);
int  dgp_net_profile_end(dgp_net* net, int32_t* n_steps, int32_t* n_launches);
/* Average duration (ms, over the profiled steps), name and algorithmic FLOPs of launch i. */
int  dgp_net_profile_launch(dgp_net* net, int32_t launch, char* name, int32_t name_cap,
                            double* flops, double* avg_ms);

/* ---- DGP loss, forward + backward w.r.t. the head outputs (dgp_loss in DGP/models/fitdgp.py:848-1144).
 * Replaces the loss sub-graph evaluated inside sess.run([loss, train_op]) (fitdgp.py:818).
 * Marker id = frame_in_batch * nj + joint (DGP/dataset.py:187-239). */
typedef struct dgp_loss_desc {
    int32_t nt, H, W, nj, nl;             /* frames in batch, scoremap size, joints, limbs (rows of S0) */
    int32_t n_visible, n_hidden;          /* lengths of visible_marker / hidden_marker */
    int32_t gm2, gm3;                     /* hidden-loss variants: gm2 in {0,1,2}, gm3 in {0,3} (fitdgp.py:994-1034) */
    int32_t gauss_len, huber;             /* soft-argmax blur length; 1 = Huber locref loss, 0 = MSE */
    float   gamma, lengthscale, stride;   /* dgp_cfg.gamma, .lengthscale, .stride (fitdgp.py:645-647) */
    float   wn_visible, wn_hidden, locref_loss_weight;
    float   n_frames_total, n_visible_frames_total;   /* data_batcher totals (fitdgp.py:869-873) */
    int32_t use_wt, Hin, Win;             /* temporal clique (dgp_cfg.wt > 0): flow field size [nt-1, Hin, Win] */
    float   wt_max;                       /* dgp_cfg.wt_max */
} dgp_loss_desc;

int  dgp_loss_scratch_bytes(const dgp_loss_desc* d, size_t* out_bytes);
/* All pointers are device memory.  pred [nt,H,W,nj], locref_pred / locref_map / locref_mask [nt,H,W,2nj],
 * targets [n_vis_frames*nj, 2] labels in scoremap (row, col) units with NaN already replaced by 0,
 * visible_in_targets indexes rows of `targets`; S0 [nl,nj], ws / ws_max [nl].
 * Outputs: dpred, dlocref (same shapes as the predictions), mu [nt*nj,2],
 * losses[8] = {visible_loss_pred, hidden_loss_pred, visible_loss_locref, ws_loss, total_loss, total_loss_visible,
 *             wt_loss, 0}.
 * Temporal clique (use_wt; fitdgp.py:1079-1124): wt_loss = || (relu(D - wt_max) + wt_max) * w ||_F * scale with D the pixel distance of
 * a marker between consecutive frames and w the flow weight min(1 / mean flow, 1)^3 * wt_batch / H / W over the +-10 px box of the two
 * positions (bilinear crop_and_resize mean of vector_field).  The backward pass differentiates D AND w: the boxes are functions of the
 * (hidden, soft-arg-max) targets and TF's crop_and_resize has a gradient with respect to its boxes -- the weight is not a stop-gradient.
 * Any H x W (the reference's placeholders are [None, None, None, nj], fitdgp.py:1130-1142): maps up to 19 200 cells keep their Gaussian
 * target and sigmoid in LDS, larger ones recompute them where they are read -- same functions, bit-identical results. */
int  dgp_loss_fwd_bwd(const dgp_loss_desc* d, const float* pred, const float* locref_pred, const float* targets,
                      const float* locref_map, const float* locref_mask, const int32_t* visible_marker,
                      const int32_t* hidden_marker, const int32_t* visible_in_targets, const float* S0,
                      const float* ws, const float* ws_max, const float* vector_field /* [nt-1,Hin,Win] or NULL */,
                      const float* wt_batch /* [nt-1] = wt * batch_mask, or NULL */, float* dpred, float* dlocref,
                      float* mu, float* losses, void* scratch, size_t scratch_bytes, void* stream);

/* ---- DLC step-0 loss (fit_dlc: DGP/models/fitdgp.py:124; pose_net.train in DeepLabCut
 * pose_estimation_tensorflow/nnet/pose_net.py:159-190, huber_loss in nnet/losses.py:16-45):
 *   part_loss   = sum(w * sigmoid_ce(pred, part_targets)) / #nonzero(w)     (w = part_weights, NULL = all ones)
 *   locref_loss = locref_loss_weight * sum(mask * huber(locref_pred - locref_targets)) / #nonzero(mask)
 * All pointers device memory; locref_pred NULL = location_refinement off.  pred/part_* [nt,H,W,nj],
 * locref_* [nt,H,W,2nj].  Writes dpred, dlocref and losses[4] = {part_loss, locref_loss, total_loss, 0}.
 * scratch: >= 32 bytes, 8-byte aligned. */
int  dgp_dlc_loss_fwd_bwd(const float* pred, const float* locref_pred, const float* part_targets,
                          const float* part_weights, const float* locref_targets, const float* locref_mask, int32_t nt,
                          int32_t H, int32_t W, int32_t nj, float locref_loss_weight, int32_t huber, float* dpred,
                          float* dlocref, float* losses, void* scratch, size_t scratch_bytes, void* stream);

/* ---- training step (config 4): replaces sess.run([loss, train_op]) of fit_dgp / fit_dgp_labeledonly
 * (DGP/models/fitdgp.py:708-713,818; 416-418,501-505).  The trainer owns the master parameters (flat fp32 buffer
 * of every trainable TF variable: conv weights, BN gamma/beta, head weights/biases), their gradients, the momentum
 * slots and the frozen BN statistics; activations live in the caller's workspace.
 *   dgp_trainer_sync_weights : master -> forward panels, folded BN, data-gradient panels (call after every update)
 *   dgp_train_forward        : forward pass retaining activations; returns device pointers of both head outputs
 *   dgp_loss_fwd_bwd         : (above) loss and d loss / d heads
 *   dgp_train_backward       : gradients of all trainables into the flat grads buffer
 *   dgp_sgd_momentum_clip    : tf.clip_by_global_norm(clip) + MomentumOptimizer(momentum) */
typedef struct dgp_trainer dgp_trainer;
int    dgp_trainer_create(dgp_net* net, dgp_trainer** out);            /* net must have with_locref = 1 */
void   dgp_trainer_destroy(dgp_trainer* tr);
int    dgp_trainer_num_tensors(const dgp_trainer* tr, int32_t* n_tensors, int64_t* n_trainable_floats,
                               int64_t* n_stat_floats);
int    dgp_trainer_tensor_info(const dgp_trainer* tr, int32_t i, char* name, int32_t cap, int64_t* offset,
                               int64_t* size, int32_t* is_stat);
float* dgp_trainer_buffer(dgp_trainer* tr, int32_t which);  /* 0 params, 1 grads, 2 momentum, 3 BN statistics; 4: the 16-bit pass's failure flag
                                                              * (ONE int32 on the device; data-parallel runs all-reduce it with MAX before the update) */
int    dgp_trainer_upload(dgp_trainer* tr, int32_t which, int64_t offset, const float* host, int64_t n);
int    dgp_trainer_download(dgp_trainer* tr, int32_t which, int64_t offset, float* host, int64_t n);
int    dgp_trainer_workspace_bytes(const dgp_trainer* tr, int32_t nt, size_t* out_bytes);
int    dgp_trainer_sync_weights(dgp_trainer* tr, void* stream);
int    dgp_train_forward(dgp_trainer* tr, const uint8_t* frames, int32_t nt, void* workspace, size_t workspace_bytes,
                         float** scmap, float** locref, void* stream);
int    dgp_train_backward(dgp_trainer* tr, int32_t nt, void* workspace, size_t workspace_bytes, const float* dscmap,
                          const float* dlocref, void* stream);
/* (gnorm_host_or_null != NULL synchronises the stream.  After a fast / 16-bit pass the update is conditional ON THE DEVICE: a pass
 * that raised the range flag leaves parameters and momentum untouched -- see dgp_trainer_step_status.) */
int    dgp_sgd_momentum_clip(dgp_trainer* tr, float lr, float momentum, float clip_norm, float* gnorm_host_or_null,
                             void* stream);
/* Fast pass of the training step (no reference counterpart; the step's results are the same fp32-class numbers either way).
 * dgp_trainer_fast_mode(tr, 1): the NEXT dgp_train_forward keeps the retained activations of blocks 2-4 as H2 tensors (fp16 high / low
 * cells, no fp32 twin) whose power-of-two scales are PREDICTED from the ranges the same tensors had in the previous pass (max -> [2^10,
 * 2^11): five bits of headroom), so its convs run the inference engine's cell kernels and the weight gradients read the activations in
 * place by LDS-DMA; dgp_train_backward follows what the forward did.  Ask for it only after a pass with the same frame count and
 * input size (plain or fast) -- never for the first pass after dgp_trainer_create, a weight upload or a shape change.
 * dgp_trainer_fast_status (synchronises the device): *failed != 0 -- a tensor left its predicted range (a jump of more than 2^5 up or
 * 2^7 down within one step); scoremaps, losses and gradients of that step are NOT valid: run it again with fast_mode(tr, 0) before
 * using any of them.  deepgraphpose_amd/train.py (Trainer.forward_backward) does exactly that. */
/* Precision tier of the training step (BASELINE configs[3] names bf16).  0 (default): parity -- fp32-class arithmetic, fp32 retained
 * activations.  1: the 16-bit tier -- after dgp_trainer_sync_weights, every pass requested with dgp_trainer_fast_mode(tr, 1) keeps the
 * retained activations AND the gradient tensors of blocks 2-4 as H1 cells (2 bytes per channel, scales predicted from the previous
 * step's ranges, no fp32 twins), runs their forward / data-gradient convs on one MFMA per product and reads both operands of their
 * weight gradients in place by LDS-DMA; fp32 master weights, momentum, gradient accumulation and loss; block1, the root block and the
 * heads as in tier 0.  Same protocol as the fast pass: never the first pass of a shape; dgp_trainer_fast_status reports a step whose
 * tensors left their predicted ranges (its results are invalid: run it again with fast_mode(tr, 0)).  Call dgp_trainer_sync_weights
 * after changing the tier.  A reported tier with measured gradient error, not the parity claim. */
int    dgp_trainer_set_tier(dgp_trainer* tr, int32_t tier);
int    dgp_trainer_get_tier(const dgp_trainer* tr);
/* Data-parallel training (the reference trains on one GPU; DGP/models/fitdgp.py:708-713 is the update every replica applies): the
 * gradient all-reduce of a GROUP of layers can start while dgp_train_backward is still computing the earlier layers.  Group k = floats
 * [lo[k], hi[k]) of the flat gradient buffer (dgp_trainer_buffer(tr, 1)), in the order the LAST backward pass completes them (heads and
 * last bottleneck units first, stem last; the groups tile the buffer; *n_groups = 0 before the first pass).
 * dgp_trainer_grad_group_wait makes `stream` wait for group k (hipStreamWaitEvent: nothing blocks on the host). */
int    dgp_trainer_grad_groups(dgp_trainer* tr, int32_t max_groups, int32_t* n_groups, int64_t* lo, int64_t* hi);
int    dgp_trainer_grad_group_wait(dgp_trainer* tr, int32_t k, void* stream);
int    dgp_trainer_fast_mode(dgp_trainer* tr, int32_t enable);
int    dgp_trainer_fast_status(dgp_trainer* tr, int32_t* was_fast, int32_t* failed);
/* ONE synchronisation per training step: enqueue forward, loss, backward, dgp_sgd_momentum_clip(..., NULL gnorm, ...) and
 * dgp_trainer_sync_weights on `stream` without reading anything back, then call this -- a last tiny kernel writes the n_losses (<= 8)
 * device floats at d_losses (dgp_loss_fwd_bwd's `losses`; NULL / 0: none), the gradient norm the optimiser saw and the pass status into
 * pinned host memory, and the call waits for the stream once.  After a fast / 16-bit pass whose tensors left their predicted ranges
 * (*failed != 0) the momentum kernel has SKIPPED its update on the device (it reads the flag), so the step can be run again on the
 * parity path as if nothing had happened.  Replaces the read-backs of sess.run([loss, train_op]) (DGP/models/fitdgp.py:818). */
int    dgp_trainer_step_status(dgp_trainer* tr, const float* d_losses, int32_t n_losses, float* losses, float* gnorm,
                               int32_t* was_fast, int32_t* failed, void* stream);

/* ---- single-layer entry points (used by the parity tests and by fit_dgp later) ---- */

/* slim.conv2d / conv2d_same semantics on NHWC fp32 with HWIO weights supplied packed by
 * dgp_pack_conv_weights.  y = relu?( conv(x) * scale + bias (+ residual) ). */
typedef struct dgp_conv_desc {
    int32_t N, H, W, Cin;          /* input  (Cin % 4 == 0) */
    int32_t Cout;                  /* real output channels */
    int32_t KH, KW, stride, rate;
    int32_t pad_t, pad_l;          /* zero padding before (TF SAME / conv2d_same) */
    int32_t Ho, Wo;                /* output spatial size */
    int32_t relu;                  /* 1: apply ReLU */
    int32_t res_stride;            /* 0: no residual; s>=1: residual[n, ho*s, wo*s, :] */
    int32_t res_H, res_W;          /* residual spatial size */
} dgp_conv_desc;

size_t dgp_packed_weight_floats(int32_t KH, int32_t KW, int32_t Cin, int32_t Cout);
/* host -> host repack: HWIO [KH,KW,Cin,Cout] -> k-chunked panels [Kp/4][CoutP][4] */
int  dgp_pack_conv_weights(const float* hwio, int32_t KH, int32_t KW, int32_t Cin,
                           int32_t Cout, float* packed);
int  dgp_conv2d(const dgp_conv_desc* d, const float* x, const float* packed_w,
                const float* scale /*[Cout] or NULL*/, const float* bias /*[Cout] or NULL*/,
                const float* residual, float* y, void* stream);
/* Same, with the operand ranges the fp16-split kernels need.  A range is a device array of DGP_ABSMAX_SLOTS floats
 * whose maximum is (an upper bound of) max |tensor| (producers spread their atomics over the slots); fill one with
 * dgp_tensor_absmax.  With x_absmax and w_absmax present the fp16 high/low split (3 MFMAs per fp32-class product)
 * may be used, otherwise the bf16 3-way split (6 MFMAs, no range requirement).  y_absmax (zero it first; may be
 * NULL) receives max |y| so that y can feed the next ranged call. */
#define DGP_ABSMAX_SLOTS 256
int  dgp_conv2d_ranged(const dgp_conv_desc* d, const float* x, const float* packed_w, const float* scale, const float* bias,
                       const float* residual, float* y, const float* x_absmax, const float* w_absmax, float* y_absmax,
                       void* stream);
/* ---- single-layer BACKWARD entry points: the launchers dgp_train_backward uses, exposed so that the weight-gradient and
 * data-gradient kernels can be checked layer by layer at the real shapes (no reference counterpart: TF differentiates the graph,
 * DGP/models/fitdgp.py:708-713).  `d` always describes the FORWARD conv x [N,H,W,Cin] -> y [N,Ho,Wo,Cout].
 *   dgp_conv2d_wgrad : dw_raw[(kh, kw, ci)][co] = sum over output pixels of x[...] * dy[...] (HWIO order, overwritten) and
 *                      colsum[co] = sum dy[.., co] (2 * Cout floats, second half scratch; may be NULL).  With both operand ranges
 *                      (DGP_ABSMAX_SLOTS floats each) and a tile of 128 x 128 the fp16-split kernel runs, else fp32 MFMA.
 *   dgp_conv2d_dgrad : dx = gate(convT(dy; w * scale) + dx_add); w_hwio device HWIO weights; scale [Cout] or NULL; mask [N,H,W,Cin]
 *                      or NULL (gate: mask > 0); dx_add NULL, or a gradient on dx's grid (add_mode 1) or on the 2x coarser grid
 *                      (add_mode -2, the subsample shortcut).  scratch: dgp_conv2d_dgrad_scratch_bytes(d) device bytes.
 *                      Cout % 32 == 0, Cin % 4 == 0; ranged bit 1 (the gate is an H2 tensor): bit 0, a mask, Cin % 8 == 0 and Cin >= 64. */
int    dgp_conv2d_wgrad(const dgp_conv_desc* d, const float* x, const float* dy, const float* x_absmax, const float* dy_absmax,
                        float* dw_raw, float* colsum, void* stream);
/* dgp_conv2d_wgrad on the LDS-DMA tile of the training step (csrc/dgp_train.hip, wgrad_dma).  Inside dgp_train_backward both operands
 * of a 128 x 128 weight-gradient tile also exist as fp16 high / low cells (the H2 layout below), written by their producers' epilogues
 * with a power-of-two scale PREDICTED from the range the same tensor had one step earlier (max -> [2^10, 2^11)); the kernel moves the
 * cells by LDS-DMA (no split arithmetic, no register staging) and checks this step's ranges against the predicted scales: outside
 * [2^4, 65000) after scaling -- first step, a jump of more than 2^5 -- the workgroup computes its tile from the fp32 tensors on the
 * fp32 matrix pipe instead.  This entry point makes the copies itself from x / dy and the given "previous" ranges, so that the tile
 * and both of its paths can be checked layer by layer.  scratch: 4 * (numel(x) + numel(dy)) device bytes; x_absmax / dy_absmax /
 * x_prev / dy_prev: DGP_ABSMAX_SLOTS floats each; Cin = 16 * 2^k, Cout % 8 == 0, KH * KW * Cin >= 128, Cout >= 128. */
int    dgp_conv2d_wgrad_shadow(const dgp_conv_desc* d, const float* x, const float* dy, const float* x_absmax, const float* dy_absmax,
                               const float* x_prev, const float* dy_prev, void* scratch, float* dw_raw, float* colsum, void* stream);
size_t dgp_conv2d_dgrad_scratch_bytes(const dgp_conv_desc* d);
int    dgp_conv2d_dgrad(const dgp_conv_desc* d, const float* dy, const float* w_hwio, const float* scale, const float* mask,
                        const float* dx_add, int32_t add_mode, float* dx, void* scratch, int32_t ranged, void* stream);
/* ---- "H2" activation format.  Inside dgp_forward / dgp_infer every tensor from the pool output to the block4 features lives
 * in HBM as fp16 high / low cell pairs: per pixel and 8 channels [8 halves hi | 8 halves lo] = 32 bytes (the footprint and the
 * addresses of 8 fp32 channels) holding x * 2^exp, hi = fp16(x 2^exp), lo = fp16(x 2^exp - hi): 22 significant bits.  The
 * producing conv splits once in its epilogue; the consuming conv's K loop copies the cells into the MFMA operand image and only
 * issues ds_read + MFMA.  exp is per tensor, calibrated by the FIRST forward after dgp_net_load_weights (layer by layer, with
 * hidden stream syncs: max |tensor| * 2^exp lands in [2^10, 2^11), 4 bits under the fp16 limit) and frozen afterwards, so the
 * results are a deterministic function of the frames and the scales.  Every forward checks the tracked ranges against the scales
 * on the device; dgp_net_range_status reports (and clears) an overflow -- the results of a forward that overflowed are invalid,
 * the next forward re-calibrates on its own batch and the caller re-runs what it needs.  DGP_H2=0 keeps fp32 activations.
 * The entry points below expose the format for tests and for features handed out at the boundary. */
int  dgp_f32_to_h2(const float* x, size_t n_floats, int32_t scale_exp, void* out, void* stream);
int  dgp_h2_to_f32(const void* x, size_t n_floats, int32_t scale_exp, float* out, void* stream);
/* dgp_conv2d on H2 tensors: x (H2, exponent x_exp) -> y (H2 with y_exp, or fp32 when y_is_h2 == 0); residual fp32 or H2.
 * w_absmax: range slots of packed_w (dgp_tensor_absmax); cells_scratch: >= dgp_packed_weight_floats(...) * 4 device bytes.
 * fp32 output exists for 1x1 / stride-1 layers only (the heads' pointwise GEMM) and takes an fp32 residual; every other combination
 * with y_is_h2 == 0, and an H2 residual with an fp32 output, returns DGP_ERR_INVALID. */
int  dgp_conv2d_h2(const dgp_conv_desc* d, const void* x_h2, int32_t x_exp, const float* packed_w, const float* w_absmax,
                   const float* scale, const float* bias, const void* residual, int32_t res_is_h2, int32_t res_exp, void* y,
                   int32_t y_is_h2, int32_t y_exp, float* y_absmax, void* cells_scratch, void* stream);
/* ---- "H1" activation format: the 16-bit tier.  dgp_net_set_tier(net, 1) makes dgp_forward / dgp_infer keep every tensor from the pool
 * output to the block4 features as ONE 16-byte cell of 8 halves per pixel and 8 channels -- fp16(x * 2^exp), bit for bit the high cell
 * of the H2 pair: plain NHWC fp16 with the same calibrated per-tensor scales -- and multiply them with fp16 weights on one MFMA per
 * product (fp32 accumulation, fp32 epilogue arithmetic, fp32 heads and soft-argmax).  Half the activation bytes and a third of the
 * matrix work of the parity tier (tier 0, the default); 11-bit operands, so NOT inside the 1e-3 px gate: a reported tier with measured
 * error (bench.py `tier_f16`), never the parity claim.  No reference counterpart (the reference computes in fp32 on TF-1.x).
 * Switching tiers re-calibrates on the next forward.  DGP_CONV_MODE=f16 in the environment makes tier 1 the default of new nets. */
int  dgp_net_set_tier(dgp_net* net, int32_t tier);
int  dgp_net_get_tier(const dgp_net* net);
int  dgp_f32_to_h1(const float* x, size_t n_floats, int32_t scale_exp, void* out, void* stream);
int  dgp_h1_to_f32(const void* x, size_t n_floats, int32_t scale_exp, float* out, void* stream);
/* dgp_conv2d on H1 tensors (layer tests): x (H1, exponent x_exp; Cin % 64 == 0) -> y (H1 with y_exp, or fp32 when y_is_h1 == 0: 1x1 /
 * stride-1 layers only, no residual); residual_h1: an H1 tensor on y's grid (d->res_stride) or NULL.  cells_scratch:
 * >= dgp_packed_weight_floats(...) * 2 device bytes. */
int  dgp_conv2d_h1(const dgp_conv_desc* d, const void* x_h1, int32_t x_exp, const float* packed_w, const float* w_absmax,
                   const float* scale, const float* bias, const void* residual_h1, int32_t res_exp, void* y,
                   int32_t y_is_h1, int32_t y_exp, float* y_absmax, void* cells_scratch, void* stream);
/* The engine's chain kernel at layer level (tests): conv3 of a bottleneck unit + shortcut + ReLU (X', written once) and conv1 of the
 * NEXT unit (R1') in one launch; conv1 takes X' from the accumulator registers, so the 4C-wide tensor is not re-read.  Replaces two
 * slim.conv2d calls of consecutive `bottleneck` units (PET/nnet/pose_net.py:46-52 -> slim resnet_v1.bottleneck: conv3 without
 * activation, relu(shortcut + residual), then conv1 of the next unit).  All device tensors H2 with the given exponents; weights
 * (row-major [K][N], K = C (+ CIN2 rows of the BN-folded shortcut conv), N = 4 C; w1 [4 C][C1]) and BN affines are HOST arrays.
 * res_mode: 0 K-concatenated shortcut conv (src2 = its input [M][CIN2], same exponent as r2), 1 identity (src2 = X [M][4C]),
 * 2 subsample of a stride-2 unit (src2 = X [N, res_H, res_W, 4C], read at (2 ho, 2 wo)).  DGP_ERR_INVALID when no kernel instance
 * exists for the shape.  Synchronises the stream (per-call weight packing). */
int  dgp_chain_h2(int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode, int32_t res_H, int32_t res_W,
                  const void* r2_h2, int32_t r2_exp, const void* src2_h2, int32_t src2_exp,
                  const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                  void* xout_h2, int32_t xout_exp, void* r1_h2, int32_t r1_exp, float* xout_absmax, float* r1_absmax, void* stream);
/* The engine's unit kernel at layer level (tests; block1 shapes, C = 64): additionally conv2 (3x3, stride 1, SAME, BN, ReLU) of the
 * unit in front -- slim `bottleneck`'s conv2 -> conv3 -> add -> relu and the next unit's conv1 in ONE launch.  r1 [N, H, W, C] is
 * conv2's input (read with a one-pixel halo: r1out must be another buffer); R2 exists only in registers (r2_exp: the scale its fp16
 * fragments are split with; its range is tracked into r2_absmax).  w2: HWIO [3][3][C][C] host array.  Other arguments as dgp_chain_h2. */
int  dgp_unit_h2(int32_t N, int32_t H, int32_t W, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode,
                 const void* r1_h2, int32_t r1_exp, const void* src2_h2, int32_t src2_exp,
                 const float* w2, const float* scale2, const float* bias2, int32_t r2_exp,
                 const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                 void* xout_h2, int32_t xout_exp, void* r1out_h2, int32_t r1out_exp, float* r2_absmax, float* xout_absmax, float* r1_absmax,
                 void* stream);
/* dgp_chain_h2 / dgp_unit_h2 on H1 tensors (the 16-bit tier's instances of the same kernels: 2 bytes per channel, the chunks' high weight
 * fragments only, one MFMA per product; a REPORTED tier, see dgp_net_set_tier).  Same arguments, every device tensor H1. */
int  dgp_chain_h1(int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode, int32_t res_H, int32_t res_W,
                  const void* r2_h1, int32_t r2_exp, const void* src2_h1, int32_t src2_exp,
                  const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                  void* xout_h1, int32_t xout_exp, void* r1_h1, int32_t r1_exp, float* xout_absmax, float* r1_absmax, void* stream);
int  dgp_unit_h1(int32_t N, int32_t H, int32_t W, int32_t C, int32_t C1, int32_t CIN2, int32_t res_mode,
                 const void* r1_h1, int32_t r1_exp, const void* src2_h1, int32_t src2_exp,
                 const float* w2, const float* scale2, const float* bias2, int32_t r2_exp,
                 const float* w3cat, const float* scale3, const float* bias3, const float* w1, const float* scale1, const float* bias1,
                 void* xout_h1, int32_t xout_exp, void* r1out_h1, int32_t r1out_exp, float* r2_absmax, float* xout_absmax, float* r1_absmax,
                 void* stream);
/* Synchronises `stream`, then: *overflow = 1 if any forward since the last call outgrew a calibrated scale (the flag is cleared
 * and the net re-calibrates on its next forward); *calibrations = calibration passes run so far. */
int  dgp_net_range_status(dgp_net* net, int32_t* overflow, int32_t* calibrations, void* stream);
int  dgp_net_recalibrate(dgp_net* net);      /* force a calibration pass on the next forward */
/* What dgp_net_range_status does to the engine when it reports an overflow (re-calibrate on the next forward with 3 more bits of
 * headroom), for a rank that did not overflow itself but must follow one that did (sharded runs stay bit-identical). */
int  dgp_net_widen(dgp_net* net);
/* Back to the state right after dgp_net_load_weights as far as the activation scales go: default headroom, calibration pass on the next
 * forward.  For an engine that is kept between videos (models/eval.py, the session kept by setup_dgp_eval_graph): the next video's first
 * batch then sets the same scales -- and every later frame gets the same bits -- as on a freshly built engine. */
int  dgp_net_reset_scales(dgp_net* net);
/* The calibrated activation scales of `src` (same network, weights, tier and frame size) become `dst`'s: what `dst` would have found by
 * calibrating on the same batch itself -- the calibration is deterministic -- without its layer-by-layer pass.  For callers that keep
 * several engines in step (two batches in flight: one engine calibrates on the video's first batch, the others copy).  Synchronises
 * `stream` (the stream `dst` runs on). */
int  dgp_net_copy_scales(dgp_net* dst, const dgp_net* src, void* stream);

/* slots = max(slots, max |x[0..n)|) on the device (zero the DGP_ABSMAX_SLOTS floats before the first call). */
int  dgp_tensor_absmax(const float* x, size_t n, float* absmax_dev, void* stream);
int  dgp_maxpool_3x3s2_same(const float* x, int32_t N, int32_t H, int32_t W, int32_t C,
                            float* y, void* stream);
int  dgp_preprocess_u8(const uint8_t* frames, int64_t n_pixels, const float mean[3],
                       float* out_nhwc4, void* stream);
/* Motion energy of a uint8 frame sequence on the device (replaces the per-frame numpy loop of calculate_motion_energy,
 * DGP/dataset.py:29-43).  sums[t] = sum over the frame's bytes of (frames[t] - frames[t-1]) mod 256 -- the
 * reference subtracts uint8 arrays, so the difference wraps -- for t >= 1, and for t = 0 against prev_frame (the last frame of
 * the previous chunk; NULL: sums[0] = 0).  The reference's value is sums[t] / frame_bytes in float64 (exact).  frames:
 * [n_frames][frame_bytes] device bytes; sums: n_frames device uint64 (overwritten).  Integer sums, bit-exact, order-independent. */
int  dgp_motion_energy(const uint8_t* frames, int64_t frame_bytes, int32_t n_frames, const uint8_t* prev_frame,
                       uint64_t* sums, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGP_HIP_H */
