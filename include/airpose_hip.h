/* airpose_hip.h -- C ABI of libairpose_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the per-frame inference hot path of AirPose.  Nothing like it exists
 * upstream (the reference is pure Python); every entry point names the reference interface it
 * replaces.  Paths are relative to the reference checkout.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator in the
 *     Python host); the library owns only packed weights and a per-handle workspace;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no hidden sync, except
 *     that a call which has to grow the workspace synchronises the device once;
 *   - return value: 0 = ok, negative = AP_E* argument/state error, positive = hipError_t;
 *     ap_last_error() returns a thread-local description;
 *   - handles are not re-entrant: one in-flight call per handle (the Python shim takes a lock);
 *   - all tensors are dense row-major float32 unless stated.
 */
#ifndef AIRPOSE_HIP_H
#define AIRPOSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AP_OK 0
#define AP_EINVAL (-1)  /* bad argument */
#define AP_ESHAPE (-2)  /* tensor name / shape mismatch */
#define AP_ESTATE (-3)  /* handle not finalised, missing tensors */
#define AP_ENOMEM (-4)
#define AP_ERANGE (-5)  /* AP_PREC_F16: a stored activation left the fp16 range (non-finite trunk features) */

#define AP_PREC_FP32 0 /* fp32 storage, v_mfma_f32_16x16x4_f32: parity mode (1e-4 vs the CPU reference) */
#define AP_PREC_BF16 1 /* bf16 storage (v_mfma_f32_16x16x32_bf16), fp32 accumulate and epilogues: throughput mode with fp32's
                        * exponent range; 3.0e-4 against the reference's CPU path (8 significand bits through 53 convolutions) */
#define AP_PREC_BF16X2 2 /* split-bf16 storage: every value as hi = rne(x), lo = rne(x - hi) in bf16 (16 mantissa bits, fp32
                          * bytes), planar in groups of 8 channels (32 bytes = 8 hi | 8 lo); every product as
                          * hi*hi + hi*lo + lo*hi on the bf16 matrix pipe (three MFMAs per 8 K elements), fp32 accumulate:
                          * the fast parity mode (meets the 1e-4 bar at ~5x the fp32-MFMA rate) */
#define AP_PREC_F16 3 /* IEEE fp16 storage (v_mfma_f32_16x16x32_f16), fp32 accumulate and epilogues: the throughput kernels of
                       * AP_PREC_BF16 at the same MFMA rate with 11 instead of 8 significand bits.  Against the reference's CPU path:
                       * 2.6e-5 .. 6.2e-5 on the benchmark checkpoints (under the 1e-4 bar), 1.8e-3 on a checkpoint with a wider
                       * BatchNorm-statistics range -- the bar is a property of the CHECKPOINT, which ap_net_parity_probe measures
                       * on the GPU (airpose_amd: precision="auto" picks the fastest mode whose probe holds it).
                       * Range: |stored value| <= 65504.  ap_net_finalize refuses a checkpoint whose (BatchNorm-folded) weights
                       * leave that range; at run time EVERY epilogue folds the values it stores into a range sentinel (see
                       * ap_net_set_range_check below): an activation that overflows makes the handle report AP_ERANGE */

typedef struct ap_net ap_net;     /* ResNet-50 trunk + IEF regressor */
typedef struct ap_smplx ap_smplx; /* SMPL-X body model */

/* ABI number of this header: bumped whenever an exported signature changes or an entry point is added or removed
 * (8: round 6 -- ap_net_parity_probe, ap_net_range_peek / _mark_next / _slot, ap_regressor_feat_part / _step_local / _step_finish;
 *  7: ap_conv_pw_*, ap_block_img_*; 6: ap_set_pair_groups removed).  A binding built against another number must refuse to load
 * the library (airpose_amd/_native.py does). */
#define AP_ABI_VERSION 9
const char* ap_version(void);
int ap_abi_version(void);
const char* ap_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Network handle.  Replaces model_copenet.getcopenet()/copenet.__init__
 * (copenet/src/copenet/models/model_copenet.py:53-92, 229-239) and load_state_dict.
 * Tensors are handed over under the reference's own state_dict names ("conv1.weight",
 * "layer3.2.bn1.running_var", "fc1.bias", "init_pose", ...) as HOST float32 arrays in PyTorch layout
 * (conv OIHW, linear [out][in]); ap_net_finalize folds BN into a per-channel scale/shift, repacks
 * the weights K-contiguous NHWC in the handle's precision and uploads them.  Calling set_tensor +
 * finalize again re-packs (fine-tuned weights).
 * variant: 0 = copenet two-view (fc1 in = 2332), 1 = hmr single-view head (fc1 in = 2193, ap_hmr_fwd),
 *          2 = copenet_singleview (fc1 in = 2196, ap_singleview_fwd), 3 = muhmr (fc1 in = 2329, ap_muhmr_fwd). */
int ap_net_create(ap_net** out, int device, int precision, int variant);
void ap_net_destroy(ap_net* h);
int ap_net_set_tensor(ap_net* h, const char* name, const float* host_data, const int64_t* shape, int ndim);
int ap_net_finalize(ap_net* h);
int ap_net_precision(const ap_net* h);
/* AP_PREC_F16 range sentinel (no-ops returning AP_OK for the other precisions).  The pooling stage of every trunk pass sets a
 * per-handle, host-mapped flag when a trunk feature is not finite.
 *   ap_net_set_range_check  mode 1 (default): deferred -- nothing is added to the hot path; the flag is sticky and the NEXT
 *                           trunk-running call on the handle returns AP_ERANGE, as does ap_net_range_status;
 *                           mode 2: synchronous -- every trunk-running call (ap_trunk_fwd, ap_copenet_fwd, ap_hmr_fwd, ...)
 *                           synchronises its stream after the trunk and returns AP_ERANGE for its OWN pass; mode 0: off
 *   ap_net_range_status     synchronises `stream` and returns AP_OK or AP_ERANGE; reset != 0 clears the flag
 * What the sentinel sees: EVERY stored activation.  An overflow can only be born where an fp32 result is converted for storage,
 * and every stored activation of the trunk is post-ReLU, i.e. it is born as +inf = 0x7c00: each epilogue of the fp16 kernel set
 * folds the packed dwords it stores into a running 16-bit maximum (one v_pk_max_i16 / v_pk_maximum3_f16 per two or four values)
 * and sets the flag once per thread at the end of the kernel (ap_common.h: ap_rng_note / ap_rng_flush); the pooling stage checks
 * the features as well.  (A check of the pooled features alone is NOT enough: measured, the NaNs an inf turns into downstream
 * carry a set sign bit and every ReLU clears them.)  The weight check of ap_net_finalize is exact.
 * In the deferred mode the forward that overflows itself returns AP_OK: a one-shot caller, or the last batch of a run, asks
 * ap_net_range_status once its stream is done (airpose_amd: Pending.synchronize() and copenet.range_status() do). */
int ap_net_set_range_check(ap_net* h, int mode);
int ap_net_range_status(ap_net* h, void* stream, int reset);
/* The same flag without any synchronisation, for callers that have already waited for the work they ask about (an event of their
 * own): AP_OK or AP_ERANGE.
 * A serving loop with several batches in flight must not synchronise the handle's streams to learn about ONE batch, and must not
 * blame batch i for what batch i+1 did.  The flag is therefore one word per internal pass stream, and
 *   ap_net_range_mark_next(h, slot)  makes the NEXT trunk-running call on the handle snapshot, on each of its pass streams behind
 *                                    its last kernel there, that stream's word into slot `slot` (of AP_RANGE_SLOTS): the snapshot
 *                                    sees the kernels of this batch and of earlier ones -- never a later batch's, whichever stream
 *                                    runs ahead.  slot = -1 cancels.  The slot's previous user must be complete (its words are cleared).
 *   ap_net_range_slot(h, slot)       reads the slot (no sync; the caller has waited for an event recorded behind that call's output
 *                                    stream): AP_OK or AP_ERANGE.
 * airpose_amd.pipeline: submit() marks the batch's slot, Pending.synchronize() reads it after the batch's own event;
 * TwoViewInference.__call__ marks, waits for its stream and reads: a one-shot forward reports its own pass. */
#define AP_RANGE_SLOTS 8
/* Kernel launches of the conv stack (layer1 .. layer4: everything between the stem + pool kernel and the average pool) in the most
 * recent trunk-running call on the handle, summed over its passes -- what the library actually chose for that batch size (fused
 * blocks, pairs, image-resident blocks), for bench.py's per-launch figures. */
int ap_net_last_conv_launches(const ap_net* h);
int ap_net_range_peek(const ap_net* h);
int ap_net_range_mark_next(ap_net* h, int slot);
int ap_net_range_slot(const ap_net* h, int slot);

/* What the handle's precision costs on THIS checkpoint (the 1e-4 bar of north_star as a run-time check, not a
 * benchmark-checkpoint sample).  Runs a seeded probe batch of n_pairs two-view pairs (crops ~ N(0,1), bb / init_position as in
 * SURVEY 8d, 3 IEF iterations: copenet.forward, model_copenet.py:112-159) through the handle's trunk AND through an exact-fp32
 * trunk packed from the same host tensors (AP_PREC_FP32 kernels; built on first use, kept until the next ap_net_finalize), both
 * followed by the handle's own fp32 regressor, synchronises `stream` and compares on the host in fp64:
 *   err8[0..3]  max|a-b| / max|b| over the slices  translation (pose[:, :3]) | 6-D rotations (pose[:, 3:]) | betas |
 *               projected translation (u, v) = 1475 t_xy / t_z + (960, 540)  (the 2-D joints see the translation error through the
 *               camera: constants.py:7-11, geometry.py:63-91)
 *   err8[4..7]  max |a-b| / (1e-2 + |b|) over the same slices (element-wise)
 * AP_PREC_FP32 handles return zeros.  Returns AP_ERANGE when the probe batch leaves the fp16 range.  1 <= n_pairs <= 64. */
int ap_net_parity_probe(ap_net* h, int n_pairs, uint64_t seed, double* err8, void* stream);

/* copenet.forward_feat_ext (model_copenet.py:161-176).
 * x: [n_img][3][224][224] NCHW fp32 (the reference input contract); feat: [n_img][2048] fp32.
 * A list of >= 128 images runs as two concurrent passes over its halves on the handle's internal streams (forked from and joined
 * back into `stream`; ap_net_set_dual_stream(h, 0): one pass on `stream` itself); a feature row does not depend on the split. */
int ap_trunk_fwd(ap_net* h, const float* x_nchw, int n_img, float* feat, void* stream);

/* The two forward_feat_ext calls of copenet.forward (model_copenet.py:140-141: xf0 = forward_feat_ext(x0), xf1 = ...(x1)) as ONE
 * call: x0, x1: [B][3][224][224] NCHW fp32; feat: [2][B][2048] fp32 (view 0 rows, then view 1 rows).  Same kernels and the same
 * two concurrent passes as inside ap_copenet_fwd, which is this call followed by ap_regressor_fwd on the handle's own feature
 * buffer; a caller that keeps the features (or runs the IEF loop / the SMPL-X stage of batch i on another stream while the trunk of
 * batch i+1 runs on this one: airpose_amd.pipeline.TwoViewInference.submit) uses the two halves. */
int ap_trunk_fwd_twoview(ap_net* h, const float* x0, const float* x1, int B, float* feat, void* stream);

/* ap_trunk_fwd_twoview with the inputs ordered on one stream and the features on another: x0 / x1 are read after the work queued
 * on in_stream so far; feat is complete in the order of out_stream; in_stream is NOT made to wait for the trunk.  Back-to-back calls
 * queue their passes behind each other on the handle's internal streams, so the stem of batch i+1 starts while the last layers
 * of batch i still run (the serving loop of airpose_amd.pipeline.TwoViewInference.submit).  The caller keeps x0 / x1 unchanged
 * until work queued on out_stream after this call has run.  in_stream == out_stream: ap_trunk_fwd_twoview. */
int ap_trunk_fwd_twoview_async(ap_net* h, const float* x0, const float* x1, int B, float* feat, void* in_stream, void* out_stream);

/* IEF loop of copenet.forward (model_copenet.py:119-159) starting from trunk features.
 * xf*: [B][2048]; bb*, pos*: [B][3]; init_theta*: [tb][>=132] with batch stride theta*_bs floats
 * (0 broadcasts one row; NULL = model mean pose); init_shape*: [sb][10] likewise (NULL = mean shape).
 * Outputs pose*: [B][135] = trans3 | root6D | body 21x6D, betas*: [B][10]. */
int ap_regressor_fwd(ap_net* h, const float* xf0, const float* xf1, const float* bb0, const float* bb1,
                     const float* pos0, const float* pos1, const float* init_theta0, int theta0_bs,
                     const float* init_theta1, int theta1_bs, const float* init_shape0, int shape0_bs,
                     const float* init_shape1, int shape1_bs, int B, int iters, float* pose0, float* betas0,
                     float* pose1, float* betas1, void* stream);

/* One forward_reg evaluation for ONE view (model_copenet.py:185-188,198-199), for the view-split /
 * on-drone topology (README.md:238-241: step1/step2/step3 with the partner's state exchanged between
 * steps).  pose_in [B][135], betas_in [B][10] = this view's current state; partner [B][partner_ld] =
 * the other view's art pose (126) | shape (10); outputs as above. */
int ap_regressor_step(ap_net* h, const float* xf, const float* bb, const float* pose_in, const float* betas_in,
                      const float* partner, int partner_ld, int B, float* pose_out, float* betas_out, void* stream);

/* ap_regressor_step in two halves, so that the cross-view exchange hides behind the partner-independent columns (SURVEY 8e;
 * model_copenet.py:185-193: xc = [xf 2048 | bb 3 | pos 3 | orient 6 | art 126 | shape 10 || partner's art 126 | shape 10], and
 * fc1 -> fc2 -> dec is one affine map Wf (145 x 2332), so a step is a sum over column groups):
 *   ap_regressor_feat_part    hfeat[b]   = bf + Wf[:, :2048] xf[b]                 once per forward (constant over the iterations)
 *   ap_regressor_step_local   partial[b] = hfeat[b] + Wf[:, 2048:2196] [bb | pose_in | betas_in]     while the exchange is in flight
 *   ap_regressor_step_finish  out[b]     = [pose_in | betas_in] + partial[b] + Wf[:, 2196:2332] partner[b]
 * hfeat, partial: [B][148] fp32, caller-owned.  Same result as ap_regressor_step up to fp32 summation order.  The folded map must
 * be in use (ap_net_fold_status == 1, two-view handle): AP_ESTATE otherwise -- such a handle steps through ap_regressor_step. */
int ap_regressor_feat_part(ap_net* h, const float* xf, int B, float* hfeat, void* stream);
int ap_regressor_step_local(ap_net* h, const float* hfeat, const float* bb, const float* pose_in, const float* betas_in, int B,
                            float* partial, void* stream);
int ap_regressor_step_finish(ap_net* h, const float* partial, const float* pose_in, const float* betas_in, const float* partner,
                             int partner_ld, int B, float* pose_out, float* betas_out, void* stream);

/* copenet_singleview baseline (models/model_copenet_singleview.py:108-168; needs a variant-2 handle: fc1 is 1024 x 2196,
 * xc = [xf | bb | pose135 | shape10]): trunk + `iters` regressor evaluations for ONE view, no cross-view input.
 * x [B][3][224][224], bb/pos [B][3], init_theta [tb][>=132] / init_shape [sb][10] as for ap_regressor_fwd (NULL = model
 * mean); outputs pose [B][135], betas [B][10]. */
int ap_singleview_fwd(ap_net* h, const float* x, const float* bb, const float* pos, const float* init_theta,
                      int theta_bs, const float* init_shape, int shape_bs, int B, int iters, float* pose, float* betas,
                      void* stream);

/* muhmr two-view baseline (models/model_muhmr.py:112-199; needs a variant-3 handle): both trunks + `iters` evaluations of
 * forward_reg with xc = [xf | cam3 | orient6 | art126 | shape10 | partner's art126, shape10] and decoders decpose (132) /
 * decshape / deccam.  init_cam* [cb][3] (batch stride cam*_bs floats, 0 broadcasts; both NULL = the model's init_cam),
 * init_theta* / init_shape* as for ap_regressor_fwd.  Outputs campose* [B][135] = pred_cam (3) | pred_pose (132),
 * betas* [B][10]. */
int ap_muhmr_fwd(ap_net* h, const float* x0, const float* x1, const float* init_cam0, int cam0_bs, const float* init_cam1,
                 int cam1_bs, const float* init_theta0, int theta0_bs, const float* init_theta1, int theta1_bs,
                 const float* init_shape0, int shape0_bs, const float* init_shape1, int shape1_bs, int B, int iters,
                 float* campose0, float* betas0, float* campose1, float* betas1, void* stream);

/* copenet.forward (model_copenet.py:112-159): both trunks (one batched 2B pass, shared weights) + IEF. */
int ap_copenet_fwd(ap_net* h, const float* x0, const float* x1, const float* bb0, const float* bb1,
                   const float* pos0, const float* pos1, const float* init_theta0, int theta0_bs,
                   const float* init_theta1, int theta1_bs, const float* init_shape0, int shape0_bs,
                   const float* init_shape1, int shape1_bs, int B, int iters, float* pose0, float* betas0,
                   float* pose1, float* betas1, void* stream);

/* model_hmr.copenet.forward (copenet/src/copenet/models/model_hmr.py:112-141; BASELINE config 0's network):
 * single view, trunk + 3 x forward_reg (:160-172) + rot6d_to_rotmat.  Needs a variant-1 handle.
 * init_theta [tb][>=132] / init_shape [sb][10] / init_cam [cb][3] with batch strides (0 = broadcast) or NULL =
 * the model's mean parameters.  Outputs rotmat [B][22][3][3], betas [B][10], cam [B][3]. */
int ap_hmr_fwd(ap_net* h, const float* x, int B, int iters, const float* init_theta, int theta_bs,
               const float* init_shape, int shape_bs, const float* init_cam, int cam_bs, float* rotmat, float* betas,
               float* cam, void* stream);

/* Feature-level evaluations of the baseline heads (the reference modules expose them as forward_reg):
 *   ap_hmr_reg         model_hmr.copenet.forward_reg (copenet/src/copenet/models/model_hmr.py:160-172), `iters` evaluations
 *                      from features xf [B][2048]; state in / out = 6-D pose [B][132] | shape [B][10] | cam [B][3]
 *                      (NULL inputs = the model's mean parameters; *_bs = batch stride in floats, 0 = broadcast)
 *   ap_singleview_reg  model_copenet_singleview.copenet.forward_reg (models/model_copenet_singleview.py:159-170)
 * The muhmr head (models/model_muhmr.py:177-203) runs through ap_regressor_fwd on its variant-3 handle with the cameras
 * in the position slots. */
int ap_hmr_reg(ap_net* h, const float* xf, int B, int iters, const float* pose_in, int pose_bs, const float* shape_in,
               int shape_bs, const float* cam_in, int cam_bs, float* pose_out, float* shape_out, float* cam_out,
               void* stream);
int ap_singleview_reg(ap_net* h, const float* xf, const float* bb, const float* pos, const float* init_theta, int theta_bs,
                      const float* init_shape, int shape_bs, int B, int iters, float* pose, float* betas, void* stream);

/* One fused convolution of the trunk: y = act(conv(x, w) * scale + shift (+ res)), the building block of
 * Bottleneck.forward (model_copenet.py:27-47: conv -> BN -> [+ residual] -> ReLU).  Exposed so the kernel can be
 * unit-tested and reused.  NHWC activations x [N][H][W][Cin], y/res [N][Ho][Wo][Cout]; w [Cout_pad][k][k][Cin]
 * with Cout_pad = Cout rounded up to 128 (zero rows), scale/shift [Cout_pad]; element type of x/w/res/y is
 * bf16 (AP_PREC_BF16), fp16 (AP_PREC_F16), float (AP_PREC_FP32) or split-bf16 (AP_PREC_BF16X2: 4 bytes per element, planar groups of 8
 * channels = 8 bf16 hi parts then 8 bf16 lo parts); Cin a multiple of 64 (bf16, fp16) / 32 (fp32, bf16x2), Cout of 8 (4: fp32). */
int ap_conv2d_nhwc(int precision, const void* x, const void* w, const float* scale, const float* shift,
                   const void* res, void* y, int N, int H, int W, int Cin, int Cout, int ksize, int stride, int pad,
                   int relu, void* stream);

/* Fused 64-plane bottleneck (layer1 of the trunk; precision = AP_PREC_BF16 or AP_PREC_F16: the storage type of x, y and the
 * weights): Bottleneck.forward, model_copenet.py:27-47, as ONE kernel with both intermediates resident in LDS.
 * x [N][H][W][Cin], y [N][H][W][256]; H, W multiples of 14.
 *   downsample = 0: Cin = 256, w1 [64..][256], w2 [64..][3][3][64], w3 [256][64];  y = relu(bn3(conv3(..)) + x)
 *   downsample = 1: Cin = 64,  w1 [64..][64],  w3 [256][128] = [conv3 | downsample conv] (K-concatenated, BN scales
 *                   folded into the weights, s3 = 1, h3 = shift3 + shift_ds);           y = relu(W3 . [mid2 | x] + h3)
 * Weight rows are K-contiguous bf16 as for ap_conv2d_nhwc; s*, h* are fp32 BatchNorm scale / shift.  Matches the
 * three-convolution path to bf16 rounding of the fp32 accumulations. */
int ap_bottleneck64_nhwc(int precision, const void* x, const void* w1, const float* s1, const float* h1, const void* w2,
                         const float* s2, const float* h2, const void* w3, const float* s3, const float* h3, void* y,
                         int N, int H, int W, int Cin, int downsample, void* stream);

/* The identity form of ap_bottleneck64_nhwc (Cin = 256) with conv1 of the NEXT bottleneck (1x1, 256 -> 128, + BN + ReLU:
 * model_copenet.py:29-31 of layer2.0) computed on the block output while it is still in registers (bottleneck2.hip, tail
 * variant): t1n [N][H][W][128] = relu(bn1n(conv1n(y))).  w1n [128][256] K-contiguous rows, s1n / h1n [128].  y_even = 1: y is
 * stored at the even (row, column) pixels only -- all that layer2.0's stride-2 downsample branch (model_copenet.py:41-42,
 * :97-102) reads of it; the other pixels of y are left untouched.  y and t1n carry the same bits as ap_bottleneck64_nhwc
 * followed by ap_conv2d_nhwc. */
int ap_bottleneck64_tail_nhwc(int precision, const void* x, const void* w1, const float* s1, const float* h1, const void* w2,
                              const float* s2, const float* h2, const void* w3, const float* s3, const float* h3, void* y,
                              const void* w1n, const float* s1n, const float* h1n, void* t1n, int y_even, int N, int H, int W,
                              void* stream);

/* One identity bottleneck of layer3 (1024 -> 256 -> 256 -> 1024 on 14 x 14 images; Bottleneck.forward, model_copenet.py:27-47
 * as iterated by :64) as ONE image-resident kernel (block_img.hip): a workgroup owns an image, both 256-channel intermediates
 * stay in its LDS, the weights stream from L2 as MFMA fragments.  x, y [N][14][14][1024] NHWC in the storage type of `precision`
 * (AP_PREC_BF16 or AP_PREC_F16); s*, h* fp32 BatchNorm scale / shift ([256], [256], [1024]).  The weights are consumed as one
 * caller-owned stream of ap_block_img_stream_bytes() bytes built by ap_block_img_pack from w1 [256][1024], w2 [256][3][3][256],
 * w3 [1024][256] (K-contiguous rows as for ap_conv2d_nhwc); re-pack whenever the weights change.
 * y = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + x) with the intermediates rounded to the storage type exactly
 * where the three-convolution path rounds them. */
int64_t ap_block_img_stream_bytes(void);
/* conv2 of a layer2 bottleneck (3 x 3, stride 1, 128 -> 128 channels at 28 x 28; model_copenet.py:32-34 with :18) + bn2 + ReLU with
 * HALF AN IMAGE resident in the LDS of one CU (conv_img3.hip: 14 of the 28 columns over all rows + a halo column either side; the
 * conv2 phase of the image-resident layer3 block as a kernel of its own).  Operator form for tests and benches: x, y [N][28][28][128]
 * 16-bit NHWC (y_tiled != 0: y in the fragment-tiled layout the pair kernel reads); wstream = ap_conv_img3_pack of the K-contiguous
 * rows [128][3][3][128] (ap_conv_img3_stream_bytes bytes, caller-owned).  Sums in conv_slab's K order: same bits as ap_conv2d_nhwc. */
int64_t ap_conv_img3_stream_bytes(void);
int ap_conv_img3_pack(int precision, const void* w2, void* wstream, void* stream);
int ap_conv_img3_nhwc(int precision, const void* x, const void* wstream, const float* scale, const float* shift, void* y, int N,
                      int y_tiled, void* stream);

/* Stride-2 3x3 convolution of layer2.0 (128 -> 128 channels, 56 x 56 -> 28 x 28) + bn2 + ReLU in polyphase form, a quarter of an
 * output image per workgroup (conv_s2p.hip; replaces conv2 of the stage's first Bottleneck, model_copenet.py:32-34 with :18).
 * x: [N][56][56][128], y: [N][28][28][128] 16-bit NHWC (y_tiled != 0: fragment-tiled); wstream = ap_conv_s2p_pack of the
 * K-contiguous rows [128][3][3][128] (ap_conv_s2p_stream_bytes bytes, caller-owned).  Its K order (the taps phase by phase) is its
 * own: the result equals ap_conv2d_nhwc's to fp32 summation order, not bit for bit. */
int64_t ap_conv_s2p_stream_bytes(void);
int ap_conv_s2p_pack(int precision, const void* w2, void* wstream, void* stream);
int ap_conv_s2p_nhwc(int precision, const void* x, const void* wstream, const float* scale, const float* shift, void* y, int N,
                     int y_tiled, void* stream);

/* Pointwise (1 x 1, stride 1) convolution + BatchNorm (+ identity) + ReLU for the 14 x 14 / 7 x 7 stages (conv1 and conv3 of
 * Bottleneck.forward, model_copenet.py:29-31, 38-45) on the one-wave-per-SIMD mainloop of conv_pw.hip: x [M][Cin], res (or NULL)
 * and y [M][Cout] NHWC pixel rows in the storage type of `precision` (AP_PREC_BF16 or AP_PREC_F16); M a multiple of 196 (whole
 * 14 x 14 images, or 7 x 7 images in fours), Cin of 128 (>= 256), Cout of 256.  The weights are consumed as a caller-owned stream
 * of ap_conv_pw_stream_bytes(Cin, Cout) bytes built by ap_conv_pw_pack from w [Cout][Cin] (K-contiguous rows as for
 * ap_conv2d_nhwc).  y = relu(scale * conv(x) + shift (+ res)): bit-identical to ap_conv2d_nhwc on the same operands. */
int64_t ap_conv_pw_stream_bytes(int Cin, int Cout);
int ap_conv_pw_pack(int precision, const void* w, int Cin, int Cout, void* wstream, void* stream);
int ap_conv_pw_nhwc(int precision, const void* x, const void* wstream, const float* scale, const float* shift, const void* res,
                    void* y, int M, int Cin, int Cout, void* stream);
/* ... of a stage's first block: conv3 with the downsample branch as a second K segment (model_copenet.py:38-45 with :41-42,
 * 97-102): y [N][Ho][Ho][Cout] = relu(scale * ([t2 | x sampled at (ho * stride, wo * stride)] . w^T) + shift), t2
 * [N][Ho][Ho][Cin], x [N][Ho stride][Ho stride][Cin2]; the stream is ap_conv_pw_pack's of w [Cout][Cin + Cin2] (pass
 * Cin + Cin2 as its Cin).  Ho * Ho must divide 196 (7 or 14) and N Ho Ho be a multiple of 196. */
int ap_conv_pw_ds_nhwc(int precision, const void* t2, const void* x, const void* wstream, const float* scale, const float* shift,
                       void* y, int N, int Ho, int Cin, int Cin2, int Cout, int stride, void* stream);
/* ... conv2 of a stage's first block: 3 x 3, stride 2, padding 1 (model_copenet.py:32-34 with :18) as nine pointwise taps on the
 * same kernel (a 64-channel chunk of ONE tap per staging step; out-of-image taps zeroed on the way into the LDS): x [N][H][H][Cin]
 * -> y [N][H/2][H/2][Cout] = relu(scale * conv(x) + shift); the stream is ap_conv_pw_pack's of w [Cout][3][3][Cin] (pass 9 Cin as
 * its Cin).  H = 14 or 28, N (H/2)^2 a multiple of 196, Cin / 64 a power of two.  Bit-identical to ap_conv2d_nhwc(ksize 3,
 * stride 2, pad 1) on the same operands (K in [tap][Cin] order on both). */
int ap_conv_pw_k3s2_nhwc(int precision, const void* x, const void* wstream, const float* scale, const float* shift, void* y, int N,
                         int H, int Cin, int Cout, void* stream);
int ap_block_img_pack(int precision, const void* w1, const void* w2, const void* w3, void* wstream, void* stream);
int ap_block_img_nhwc(int precision, const void* x, const void* wstream, const float* s1, const float* h1, const float* s2,
                      const float* h2, const float* s3, const float* h3, void* y, int N, void* stream);

/* Fused pair kernel on NHWC 16-bit tensors (precision = AP_PREC_BF16 or AP_PREC_F16): conv3 (+ identity | + folded downsample,
 * ReLU) of a bottleneck and conv1 (+ ReLU) of the NEXT bottleneck as one pixel-local kernel (conv_pair.hip; replaces
 * model_copenet.py:38-45 of one block and :29-31 of the next per launch).  The two weight matrices are consumed as ONE stream of
 * 16-KiB tiles; the stream is caller-owned: ap_conv_pair_stream_bytes gives its size (negative = unsupported shape),
 * ap_conv_pair_pack builds it on the device from w3 [4P][P + P2] and w1 [N1][4P] (K-contiguous rows as for ap_conv2d_nhwc; w1 NULL
 * when N1 = 0) -- re-pack whenever the weights change; the library caches nothing.
 *   ap_conv_pair_nhwc     identity block: t2 [M][P] (conv2 output), res [M][4P] (block input), s3/h3 (bn3), s1/h1 (next bn1);
 *                         out [M][4P] = relu(bn3(conv3 t2) + res), t1n [M][N1] = relu(bn1(conv1 out)).
 *                         (P, N1) in {(128,128), (128,256), (256,256)}
 *   ap_conv_pair_ds_nhwc  stage-first block: w3 = [conv3 | downsample conv] with both BatchNorm scales folded into the rows
 *                         (s3 = ones [4P], h3 = shift3 + shift_ds); x [N][Ho*stride][Ho*stride][P2] = the block input, read at
 *                         the strided pixel as a second K segment; no identity; with the next conv1 (N1 > 0) or alone (N1 = 0:
 *                         s1 / h1 / t1n NULL).  (P, P2, N1) in {(128,256,128), (256,512,0)} */
int64_t ap_conv_pair_stream_bytes(int P, int P2, int N1);
int ap_conv_pair_pack(int precision, const void* w3, const void* w1, int P, int P2, int N1, void* wstream, void* stream);
int ap_conv_pair_nhwc(int precision, const void* t2, const void* wstream, const float* s3, const float* h3, const void* res,
                      const float* s1, const float* h1, void* out, void* t1n, int M, int P, int N1, void* stream);
int ap_conv_pair_ds_nhwc(int precision, const void* t2, const void* x, const void* wstream, const float* s3, const float* h3,
                         const float* s1, const float* h1, void* out, void* t1n, int N, int Ho, int P, int P2, int stride, int N1,
                         void* stream);

/* Tuning/testing knob (process-wide, one atomic word: safe to set while handles run on other threads; a launch sees the
 * old or the new value): tile configuration of the convolution kernels.  -1 = automatic, 0..13 = software-pipelined
 * LDS-DMA ring kernel (tile / wave / ring-depth variants, conv_pipe.hip), 100 = register-staged 2-stage kernel,
 * 14 = stride-1 3x3 convolutions with the nine taps read from one LDS slab per 64-channel chunk (conv_slab.hip; bf16,
 * image rows of at most 29 pixels; other shapes run configuration 11), which the automatic choice uses for the conv2
 * layers of layer2-4; 17 = pointwise convolutions on three lean workgroups per CU (conv_lean.hip; bf16; bit-identical to 11;
 * other shapes run configuration 11), which the automatic choice uses for short contractions with many channel tiles
 * (conv3 of layer2-4); -4 = automatic without either (ring kernel everywhere), -5 = automatic without configuration 17.
 * Results are identical (bitwise) for every setting except 14 / the automatic choice on those layers: the slab kernel sums
 * the K range channel-chunk-outer, tap-inner instead of tap-outer, i.e. it agrees to fp32 re-association. */
int ap_set_conv_config(int cfg);
/* Profiling aid: device buffer of 160 uint64 receiving per-phase cycle stamps of workgroup 0 of the pipelined
 * convolution kernel (2 waves x 8 K steps x 10 stamps); NULL (default) disables it. */
int ap_debug_set_trace(void* device_buf_160_u64);

/* Stage timing for bench.py: when enabled, HIP events bracket the stem, the implicit-GEMM conv stack,
 * the pooling tail and the regressor on the caller's stream.  ap_net_timing synchronises on the last
 * recorded events and returns the ACCUMULATED milliseconds and the number of recorded passes since
 * the last reset.  ms[0]=stem+maxpool, ms[1]=conv stack (the 52 conv layers of a pass), ms[2]=avgpool, ms[3]=regressor.
 * on = 1: every stage; on = 2: the conv stack only (two events per trunk pass: an event record costs a ~5 us bubble on
 * the stream, so the timed region of bench.py carries only the pair its roofline line needs); on = 0: off. */
int ap_net_enable_timing(ap_net* h, int on);
int ap_net_timing(ap_net* h, double ms[4], int64_t* passes, int reset);
/* forward_reg has no activation between fc1, fc2 and the decoders (dropout is the identity in eval mode,
 * model_copenet.py:186-202), so ap_net_finalize also folds them, in fp64, into one 145 x 2332 affine map
 * (like the BatchNorm fold).  on = 1 (default) evaluates the folded map, on = 0 the literal three-GEMM chain;
 * both are parity-tested against the reference. */
int ap_net_set_fold(ap_net* h, int on);
/* The fold is exact algebra but not unconditionally well-conditioned: ap_net_finalize (copenet-layout handles) evaluates the
 * fp32-rounded folded map and the literal chain in fp64 on a fixed probe batch and, when they differ by more than 1e-5 of the
 * output scale, switches THIS handle to the literal chain (a line on stderr says so; ap_net_set_fold(h, 1) is then refused).
 * Returns 1 = folded map in use, 2 = literal chain by the caller's choice, 0 = literal chain because the fold was rejected;
 * *probe_rel_err (optional) = the measured difference. */
int ap_net_fold_status(const ap_net* h, double* probe_rel_err);
/* Test aid: the bar of that check (default 1e-5); the handle must be finalised again for it to take effect. */
int ap_net_set_fold_bar(ap_net* h, double bar);
/* With the folded map (ap_net_set_fold(1)): on = 1 (default) runs a whole IEF forward as two launches (split-K feature
 * GEMM + ONE kernel for initialisation, all iterations with the cross-view swap, and the pose/betas split: the swap only
 * couples the two views of a pair, which one workgroup owns); on = 0 as one GEMM per iteration plus glue kernels. */
int ap_net_set_fuse_ief(ap_net* h, int on);
/* First block of a stage: on = 1 (default) folds the downsample branch into conv3 as a second K segment
 * (relu(bn3(conv3(t)) + bn_ds(conv_ds(x))) as ONE GEMM over [t | x]: both BN scales folded into the weights in
 * fp64, no downsample tensor written or re-read); on = 0 runs the two convolutions of Bottleneck.forward
 * (model_copenet.py:38-45) separately.  Both are parity-tested. */
int ap_net_set_fuse_ds(ap_net* h, int on);
/* 16-bit and bf16x2 modes: on = 1 (default) runs conv1+bn1+relu+maxpool as one fused kernel, on = 0 as stem + maxpool
 * kernels (bit-identical results; kept for A/B measurement).  16-bit modes: on = 1 is the persistent form (one workgroup per
 * CU, compute and feeding waves: stem.hip), on = 2 the strip kernel of rounds 2-5 (a workgroup per two pooled rows): same bits. */
int ap_net_set_fuse_stem(ap_net* h, int on);
/* 16-bit modes: on = 1 computes AvgPool2d(7) + view (model_copenet.py:173-175) in the epilogue of the last convolution
 * (layer4.2 conv3 + bn3 + identity + ReLU, :38-47): the 7 x 7 x 2048 block output is never written and no pooling kernel runs;
 * on = 0 (default: measured neutral to slightly slower inside the two-stream trunk) writes it and runs the pooling kernel.
 * Bit-identical features (same summands, same summation order). */
int ap_net_set_fuse_pool(ap_net* h, int on);
/* 16-bit modes: on = 1 (default) stores the tensors whose only reader is the fused pair kernel (conv2's output of a pair
 * block; a pair block's output when the next block is an identity pair block) in that kernel's fragment order
 * [M/16][C/8][16 pixels][8 channels] instead of NHWC rows, so each of its wave-wide 16-byte accesses covers one contiguous KiB;
 * on = 0 keeps NHWC everywhere.  The layout is internal to the trunk workspaces; features are bit-identical either way. */
int ap_net_set_tiled(ap_net* h, int on);
/* 16-bit modes: on = 1 (default) runs each layer1 bottleneck as ONE kernel (ap_bottleneck64_nhwc: the 64-channel
 * intermediates never leave the CU; bottleneck2.hip: weights resident in LDS, x in registers), on = 0 as its three
 * (two + folded-downsample) convolutions.  Both give the same bits (parity-tested). */
int ap_net_set_fuse_block(ap_net* h, int on);
/* 16-bit modes: conv3 (+ identity, ReLU) of an identity bottleneck and conv1 of the NEXT bottleneck as one pixel-local kernel
 * (conv_pair.hip; layer2 and layer3 identity blocks, layer2 -> layer3, and the first blocks of layer2 / layer3 with their
 * downsample branch as a second K segment): the block output makes one HBM trip less per block boundary.  Bit-identical to the two stand-alone kernels.  Default on; replaces model_copenet.py:38-45 (+ :29-31 of the
 * next block) per launch. */
int ap_net_set_fuse_pair(ap_net* h, int on);
/* 16-bit modes: on = 1 (default) computes conv1 of layer2.0 inside the kernel of layer1's last bottleneck
 * (ap_bottleneck64_tail_nhwc): the 56 x 56 x 256 block output is not read back for it (model_copenet.py:29-31 at :64).
 * Bit-identical to the stand-alone convolution. */
int ap_net_set_fuse_tail(ap_net* h, int on);
/* 16-bit modes: conv1 of the layer3 / layer4 bottlenecks that no fused kernel covers (layer4: all three), conv3 + downsample of
 * layer4.0 and -- in a pass that has the chip to itself -- the 3 x 3 / stride-2 conv2 of layer3.0 / layer4.0 on conv_pw.hip's kernel
 * instead of the generic 128 x 128-tile kernels: 1 (default) = when the layer's 196-pixel x 256-channel tiles fill half the chip,
 * or whole rounds of it to 80 % (BASELINE config 2: yes, +0.7 % two concurrent passes / +1.5 % one pass; 64 pairs: layer4.0
 * only); 2 = whenever the shape is supported, and conv3 + identity as well (slower than the lean kernel: A/B aid); 3 = as 1 without
 * the size rule; 4 = as 1 without the 3 x 3 layers; 0 = never.  Features are bit-identical either way. */
int ap_net_set_pw_conv(ap_net* h, int on);
/* 16-bit modes: on = 1 (default): a block output whose only remaining reader is the next block's stride-2 downsample branch
 * (model_copenet.py:41-42, :97-102; its conv1 having been computed by the producing kernel) is stored at the even pixels only.
 * Features are bit-identical either way. */
int ap_net_set_even_out(ap_net* h, int on);
/* 16-bit modes: layer3's identity bottlenecks (layer3.1 .. 3.5) as ONE image-resident kernel each (ap_block_img_nhwc) instead
 * of conv2 + the fused conv3 -> conv1 pairs: on = 1 (default) when the pass fills whole rounds of the chip (the kernel runs an
 * image per CU: >= 7/8 of ceil(n / CUs) * CUs images), 2 always, 0 never.  Both paths sum in the same order (conv2: the slab
 * kernel's): trunk features are bit-identical, so a pair's result does not depend on the batch it arrives in. */
int ap_net_set_s2p(ap_net* h, int on);           /* conv2 of layer2.0 on conv_s2p.hip: 1: at every batch size (other fp32 summation order than the generic
                                                  * kernels); 0 (default): the generic stride-2 kernels -- the polyphase kernel is faster alone and slower in the two-pass schedule */
int ap_net_set_img3(ap_net* h, int on);          /* conv2 of the layer2 identity blocks on conv_img3.hip: 1 (default) when the pass fills whole rounds of
                                                 * the chip with half images, 2 always, 0 never (slab kernel); same bits */
int ap_net_set_img_block(ap_net* h, int on);
/* images per depth-first trunk chunk (0 = library default); tuning knob, results are unaffected */
int ap_net_set_chunk(ap_net* h, int images_per_chunk);
/* Two-view forwards (>= 64 images per view, both views within one chunk) run the two views as two concurrent trunk
 * passes on two internal streams forked from / joined to the caller's stream (default on; results are bit-identical to
 * the single pass).  0 = one pass over the concatenated views. */
int ap_net_set_dual_stream(ap_net* h, int on);

/* ---------------------------------------------------------------------------------------------
 * SMPL-X handle.  Replaces smplx.SMPLX(model_dir, batch_size=.., create_transl=False)
 * (call site copenet/src/copenet/copenet_twoview.py:36-45; upstream smplx==0.1.28 semantics). */
typedef struct ap_smplx_model {
    int32_t num_verts, num_joints, num_faces; /* 10475, 55, 20908 */
    int32_t num_shape_coeffs;                 /* betas + expression columns of shapedirs (20) */
    int32_t num_extra, num_landmarks;         /* 21, 51 */
    const float* v_template;                  /* [V][3] */
    const float* shapedirs;                   /* [V][3][num_shape_coeffs] */
    const float* posedirs;                    /* [(J-1)*9][V*3] */
    const float* J_regressor;                 /* [J][V] */
    const int64_t* parents;                   /* [J], parents[0] = -1 */
    const float* lbs_weights;                 /* [V][J] */
    const int64_t* faces;                     /* [F][3] */
    const int64_t* extra_joint_verts;         /* [num_extra] */
    const int64_t* lmk_faces_idx;             /* [num_landmarks] */
    const float* lmk_bary_coords;             /* [num_landmarks][3] */
} ap_smplx_model; /* all HOST pointers */

int ap_smplx_create(ap_smplx** out, const ap_smplx_model* model, int device);
void ap_smplx_destroy(ap_smplx* h);
int ap_smplx_num_joints_out(const ap_smplx* h); /* 127 */

/* SMPLX.forward(betas, body_pose, global_orient, transl, pose2rot=False)
 * (copenet_twoview.py:237-241).  global_orient [n][3][3] or NULL (identity); body_pose [n][21][3][3];
 * extra_pose [n][33][3][3] (jaw, leye, reye, 15 left hand, 15 right hand) or NULL (identity);
 * expression [n][10] or NULL; transl [n][3] or NULL.  vertices [n][V][3], joints [n][127][3]. */
int ap_smplx_fwd(ap_smplx* h, int n, const float* betas, const float* expression, const float* global_orient,
                 const float* body_pose, const float* extra_pose, const float* transl, float* vertices,
                 float* joints, void* stream);

/* Fused caller slice for one view: rot6d_to_rotmat -> SMPLX.forward(global_orient = I, transl = 0) ->
 * transform_smpl([R_root | trans]) -> perspective_projection(R = I, t = 0)
 * (copenet_twoview.py:222-223, 237-246, 307-311).  pred_pose [n][pose_ld]: trans3 (already un-scaled) |
 * 22 x 6D; betas [n][10]; cam_center [n][2] (intr[:, :2, 2]) or NULL.  Outputs: vertices_cam [n][V][3],
 * joints_cam [n][127][3], joints2d [n][127][2] (NULL if cam_center NULL), rotmat [n][22][3][3] or NULL. */
int ap_smplx_fwd_fused(ap_smplx* h, int n, const float* pred_pose, int pose_ld, const float* betas,
                       const float* cam_center, float fx, float fy, float* vertices_cam, float* joints_cam,
                       float* joints2d, float* rotmat, void* stream);
/* The caller slice of copenet_twoview.fwd_pass_and_loss for BOTH views in one pass, nothing left to the host between
 * the network and the tail (copenet_twoview.py:214-223, 237-279, 307-317):
 *   pred_smpltrans /= trans_scale          in place on pred_pose[:, :3] (:214-218; skipped when trans_scale == 0)
 *   rot6d_to_rotmat -> SMPLX.forward(betas, body rotations, global_orient = I, transl = 0) ->
 *   transform_smpl([R_root | pred_smpltrans]) -> perspective_projection(camera_center = intr[:, :2, 2])
 * and, in test mode (in_smpltrans != NULL), the "input" meshes of :258-279: betas = 0, the same body rotations,
 * transform_smpl([I | in_smpltrans]).
 * pred_pose [2B][pose_ld] (view 0 rows first), betas [2B][10], intr0 / intr1 [B][3][3] (or both NULL: no projection),
 * in_smpltrans [2B][3] or NULL.  Outputs: vertices [2B (4B with in_smpltrans)][V][3] (rows 2B.. = the input meshes),
 * joints_cam [2B][127][3], joints2d [2B][127][2] or NULL, rotmat [2B][22][3][3] or NULL. */
int ap_smplx_fwd_twoview(ap_smplx* h, int B, float* pred_pose, int pose_ld, float trans_scale, const float* betas,
                         const float* intr0, const float* intr1, float fx, float fy, const float* in_smpltrans,
                         float* vertices, float* joints_cam, float* joints2d, float* rotmat, void* stream);

/* Arithmetic of the blend-shape contraction v_posed = v_template + [beta | expr | pose_feature] . dirs^T:
 * AP_PREC_BF16X2 (default) = operands as split-bf16 pairs, four-term products on the bf16 matrix pipe, fp32 accumulate
 * and fp32 result (~1e-7 of the vertex scale from the fp32 path); AP_PREC_FP32 = exact fp32 MFMA chain (4x slower). */
int ap_smplx_set_blend_precision(ap_smplx* h, int precision);
/* Blend-shape contraction + skinning as ONE kernel (default on): taken when the call carries no hand / face poses (K = 224),
 * the model has at most 4 bones per vertex and the contraction runs in split-bf16 form; v_posed then never leaves the chip.
 * 0 = always the two-kernel path (contraction GEMM writing v_posed, then the skinning kernel).  Same arithmetic per product;
 * results agree to fp32 re-association.  4 = the fused kernel with the joints /
 * landmarks / projection stage inside it: the LAST workgroup of each group of 32 bodies computes them from the skinned joint
 * vertices its siblings left in a side buffer (write-through stores, arrival counter, one acquire) -- one launch less, but
 * measured 7 us slower per forward of 512 bodies than the joints kernel as its own launch, so not the default. */
int ap_smplx_set_fused(ap_smplx* h, int on);
/* Test aid: fill the coefficient workspace for n bodies with 0xFF bytes (NaN patterns), as a raw allocation may hold: every slot
 * the contraction reads must be rewritten by the next forward, the zero padding included. */
int ap_smplx_debug_poison_workspace(ap_smplx* h, int n);
/* on = 1: a HIP event between every two kernels of a forward (per-stage times; every record costs a bubble of several microseconds on
 * the stream); on = 2: one event in front of the first kernel and one behind the last -- the span of the whole tail without bubbles
 * inside, reported in ms[0]; 0: off */
int ap_smplx_enable_timing(ap_smplx* h, int on);
/* ms[0]=prep/chain (on = 2: the whole tail), ms[1]=blend-shape GEMM / fused contraction + skinning, ms[2]=skin, ms[3]=joints+projection */
int ap_smplx_timing(ap_smplx* h, double ms[4], int64_t* passes, int reset);

/* ---------------------------------------------------------------------------------------------
 * Stand-alone geometry helpers (copenet/src/copenet/utils/geometry.py:47-61, 63-91;
 * copenet/src/copenet/utils/utils.py:237-256). */
/* ---------------------------------------------------------------------------------------------
 * AirPose+ fitting loop (BASELINE config 5; copenet_real_data/scripts/bundle_adj.py:262-401): Adam over a sequence of L
 * frames on z (VPoser latent, [L][32]), the per-view root 6-D rotation phi [2][L][6] (pytorch3d convention) and
 * translation tau [2][L][3], and a shared beta [10]; objective = Geman-McClure 2-D reprojection of the first 24 SMPL-X
 * chain joints in both views against two detectors + VPoser prior + temporal smoothness, gradients by hand-written
 * adjoints (decoder MLP, 6-D, rotation matrix -> axis-angle -> rotation matrix, kinematic chain, projection).
 * The third-party pieces of the script (human_body_prior, VPoser weights, pytorch3d) are absent upstream: parity
 * unpinned, see oracle/fitting_ref.py.
 * ap_fit_create: body = an ap_smplx handle (rest joints and their shape directions); w1 [512][32], b1 [512], w2 [512][512],
 * b2 [512], w3 [126][512], b3 [126] = the VPoser decoder's Linear layers (host, PyTorch layout).
 * ap_fit_run: iterations first_iter .. first_iter + n_iters - 1 of the loop (iteration index matters: the script halves
 * the hip confidences every iteration; z joins the optimised set, with a fresh Adam, at switch_iter = 100).  State
 * arrays are device pointers updated in place; j2d [2][L][2][24][3] (x, y, confidence), intr [2][4] = fx, fy, cx, cy,
 * extr [2][3][4] device; robust_host [L] host ints (frame mask).  loss_hist (optional, device [n_iters][L][4]): per-frame
 * loss parts before each step; grad_out (optional, device [L*32 + 2*L*9 + 10]): the last gradient vector
 * [dz | dphi | dtau | dbeta] (with lr = 0 a gradient probe). */
typedef struct ap_fit ap_fit;
int ap_fit_create(ap_fit** out, const ap_smplx* body, const float* w1, const float* b1, const float* w2, const float* b2,
                  const float* w3, const float* b3, int device);
void ap_fit_destroy(ap_fit* h);
int ap_fit_run(ap_fit* h, int L, float* z, float* phi, float* tau, float* beta, const float* j2d, const int* robust_host,
               const float* intr, const float* extr, int first_iter, int n_iters, int switch_iter, float lr, float sigma,
               float w_vposer, float w_temporal, float* loss_hist, float* grad_out, void* stream);

/* The network's input contract computed on the GPU (aerialpeople.py:125-141,174 + resize_with_pad,
 * utils/utils.py:214-235): frame[:, :, ::-1] / 255 -> crop -> cv2.resize semantics for float images (INTER_LINEAR,
 * half-pixel centres, border clamp) to int(scale*w) x int(scale*h), scale = 224 / max(h, w) -> centred zero padding to
 * 224 x 224 -> CHW -> Normalize(ImageNet mean, std).
 * frames: n uint8 HWC images [H][W][3] on the device, frame i at frames + i*frame_stride_bytes (0 = one shared frame);
 * bgr = 1 reverses the channel order (cv2.imread frames).  crop [n][4] = y0, y1, x0, x1 (device ints; y1/x1 exclusive,
 * inside the frame, non-empty).  out [n][3][224][224]; scale_out [n] = the resize scale (bb's third component),
 * pad_left_top_out [n][2]. */
int ap_preprocess_crops(const unsigned char* frames, int64_t frame_stride_bytes, int n, int H, int W, int bgr,
                        const int* crop_y0y1x0x1, float* out_nchw, float* scale_out, int* pad_left_top_out, void* stream);
int ap_rot6d_to_rotmat(const float* x6, int n, float* rotmat, void* stream);          /* [n][6] -> [n][3][3] */
/* tgm.rotation_matrix_to_angle_axis (torchgeometry 0.1.2) as called for pred_angles at copenet_twoview.py:323-324:
 * [n][3][cols] row-major (cols = 3, or 4 for the caller's zero-padded 3x4 input) -> [n][3] */
int ap_rotmat_to_angle_axis(const float* rotmat, int n, int cols, float* angle_axis, void* stream);
/* Axis-angle -> rotation matrix, [n][3] -> [n][3][3].  variant 0: smplx lbs.batch_rodrigues (Rodrigues formula with
 * angle = |r + 1e-8|), the function the reference's dataset code takes from the body-model package
 * (copenet/dsets/aerialpeople.py:177) and SMPLX.forward(pose2rot=True) applies to its pose inputs; variant 1:
 * copenet/utils/geometry.py:9-45 batch_rodrigues (through a unit quaternion). */
int ap_batch_rodrigues(const float* angle_axis, int n, int variant, float* rotmat, void* stream);
int ap_transform_points(const float* rt, const float* pts, int B, int P, float* out, void* stream); /* rt [B][3][4] */
int ap_perspective_projection(const float* pts, int B, int P, const float* rotation, const float* translation,
                              float fx, float fy, const float* center, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AIRPOSE_HIP_H */
