// Internal launch interface between the C-ABI layer (api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ConvArgs {
    const void* x;        // NHWC activations [N][H][W][ldx], element type T
    const void* w;        // [Cout_pad][KH*KW*Cin] (row stride wld), element type T
    const float* scale;   // [Cout_pad]
    const float* shift;   // [Cout_pad]
    const void* res;      // optional residual [M][ldr], element type T
    void* y;              // [M][ldy]
    const void* zero;     // >= 16 bytes of zeros (DMA source of predicated-off rows; conv_pipe only)
    unsigned long long* dbg;  // optional per-phase cycle stamps (profiling builds; NULL otherwise)
    // optional second K segment (downsample branch folded into conv3): after the KH*KW*Cin regular columns the
    // contraction continues over Cin2 channels of x2 sampled at pixel (ho*stride2, wo*stride2) -- a 1x1 conv whose
    // weights follow in the same packed row (wld = KH*KW*Cin + Cin2)
    const void* x2;
    int H2, W2, Cin2, stride2, ldx2;
    int N, H, W, Cin;
    int Ho, Wo, Cout;     // Cout: number of stored channels (multiple of 16 B / sizeof(T))
    int KH, KW, stride, pad;
    int M;                // N*Ho*Wo
    int ldx, ldy, ldr, wld;
    int relu;
    int mtiles, ntiles;   // filled by the launcher
    unsigned magic_hw, magic_w;   // conv_slab, filled by its launcher: ceil(2^32 / (H W)), ceil(2^32 / W) when (M + tile) * H * W < 2^32 (0 = divide)
    int out_f32;          // split-bf16 kind only: y (and res) are plain fp32 [M][ldy] instead of split pairs
    int y_tiled;          // 16-bit kinds, LDS-staged epilogues: y in the fragment-tiled layout [M/16][Cout/8][16 pixels][8 channels]
                          // (the t2 operand of the fused pair kernel) instead of NHWC rows; Cout % 8 == 0, y sized for M rounded up to 16
    // conv_lean.hip POOL variant (last convolution of the trunk): when set, y is NOT written; the 7 x 7 pixels of every image are
    // averaged per channel instead (AvgPool2d(7) + view) into pool_out [N][Cout] fp32, bit-identical to conv + avgpool_kernel
    float* pool_out;
    int* range_flag;      // fp16 storage, or NULL: host-mapped word set to 1 when a stored value leaves the fp16 range (ap_common.h)
};

// (launchers of the 16-bit-flavoured kernel sources: kernels_h16.inc, included at the end of this file)

// ---- conv3 (+ identity | + folded downsample) of a block + conv1 of the next block in one pixel-local kernel (conv_pair.hip)
struct PairArgs {
    const void* t2;               // [M][P] bf16: conv2 output of block k
    const void* res;              // identity block: [M][C3] bf16, the input of block k
    const void* wstream;          // weight tiles in consumption order (ap_launch_pair_pack)
    const float *s3, *h3;         // BatchNorm scale / shift of conv3 [C3]
    const float *s1, *h1;         // ... of the next block's conv1 [N1]
    void* out;                    // [M][C3] bf16: output of block k
    void* t1n;                    // [M][N1] bf16: conv1 output of block k+1 (N1 > 0)
    int M;                        // output pixels (N*Ho*Wo)
    // stage-first block: second K segment = the block input x2 [N][H2][W2][P2] sampled at (ho*stride2, wo*stride2)
    const void* x2;
    int Ho, Wo, H2, W2, stride2;
    int t2_tiled, res_tiled, out_tiled;   // t2 / res / out in the fragment-tiled layout [M/16][C/8][16][8] instead of NHWC rows
    int out_even;                 // `out` (NHWC only) is stored at the even (ho, wo) pixels only: its one reader is a stride-2 1x1; needs Ho, Wo
    unsigned long long* dbg;      // optional cycle stamps (AP_TRACE builds; NULL otherwise)
    int* range_flag;              // fp16 storage, or NULL: host-mapped word set when a stored value leaves the fp16 range
};

// ---- fused layer1 bottleneck (bottleneck.hip); bf16 only
struct BneckArgs {
    const void* x;                // [N][H][W][cin] bf16 (cin = 256, or 64 for the downsample block)
    void* y;                      // [N][H][W][256] bf16
    const void *w1, *w2, *w3;     // packed bf16 rows: [64..][cin], [64..][9*64], [256][64 | 128]
    const float *s1, *h1, *s2, *h2, *s3, *h3;   // BatchNorm scale / shift per conv
    const void* zero;             // >= 16 bytes of zeros
    unsigned long long* dbg;      // optional cycle stamps (AP_TRACE builds; NULL otherwise)
    int* range_flag;              // fp16 storage, or NULL: host-mapped word set when a stored value leaves the fp16 range
    int N, H, W;                  // H, W multiples of 14
    int tiles_x, tiles_per_img, total;   // filled by the launcher
    // identity block only (tail variant): conv1 of the NEXT block (256 -> 128, 1x1, model_copenet.py:29-31 of layer2.0) on the
    // block output while it is still in registers -- t1n [N][H][W][128]; with y_even the block output itself is stored at the
    // even (row, column) pixels only: all a stride-2 downsample branch (model_copenet.py:97-102) ever reads of it
    const void* w1n;              // packed rows [128][256], or NULL: plain block
    const float *s1n, *h1n;       // BatchNorm scale / shift of that conv1 [128]
    void* t1n;
    int y_even;
};

// ---- image-resident layer3 identity bottleneck (block_img.hip); 16-bit storage
struct BlkImgArgs {
    const void* x;                // [N][14][14][1024] NHWC
    void* y;                      // [N][14][14][1024]
    const void* wfrag;            // the three weight matrices as per-wave MFMA-fragment streams (ap_launch_block_img_pack)
    const float *s1, *h1, *s2, *h2, *s3, *h3;   // BatchNorm scale / shift per conv
    int N;
    int* range_flag;              // fp16 storage, or NULL
    unsigned long long* dbg;      // optional cycle stamps (AP_TRACE builds; NULL otherwise)
};

// ---- stride-1 3x3 + BN + ReLU of layer2 (128 -> 128 at 28 x 28), half an image resident in LDS (conv_img3.hip); 16-bit storage
struct ConvImg3Args {
    const void* x;                // [N][28][28][128] NHWC
    void* y;                      // [N][28][28][128] NHWC, or fragment-tiled (y_tiled)
    const void* wfrag;            // the weights as per-wave MFMA-fragment streams (ap_launch_conv_img3_pack)
    const float *scale, *shift;   // BatchNorm
    const void* zero;             // 256 bytes of zeros (DMA source of the slots outside the image)
    int N;
    int y_tiled;
    int nhalf_pad;                // filled by the launcher
    int* range_flag;              // fp16 storage, or NULL
};

// ---- stride-2 3x3 of layer2.0 in polyphase form, a quarter of an output image per workgroup (conv_s2p.hip); 16-bit storage
struct ConvS2pArgs {
    const void* x;                // [N][56][56][128] NHWC
    void* y;                      // [N][28][28][128] NHWC, or fragment-tiled (y_tiled)
    const void* wfrag;            // the weights as per-wave MFMA-fragment streams (ap_launch_conv_s2p_pack)
    const float *scale, *shift;   // BatchNorm
    const void* zero;             // 256 bytes of zeros (DMA source of the slots outside the image)
    int N;
    int y_tiled;
    int nunits_pad;               // filled by the launcher
    int* range_flag;              // fp16 storage, or NULL
};

// ---- pointwise convolution + BN (+ identity) + ReLU on the one-wave-per-SIMD mainloop (conv_pw.hip); 16-bit storage
struct PwArgs {
    const void* x;                // [M][Cin] NHWC pixel rows
    void* y;                      // [M][Cout]
    const void* res;              // [M][Cout] identity, or NULL
    const void* wfrag;            // the weight matrix as per-wave MFMA-fragment streams (ap_launch_conv_pw_pack)
    const float *scale, *shift;   // BatchNorm scale / shift [Cout]
    int M, Cin, Cout, relu;       // relu must be 1
    // second K segment (a stage's first block: the downsample branch folded into conv3, model_copenet.py:41-42, 97-102): x2
    // [N][H2][W2][Cin2] sampled at (ho * stride2, wo * stride2); Ho * Wo must divide 196.  NULL: none
    const void* x2;
    int Cin2, Ho, Wo, H2, W2, stride2;
    // k3 = 1: x is [N][H2][W2][Cin] and the convolution a 3 x 3 with stride stride2 and padding 1 onto Ho x Wo (conv2 of a stage's
    // first block, model_copenet.py:32-34 with :18): K = [tap][Cin] as the weight rows have it, a chunk of 64 channels of ONE tap
    // per staging step (im2col by address; the out-of-image taps of the top row / left column are zeroed on their way into the LDS)
    int k3;
    int* range_flag;              // fp16 storage, or NULL
};

// ---- stem / pooling (stem.hip)

// ---- regressor glue (regressor.hip); all fp32
struct RegInitArgs {
    const float *pos0, *pos1;          // [B][3]
    const float *theta0, *theta1;      // [tb][>=132] (tb = 1 or B) or NULL -> mean pose
    const float *shape0, *shape1;      // [sb][10] or NULL -> mean shape
    int theta0_bs, theta1_bs, shape0_bs, shape1_bs;   // batch strides in floats (0 = broadcast)
    int pos_bs;                        // batch stride of pos0/pos1 in floats
    int rows;                          // 2B (two views) or B (single view: only the *0 inputs are read)
    const float* mean_pose;            // [144]
    const float* mean_shape;           // [10]
    float* state;                      // [2B][148]: pos3 | orient6 | art126 | shape10 | pad3
    int B;
};
hipError_t ap_launch_reg_init(const RegInitArgs& a, hipStream_t st);
// folded IEF fast path: feature GEMM (split-K partial sums into `part`, ap_reg_fold_part_floats(rows) floats) and ONE
// kernel for initialisation + all iterations + output.  wt_feat [2048][148], wt_state [284][148] (k-major), bias [148]
hipError_t ap_launch_reg_fold_ief(const RegInitArgs& a, const float* xf0, const float* xf1, const float* bb0,
                                  const float* bb1, const float* partner, int partner_ld, const float* wt_feat,
                                  const float* wt_state, const float* bias, float* part, int iters, int two_view,
                                  float* pose0, float* betas0, float* pose1, float* betas1, hipStream_t st);
int ap_reg_fold_part_floats(int rows);
hipError_t ap_launch_word_copy(const int* src, int* dst, hipStream_t st);                 // host-mapped word -> host-mapped word, stream-ordered
hipError_t ap_launch_probe_inputs(float* x, size_t n, float* bb, int rows, uint64_t seed, hipStream_t st);   // ap_net_parity_probe's batch
// view-split step in two halves (regressor.hip): feature part once per forward, the 148 partner-independent state columns, then
// the 136 partner columns + residual; hfeat / partial: [B][148] fp32
hipError_t ap_launch_reg_feat_part(const float* xf, int B, const float* wt_feat, const float* bias, float* part, float* hfeat,
                                   hipStream_t st);
hipError_t ap_launch_reg_step_local(const float* hfeat, const float* bb, const float* pose_in, const float* betas_in, int B,
                                    const float* wt_state, float* partial, hipStream_t st);
hipError_t ap_launch_reg_step_finish(const float* partial, const float* pose_in, const float* betas_in, const float* partner,
                                     int partner_ld, int B, const float* wt_state, float* pose_out, float* betas_out, hipStream_t st);
// state (+= delta[:, :145] if delta) ; S[row] = [bb3 pos3 orient6 art126 shape10 art_other126 shape_other10 0 0 0 0]
hipError_t ap_launch_reg_update_assemble(float* state, const float* delta, int ldd, const float* bb0, const float* bb1,
                                         const float* partner, int partner_ld,
                                         float* S, int B, int two_view, hipStream_t st);
hipError_t ap_launch_reg_output(const float* state, float* pose0, float* betas0, float* pose1, float* betas1,
                                int B, int two_view, hipStream_t st);

// single-view HMR head glue (regressor.hip): state rows of 160 floats = pose132 | shape10 | cam3 | pad
hipError_t ap_launch_hmr_init(const float* theta, int theta_bs, const float* shape, int shape_bs, const float* cam,
                              int cam_bs, const float* mean_pose, const float* mean_shape, const float* mean_cam,
                              float* state, int B, hipStream_t st);
hipError_t ap_launch_hmr_update(float* state, const float* delta, int ldd, int B, hipStream_t st);
hipError_t ap_launch_hmr_output(const float* state, float* rotmat, float* betas, float* cam, int B, hipStream_t st);

// ---- SMPL-X (smplx.hip)
struct SmplxModelDev {
    int V, J, K;                  // vertices, joints (55), bones per vertex
    int ncoef;                    // columns of the coefficient matrix (512)
    int coef_split;               // coefficient rows / blend-shape operand held as split-bf16 pairs (default) or fp32
    int ldv;                      // row stride of v_posed workspace (floats)
    const float* j_template;      // [J][3]
    const float* j_shapedirs;     // [J][3][20]
    const int* parents;           // [J]
    const int* depth;             // [J]
    int max_depth;
    const int* skin_idx;          // [V][K]
    const float* skin_w;          // [V][K]
    const int* extra_verts;       // [21]
    const int* lmk_tri;           // [51][3]
    const float* lmk_bary;        // [51][3]
    int n_extra, n_lmk;
    // fused contraction + skinning (smplx_lbs_fused_kernel)
    const void* dirs_frag;        // blend-shape directions as split-bf16 MFMA A fragments in register order:
                                  // [vertex group of 16][K step of 32 (8)][x, y, z][hi, lo][lane 64][8 bf16]
    const float* v_template;      // [V padded to 16][3]
    const uint32_t* skin_idx8;    // [V padded to 16]: the 4 bone indices of a vertex, 6 bits each (K == 4, J <= 64) | (joint-vertex slot + 1) << 24
    const float* skin_w4;         // [V padded to 16][4]
    const int* jv_slot;           // [V]: slot of the vertex in the joint-vertex side buffer, -1 = none
    int n_jv;                     // slots (distinct vertices among the 21 picks and the 51 landmark triangles)
    const float4* jt_pack;        // [n_extra + n_lmk][6]: per output joint {slot x3 | packed bone ids x3 | weights x3 | barycentric} of its three corner vertices
    // body-only calls (no hand / jaw / eye pose: joints nb .. J-1 keep the identity rotation, so A_j == A_rep(j) for the nearest
    // posed ancestor rep(j) < nb: G_j = G_p [I | J_j - J_p] and A_j = G_j [I | -J_j] = G_p [I | -J_p]): the skin table over the nb
    // transforms that differ, duplicate bones merged (weights summed), heaviest first, zero weights last
    int nb;                       // 22 = root + 21 body joints (0 = no such table)
    const uint32_t* skin_idx8b;   // as skin_idx8 with bone indices < nb
    const float* skin_w4b;
};
struct SmplxFwdArgs {
    int n;                        // bodies
    const float* betas;           // [n][10]
    const float* expression;      // [n][10] or NULL
    // pose input, one of:
    const float* pose6d;          // [n][pose6d_ld]: 22 x 6D (root first); root becomes the post rotation
    int pose6d_ld;
    const float* global_orient;   // [n][9] or NULL (identity)
    const float* body_pose;       // [n][21][9]
    const float* extra_pose;      // [n][33][9] or NULL (identity)
    const float* transl;          // [n][3] or NULL
    // two-view pipeline entry (ap_smplx_fwd_twoview): bodies [0, n_main) are the regressed ones; bodies [n_main, n) are
    // the test-mode "input" meshes = body (b - n_main) with betas = 0, identity root and translation in_trans
    int n_main;                   // 0 = all bodies are regular
    const float* in_trans;        // [n - n_main][3]
    float* pose_rw;               // pred_pose base for the in-place translation un-scale, or NULL
    float trans_scale;
    const float *intr0, *intr1;   // [n_main / 2][9] each: camera centre = intr[:, :2, 2] of the body's view, or NULL
    float* cc_ws;                 // [n][2] resolved camera centres (workspace)
    // post transform (transform_smpl): X' = R X + t
    const float* post_rt;         // [n][12] (3x4 row-major) or NULL
    const float* post_t;          // with pose6d: translation [n][post_t_ld]
    int post_t_ld;
    // projection
    const float* cam_center;      // [n][2] or NULL
    float fx, fy;
    // workspace
    float* coef;                  // [n][ncoef]
    float* A;                     // [n][J][12]
    float* A22;                   // [n][22][12] or NULL: post transform o A_j of the 22 posed joints (what the fused kernel's merged form skins with)
    float* jposed;                // [n][J][3]
    float* post;                  // [n][12] resolved post transform
    const float* vposed;          // [n][ldv]
    // outputs
    float* vertices;              // [n][V][3]
    float* joints;                // [n][J+21+51][3]
    float* joints2d;              // [n][127][2] or NULL
    float* rotmat_out;            // [n][22][9] or NULL (pose6d mode)
    unsigned long long* dbg;      // optional cycle stamps (AP_TRACE builds of smplx.hip; NULL otherwise)
    float* vp_side;               // fused path: v_posed of the joint vertices [n][n_jv][3]; NULL: the joints kernel reads vposed
    int* grp_cnt;                 // fused path, second cut: per body group of 32 an arrival counter (zero between launches).  When set,
                                  // vp_side holds the SKINNED joint vertices and the group's last workgroup computes the joints /
                                  // landmarks / projection itself: no joints launch
};
hipError_t ap_launch_smplx_prep(const SmplxModelDev& m, const SmplxFwdArgs& a, hipStream_t st);
hipError_t ap_launch_smplx_skin(const SmplxModelDev& m, const SmplxFwdArgs& a, hipStream_t st);
hipError_t ap_launch_smplx_joints(const SmplxModelDev& m, const SmplxFwdArgs& a, hipStream_t st);
// blend-shape contraction + skinning in one kernel (K = 4 bones per vertex, body-only pose feature, split-bf16 coefficients)
bool ap_smplx_lbs_fused_supported(const SmplxModelDev& m);
size_t ap_smplx_dirs_frag_bytes(int V);
hipError_t ap_launch_smplx_lbs_fused(const SmplxModelDev& m, const SmplxFwdArgs& a, int n_cu, int merged, hipStream_t st);

// ---- stand-alone geometry helpers (smplx.hip)
hipError_t ap_launch_rot6d(const float* x6, int n, float* R, hipStream_t st);
hipError_t ap_launch_rotmat_to_angle_axis(const float* R, int n, int ld, float* out, hipStream_t st);
hipError_t ap_launch_batch_rodrigues(const float* aa, int n, int variant, float* R, hipStream_t st);
hipError_t ap_launch_transform_points(const float* rt, const float* pts, int B, int P, float* out, hipStream_t st);
hipError_t ap_launch_projection(const float* pts, int B, int P, const float* R, const float* t, float fx, float fy,
                                const float* center, float* out, hipStream_t st);

// ---- AirPose+ fitting loop (fitting.hip); all fp32
struct FitArgs {
    int L;                        // frames of the sequence
    // VPoser decoder output / its gradient, [L][ldo] (126 used), and pose_body of every frame [L][21][3]
    const float* O; float* dO; int ldo;
    const float* aa_all;
    // optimised state: z [L][32], phi [2][L][6], tau [2][L][3], beta [10]; gradients alongside
    float *z, *phi, *tau, *beta;
    const float* dz; int ldz;
    float *dphi, *dtau, *dbeta_part;   // dbeta_part [L][10] (summed by the Adam kernel)
    float* loss_part;             // [L][4]: 2-D | temporal(pose_body) | temporal(phi, tau) | -
    // body model: rest joints J = j_template + j_shapedirs beta (first 24 chain joints)
    const float* j_template; const float* j_shapedirs; int jsd_ld;
    // observations: j2d [2 views][L][2 detectors][24][3 = x, y, conf], robust [L], intr [2][4], extr [2][12]
    const float* j2d; const int* robust; int n_robust, n_pairs;
    const float* intr; const float* extr;
    float sigma, w_temporal, w_vposer, lr;
    float *adam_m, *adam_v;       // [L*32 + 2*L*9 + 10]
    float* grad_out;              // optional copy of the gradient vector (tests)
};
hipError_t ap_launch_fit_linear(const float* X, int ldx, int K, const float* Wt, int ldw, const float* bias, const float* G,
                                int ldg, float* Y, int ldy, int L, int N, int act, hipStream_t st);
hipError_t ap_launch_fit_aa(const float* O, int ldo, float* aa, int L, hipStream_t st);
hipError_t ap_launch_fit_frame(const FitArgs& a, int it, hipStream_t st);
hipError_t ap_launch_fit_adam(const FitArgs& a, int step, int with_z, hipStream_t st);
// fused decoder passes: one launch for z -> H1, H2, O, pose_body; one for dO -> dz followed by Adam on every quantity
hipError_t ap_launch_fit_decode(const float* z, int L, const float* w1t, const float* b1, const float* w2t, const float* b2,
                                const float* w3t, const float* b3, float* H1, float* H2, float* O, float* aa, hipStream_t st);
hipError_t ap_launch_fit_backprop_adam(const FitArgs& a, const float* w3, const float* w2, const float* w1, const float* H1,
                                       const float* H2, int step, hipStream_t st);

// ---- launchers that exist once per 16-bit storage flavour (ap_common.h: AP_NS).  A kernel source sees its own set; api.hip
// (AP_API_TU) sees both and picks by the handle's precision
#ifdef AP_API_TU
namespace k_bf16 {
#include "kernels_h16.inc"
}
namespace k_f16 {
#include "kernels_h16.inc"
}
#else
namespace AP_NS {
#include "kernels_h16.inc"
}
#endif
