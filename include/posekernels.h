/* libposekernels — C-ABI of the MI355X (gfx950) pose-estimation hot path.
 *
 * The reference (MarkJhonBao/InfantPoseEstimation_GaussianBias) is pure Python/PyTorch and has NO
 * FFI/plugin boundary of its own; the Python class/function surface is its interface.
 * This header is therefore the boundary this build defines underneath that surface: every entry point
 * replaces one group of ATen calls made by a reference function, cited as (file:line) below.
 *
 * Conventions (all entries):
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - every pointer is a BORROWED DEVICE pointer (caller allocates inputs, outputs and workspaces);
 *     the library allocates nothing and keeps no state besides a thread-local error string.  It reads four environment
 *     switches, per call, that route convolutions to / away from the two specialised kernels (PK_CONV8P, PK_CONV8P_MIN_TILES,
 *     PK_CONV3H, PK_CONV3H_MIN_TILES: used by the parity tests to push small shapes through them); the tuning knobs of the
 *     measurement rounds are compile-time constants unless the library is built with `make TUNING=1`.
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no internal synchronisation; graph-capturable.
 *   - returns 0 on success, a negative PK_ERR_* on argument validation failure (nothing launched),
 *     or a positive hipError_t from the launch.  Never throws, never aborts.
 *   - layouts: "maps" are (B,K,H,W) fp32 contiguous; features are NHWC; bf16 is the 16-bit upper half of fp32.
 */
#ifndef POSEKERNELS_H
#define POSEKERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_OK 0
#define PK_ERR_INVALID (-1)     /* bad shape / null pointer / misalignment */
#define PK_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for */

#define PK_DT_F32 0
#define PK_DT_BF16 1

int pk_version(void);                      /* 10000*major + 100*minor + patch */
const char* pk_last_error_string(void);    /* thread-local; valid until the next failing call on this thread */
int pk_marker(int id, void* stream);       /* profiling aid: empty launch of `id` workgroups that cuts a kernel trace into sections */

/* ---- T1: COCOPoseDataset._generate_target (datasets/coco_dataset.py:185-250), batched ------------------
 * keypoints (B,K,2) f32 input-px; visible (B,K) f32 (COCO v: 0/1/2); lut = host-built patch values indexed
 * by integer d2=dx^2+dy^2 (lut_len entries; built with the reference's own float32 numpy expression so the
 * result is bit-exact); patch side n, centre c; reach = 3*sigma; stride = input/heatmap size (double).
 * Writes EVERY element of target (B,K,Hh,Wh) and weight (B,K,1).                                          */
int pk_gaussian_target(const float* keypoints, const float* visible, const float* lut, int lut_len,
                       float* target, float* weight, int B, int K, int Hh, int Wh,
                       double stride_x, double stride_y, double reach, int patch_n, int patch_c, void* stream);

/* ---- T2: GenerateTarget (data/pose_transforms.py:385-457): dense sub-pixel Gaussian, weight 0/1 -------- */
int pk_dense_target(const float* keypoints, const float* visible, float* heatmaps, float* weights,
                    int B, int K, int Hh, int Wh, float scale_x, float scale_y, float sigma, void* stream);

/* ---- D2/D3: argmax decoders.  mode 0: get_max_preds (utils/postprocess.py:10-34); mode 1: quarter shift of
 * PoseEstimator.decode_heatmaps (models/pose_estimator.py:331-373); mode 2: Taylor sub-pixel of
 * get_max_preds_with_subpixel (utils/postprocess.py:37-75).  index: first maximum (ties -> lowest index).    */
int pk_argmax_decode(const float* heatmaps, int32_t* index, float* maxval, float* coords,
                     int BK, int H, int W, int mode, void* stream);

/* ---- D1: HeatmapRegressionHead.decode (models/fusion_head.py:309-365) = SoftArgmax2D (:24-71) +
 * LocalGaussianRefinement (:74-128) + alpha blend (:169-170) + offset sampling.  alpha/fusion_weight are the
 * RAW parameters (sigmoid applied inside), read from device memory.  offsets may be NULL (apply_offset=False). */
int pk_softargmax_refine_decode(const float* heatmaps, const float* offsets, const float* alpha_param,
                                const float* fusion_weight_param, float* coords, float* scores,
                                int BK, int H, int W, int local_radius, void* stream);

/* ---- D3 remainder (utils/postprocess.py:78-184,226-238,270-292) on (B,K,2) coordinates ----------------- */
int pk_window_refine(const float* heatmaps, const float* coords_in, float* coords_out, int BK, int H, int W,
                     int window, void* stream);                                  /* coordinate_refinement */
int pk_fused_blend(const float* hp, const float* maxvals, const float* regression, float* out, int BK,
                   float sx, float sy, float reg_scale, void* stream);           /* fused_decode tail */
int pk_affine_coords(const float* coords, const float* center, const float* scale, float* out, int B, int K,
                     float mul_x, float mul_y, const float* mask_maxvals, float threshold, void* stream);
                     /* out = mask * coords * (scale*mul) + center - scale/2 : transform_preds / train.py:328-336 */

/* ---- f1: COCOEvaluator.update arrays (utils/metrics.py:84-106): records (B,K,3) = [x, y, score], instance_score (B) = mean of
 * the strictly positive keypoint scores (0 if none) ----------------------------------------------------------------------- */
int pk_pose_records(const float* keypoints, const float* scores, float* records, float* instance_score, int B, int K,
                    void* stream);

/* ---- f2: input pipeline (datasets/transforms.py:42-47,128-131,212-217; datasets/coco_dataset.py:156-163; inference.py:83-110):
 * affine crop (OpenCV 8-bit warpAffine INTER_LINEAR / BORDER_CONSTANT 0 integer algorithm, see oracle/warp.py; parity unpinned vs cv2
 * itself) + optional column mirror + optional BGR->RGB + ((v/255) - mean)/std, whole batch, one launch.  src_u8: device buffer holding the
 * decoded (H,W,3) uint8 images; desc_table rows (72 bytes): { int64 src_offset; int32 H, W, flip, bgr; double minv[6] } with minv the
 * INVERSE (destination->source) matrix in float64; mean3/std3 are HOST pointers (read before the launch returns).  Outputs (either may be
 * NULL): fp32 NCHW (B,3,out_h,out_w) -- the reference's batch['img'] -- and bf16 NHWC with 8-channel pixels (B,out_h,out_w,8), the stem
 * convolution's input layout (channels 3..7 zero).                                                                                 */
int pk_affine_crop_normalize(const void* src_u8, const void* desc_table, int n_samples, int out_w, int out_h, float* out_nchw_f32,
                             void* out_nhwc8_bf16, const float* mean3, const float* std3, void* stream);

/* ---- D4: flip-test merge (models/pose_estimator.py:303-319): out = (a + swapLR(flipW(b)))/2 ------------- */
int pk_flip_merge(const float* a, const float* b_flipped, const int32_t* partner, float* out,
                  int B, int K, int H, int W, void* stream);

/* ---- video post-processing (utils/postprocess.py:187-267) -------------------------------------------------------
 * pk_temporal_smooth: coords/out (T, C=2K) fp32, weights: `window` doubles as np.convolve receives them (the kernel is flipped
 * like np.convolve does; edge padding window/2 on both sides; float64 accumulation, float32 result); window must be odd.
 * pk_nms_pose: greedy in-sample suppression, preds (B,K,2), maxvals (B,K), out (B,K,2) = preds * keep, keep (B,K) uint8.     */
int pk_temporal_smooth(const float* coords, float* out, const double* weights, int T, int C, int window, void* stream);
int pk_nms_pose(const float* preds, const float* maxvals, float* out, uint8_t* keep, int B, int K, float distance_threshold,
                void* stream);

/* ---- L1/L2: FusionPoseLoss.forward (models/fusion_head.py:745-806 with :405-559, :637-743) ---------------
 * stats: workspace of B*K*PK_LOSS_STAT floats + B*16*4 floats + 16 floats (see PK_LOSS_WS_FLOATS);
 * losses: 7 floats (heatmap, offset, peak, variance, overlap, shape, total — already multiplied by lambdas).
 * lambdas7: device array of 7 floats = the six term weights + the overlap threshold (GaussianDistributionConstraint, default 0.5). */
#define PK_LOSS_STAT 24
#define PK_LOSS_WS_FLOATS(B, K) ((B) * (K) * PK_LOSS_STAT + (B) * 16 * 4 + 16)
int pk_fusion_loss_fwd(const float* heatmaps, const float* offsets, const float* variances, const float* target,
                       const float* weight, const float* gt_keypoints, float* ws, float* losses,
                       int B, int K, int H, int W, float in_w, float in_h, float sigma_t,
                       const float* lambdas7, int use_target_weight /* 0: heatmap/offset/peak terms are plain (B,K) means */, void* stream);
int pk_fusion_loss_bwd(const float* heatmaps, const float* offsets, const float* variances, const float* target,
                       const float* weight, const float* ws, const float* grad_total /*device scalar or NULL=1*/,
                       float* d_heatmaps, float* d_offsets, float* d_variances,
                       int B, int K, int H, int W, float sigma_t, const float* lambdas7, void* stream);

/* The same terms with COORDINATES HANDED IN by the caller: the public methods GaussianDistributionConstraint.compute_heatmap_variance /
 * variance_alignment_loss / spatial_overlap_loss / distribution_shape_loss / forward (models/fusion_head.py:405-575) and
 * FusionPoseLoss.heatmap_loss / offset_loss / peak_localization_loss (:637-743) take any (B,K,2) coordinates, not only the map's own
 * soft-argmax.  coords == NULL is pk_fusion_loss_fwd.  sigma (B,K) or NULL receives compute_heatmap_variance's result.  Backward:
 * d_coords (B,K,2) receives the coordinate gradient (NULL: the coordinates were the soft-argmax, their gradient flows into d_heatmaps);
 * grad_sigma (B,K) or NULL is an upstream gradient on `sigma`.  A term is selected by its lambda (0 elsewhere).                         */
int pk_fusion_terms_fwd(const float* heatmaps, const float* offsets, const float* variances, const float* target,
                        const float* weight, const float* gt_keypoints, const float* coords, float* ws, float* losses, float* sigma,
                        int B, int K, int H, int W, float in_w, float in_h, float sigma_t,
                        const float* lambdas7, int use_target_weight, void* stream);
int pk_fusion_terms_bwd(const float* heatmaps, const float* target, const float* ws, const float* grad_total,
                        const float* grad_sigma, const float* grad_losses /* 7 upstream gradients of `losses` or NULL (= total only) */,
                        float* d_heatmaps, float* d_offsets, float* d_variances, float* d_coords,
                        int B, int K, int H, int W, float sigma_t, const float* lambdas7, void* stream);
/* LocalGaussianRefinement.forward (models/fusion_head.py:74-128) about given coordinates; backward of SoftArgmax2D.forward (:24-71)        */
int pk_local_gaussian_refine(const float* heatmaps, const float* coords_in, float* coords_out, int BK, int H, int W,
                             int local_radius, void* stream);
int pk_softargmax_bwd(const float* heatmaps, const float* coords, const float* scores, const float* grad_coords,
                      const float* grad_scores, float* d_heatmaps, int BK, int H, int W, void* stream);

/* ---- L3/L4: KeypointMSELoss (models/pose_estimator.py:102-143) and models/losses.py per-pixel losses ------
 * kind 0: mean(((p-t)*w)^2)  [KeypointMSELoss]; 1: mean(w*(p-t)^2) [FusedPoseLoss mse];
 * 2: mean(w*smoothl1(p-t)) [FusedPoseLoss smoothl1]; 3: JointsMSELoss (0.5*mean((p-t)^2 w^2)).
 * partial: workspace of PK_REDUCE_BLOCKS floats.                                                            */
#define PK_REDUCE_BLOCKS 1024
int pk_pixel_loss_fwd(const float* pred, const float* target, const float* weight, float* partial, float* loss,
                      int B, int K, int HW, int kind, void* stream);
int pk_pixel_loss_bwd(const float* pred, const float* target, const float* weight, const float* grad_out,
                      float* d_pred, int B, int K, int HW, int kind, void* stream);
/* MorphologyShapeLoss statistics (models/losses.py:69-104): mean (BK,2), variance (BK,2) of hm/(sum+1e-8) */
int pk_spatial_stats(const float* heatmaps, float* mean, float* var, int BK, int H, int W, void* stream);
/* backward of pk_spatial_stats (MorphologyShapeLoss, models/losses.py:50-135): grad_mean / grad_var (BK,2) -> grad_heatmaps */
int pk_spatial_stats_bwd(const float* heatmaps, const float* mean, const float* var, const float* grad_mean, const float* grad_var,
                         float* grad_heatmaps, int BK, int H, int W, void* stream);

/* ---- S1: AdamW on one flat fp32 buffer (train.py:55-97 grouping; torch.optim.AdamW arithmetic) -----------
 * decay_mask: 1 byte per element (1 = apply weight decay).  lr/step come from DEVICE scalars so a captured
 * graph can be replayed while the schedule advances.  Also refreshes the bf16 compute copy (may be NULL).   */
int pk_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                  uint16_t* param_bf16, int64_t n, const float* lr_dev, const int32_t* step_dev,
                  float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);


/* ======================= network contractions (bf16 MFMA, fp32 accumulate; NHWC activations) =======================
 * These replace the groups of ATen calls inside the reference's modules; weights are bf16 "compute copies" packed by
 * pk_pack_weights from the fp32 master parameters (which keep the reference's OIHW / (out,in) layouts).            */

/* A5/A6/H1/H2/A4 conv: nn.Conv2d(k=3|1, stride 1|2, padding k/2, bias=False) of models/hrnet.py:24-33,68-79,
 * models/hrformer.py:309-320,548-551, models/fusion_head.py:211-251 (+ their bias'd 1x1 heads).
 * x (B,Hs,Ws,Cin) bf16; w_packed [Cout][k*k][Cin] bf16; out_mode 0: (B,Ho,Wo,Cout) bf16, 1: same fp32,
 * 2: (B,Cout,Ho,Wo) fp32 planes (head outputs).  stats_partial (optional): [pk_conv_stats_tiles(M)][2][Cout] column
 * sums / sums of squares of the fp32 results for train-mode BatchNorm.  dilated_input=1 evaluates the stride-2
 * data-gradient (input rows/cols are the zero-stuffed output gradient; pass flipped weights, mode 1 of pk_pack_weights).
 * act: 0 none, 2 softplus (nn.Softplus of fusion_head.py:250).                                                      */
/* addend (optional, bf16, shape of the output, out_mode 0, no statistics): out = conv(x) + addend.  Used by the data-gradient launch
 * of the first conv of a residual block (hrnet.py:44-52,92-102): the skip connection's gradient is added in the epilogue instead
 * of by a separate elementwise pass over both gradients.                                                                       */
int pk_conv2d_nhwc(const void* x, const void* w_packed, void* out, float* stats_partial, const float* bias,
                   int B, int Hs, int Ws, int Cin, int Cout, int ksize, int stride, int dilated_input, int Ho, int Wo,
                   int act, int out_mode, const void* addend, void* stream);
/* conv -> eval-mode BatchNorm (-> + residual) (-> ReLU) as ONE launch (models/hrnet.py:24-52 / :92-102 in eval mode: the running
 * statistics make the normalisation a per-channel affine map): out = relu?(col_scale[n] * conv(x)[., n] + col_shift[n] + residual),
 * bf16 NHWC in and out, fp32 scale / shift.  Inference only; ksize 1 / 3, stride 1 / 2. */
int pk_conv2d_affine_nhwc(const void* x, const void* w_packed, void* out, const float* col_scale, const float* col_shift,
                          const void* residual, int relu, int B, int Hs, int Ws, int Cin, int Cout, int ksize, int stride, int Ho,
                          int Wo, void* stream);
int pk_conv_stats_tiles(int M);
/* ---- Grouped launches (round 4): the exchange units (models/hrformer.py:420-491 == models/hrnet.py:157-227) are 2-12 small conv + BatchNorm
 * layers per dependency level; each level runs as ONE launch per kernel family.  Descriptor arrays are HOST memory, read before the call
 * returns (they travel in the kernel arguments); every pointer inside is a borrowed device pointer.  At most PK_GROUP_MAX members.       */
#define PK_GROUP_MAX 12
typedef struct PkConvDesc {        /* one pk_conv2d_nhwc (out_mode 0) or pk_conv2d_affine_nhwc problem                                   */
    const void* x; const void* w; void* out;
    float* stats;                  /* [pk_conv_stats_tiles(B*Ho*Wo)][2][Cout] partial statistics, or NULL                                */
    const float* col_scale;        /* eval-mode BatchNorm folded into the epilogue (with bias = shift), or NULL                          */
    const float* bias;
    const void* res;               /* residual (affine form) or addend (data gradient), bf16, shape of out, or NULL                      */
    int B, Hs, Ws, Cin, Cout, ksize, stride, dilated_input, Ho, Wo;
    int act;                       /* 0 none, 3 ReLU after the residual add                                                              */
} PkConvDesc;
int pk_conv2d_group(const PkConvDesc* descs, int n, void* stream);
typedef struct PkBnFwdDesc {       /* one pk_bn_train_fwd problem (finalize + apply in one pass; any row count)                          */
    const void* raw; const float* stats_partial; const float* gamma; const float* beta;
    float* running_mean; float* running_var; int64_t* num_batches_tracked;
    const void* residual; void* y; float* save_mean; float* save_rstd; uint8_t* relu_mask /* optional, see pk_bn_act */;
    int64_t rows; int tiles, C; float momentum, eps; int relu;
} PkBnFwdDesc;
int pk_bn_train_fwd_group(const PkBnFwdDesc* descs, int n, void* stream);
typedef struct PkBnBwdDesc {       /* one pk_bn_bwd problem; partial: [pk_bn_bwd_group_blocks(rows)][2][C] floats                        */
    const void* dy; const void* y_act; const void* raw; const float* save_mean; const float* save_rstd; const float* gamma;
    float* partial; float* dgamma; float* dbeta; void* dx; void* dresidual; const uint8_t* relu_mask /* optional: replaces y_act */;
    int64_t rows; int C; int relu;  /* bit 0: mask by relu_mask / y_act > 0; bit 1: eval-mode statistics (no batch-mean terms)               */
} PkBnBwdDesc;
int pk_bn_bwd_group_blocks(int64_t rows);
int pk_bn_bwd_group(const PkBnBwdDesc* descs, int n, void* stream);
typedef struct PkFuseDesc {        /* one pk_fuse_sum problem; mask_y (optional, shape of out): input 0 counts only where mask_y > 0 (the ReLU
                                    * backward of a fused sum folded into the sum of input gradients)                                      */
    const void* inputs[4]; int in_h[4]; int in_w[4]; int n_inputs;
    void* out; const void* mask_y; int B, H, W, C, relu;
} PkFuseDesc;
int pk_fuse_sum_group(const PkFuseDesc* descs, int n, void* stream);
typedef struct PkUpBwdDesc {       /* one pk_upsample_bilinear_bwd problem; mask_y (optional, shape of dy): dy counts only where mask_y > 0  */
    const void* dy; const void* mask_y; void* dsrc; int B, H, W, Hs, Ws, C;
} PkUpBwdDesc;
int pk_upsample_bwd_group(const PkUpBwdDesc* descs, int n, void* stream);
typedef struct PkWgradDesc {       /* weight-gradient SLABS of one 1x1 stride-1 or 3x3 stride-2 conv (all members of one kind): workspace =
                                    * pk_wgrad_group_slices(...) x N x k*k x Cin floats, layout [S][N][k*k][Cin], reduced by pk_reduce_many */
    const void* x; const void* grad_out; float* workspace;
    int B, Hs, Ws, Ho, Wo, N, Cin, ksize, stride;
} PkWgradDesc;
int pk_wgrad_group_slices(int M, int N, int Cin, int ksize, int stride);
int pk_wgrad_group(const PkWgradDesc* descs, int n, void* stream);
int pk_sizeof_group_desc(int which);   /* sizeof of PkConvDesc (0), PkBnFwdDesc (1), PkBnBwdDesc (2), PkFuseDesc (3), PkUpBwdDesc (4), PkWgradDesc (5) */

/* Rows of the [rows][2][Cout] partial-statistics buffer a pk_conv2d_nhwc launch with bf16 output and statistics writes for this geometry
 * (the 3x3 halo kernel emits one row per 64 padded positions; everything else pk_conv_stats_tiles(B*Ho*Wo)).  Replaces the per-call
 * `torch.var_mean` inside nn.BatchNorm2d of the reference (models/hrnet.py:24-52): pk_bn_finalize sums the rows it is given. */
int pk_conv_stats_rows(int B, int Hs, int Ws, int Cin, int Cout, int ksize, int stride, int Ho, int Wo);

/* A1/A3 linear layers: nn.Linear qkv/proj (models/hrformer.py:167-169,180,197) and Mlp fc1/fc2 (:53-55,58-64).
 * out[o_rowmap[m]] = residual + res_scale[sample] * act(x[a_rowmap[m]] @ w^T + bias).  Row maps implement
 * window_partition / window_reverse (hrformer.py:67-114) inside the GEMM: a_rowmap -1 = zero pad token, o_rowmap -1 =
 * cropped token.  act 1 = exact-erf GELU (nn.GELU, hrformer.py:54).  preact_out saves the pre-GELU values;
 * gelu_grad_of multiplies the result by gelu'(z) (backward through the activation).                                  */
int pk_linear_bf16(const void* x, const void* w, void* out, const float* bias, const void* residual, const float* res_scale,
                   const int32_t* a_rowmap, const int32_t* o_rowmap, void* preact_out, const void* gelu_grad_of,
                   int M, int N, int K, int rows_per_sample, int act, int out_fp32, void* stream);

/* A2 as stand-alone functions: window_partition / window_reverse (models/hrformer.py:67-114) move whole rows through the window row map
 * (window-order token -> pixel row, -1 = zero token appended at the bottom / right).  scatter 0: dst[r] = src[rowmap[r]] (or zeros) =
 * partition; scatter 1: dst[rowmap[r]] = src[r] where rowmap[r] >= 0 = reverse + crop.  Any element type, rows of whole dwords.
 * drop_path (hrformer.py:15-24): out = (x / keep_prob) * mask[sample], fp32.                                                          */
int pk_rows_by_map(const void* src, void* dst, const int32_t* rowmap, int64_t n_rows, int row_bytes, int scatter, void* stream);
int pk_drop_path_f32(const float* x, const float* mask, float* out, int64_t batch, int64_t per_sample, float keep_prob, void* stream);

/* weight gradient of either form: dw = sum_m grad_out[m]^T (x) A(m); workspace = pk_wgrad_slices(...)*N*(k*k*Cin + 1) floats.
 * out_layout 0: [N][k*k][Cin]; 1: OIHW (the reference's nn.Conv2d.weight layout).  Ho == 0 selects the linear form.
 * dbias (optional, n_bias <= N entries): the bias gradient = column sums of the same (gathered, scaled) grad_out rows,
 * accumulated from the tile already staged in LDS.                                                                       */
/* Row maps (linear form): a_rowmap / g_rowmap give the source row of x / grad_out for GEMM row m (-1 = zero row); g_scale multiplies
 * grad_out row r by g_scale[r / g_rows_per_sample] (DropPath).  When the maps are the 7x7 WINDOW PARTITION of a (B, Hs, Ws) token
 * grid (nnops.window_rowmap: m = ((b*nh + wy)*nw + wx)*49 + ty*7 + tx -> pixel (wy*7+ty, wx*7+tx), pad tokens -1), pass B, Hs, Ws
 * with Ho = Wo = 0: the streaming kernel then RECOMPUTES the map per DMA lane instead of loading it (M must equal B*nh*nw*49, and
 * g_rows_per_sample must be Hs*Ws).  With B = Hs = Ws = 0 the map is loaded (any map, slower kernel).                          */
/* dw == NULL: write the slabs only (weight slabs [S][N*k*k*Cin], then bias slabs [S][N] when n_bias > 0) and let the caller
 * reduce them later with pk_reduce_many.                                                                                  */
/* pk_wgrad_slices: the slab count S of the launch pk_wgrad_bf16 will make for the same arguments (Hs = Ws = 0 for the linear
 * form; flags bit 0 = a_rowmap given, bit 1 = g_rowmap given, bit 2 = g_scale given).                                         */
int pk_wgrad_slices(int M, int N, int Cin, int ksize, int stride, int Hs, int Ws, int flags);
int pk_wgrad_bf16(const void* x, const void* grad_out, float* workspace, float* dw, float* dbias, int n_bias,
                  const int32_t* a_rowmap, const int32_t* g_rowmap, const float* g_scale, int g_rows_per_sample, int M, int N,
                  int Cin, int ksize, int stride, int B, int Hs, int Ws, int Ho, int Wo, int out_layout, void* stream);

/* window attention core (models/hrformer.py:183-196): softmax(scale*q k^T + table[index]) v per (window, head);
 * qkv (n_windows*49, 3C) bf16 in the reference's channel order s*C + h*d + e; rel_table (169, heads) fp32.          */
/* head_dim d = C/heads: multiple of 8, <= 64.  softmax_scale <= 0 selects the reference's d^-0.5 (hrformer.py:140); a
 * caller that pads 39-wide heads to 40 passes 39^-0.5 explicitly.                                                        */
int pk_window_attn_fwd(const void* qkv, const float* rel_table, void* out, float* lse, int n_windows, int heads, int C,
                       float softmax_scale, void* stream);
int pk_window_attn_bwd_groups(int n_windows, int heads);
int pk_window_attn_bwd_ws_floats(int n_windows, int heads);  /* size of dbias_partial in floats */
/* fwd_out = the `out` tensor pk_window_attn_fwd produced for the same qkv (delta_i = sum_e dout[i][e] * out[i][e]). */
int pk_window_attn_bwd(const void* qkv, const float* rel_table, const void* fwd_out, const void* dout, const float* lse, void* dqkv,
                       float* dbias_partial, float* dtable, int n_windows, int heads, int C, float softmax_scale, void* stream);

/* ======================= normalisation / elementwise (HBM-bound, NHWC bf16, 16 bytes per lane) ==================== */
int pk_sum_partials(const float* partial, int nb, int K, int stride, float* out, float scale, int accumulate, void* stream);
/* nn.BatchNorm2d train mode (eps 1e-5, momentum 0.1, unbiased running_var): finalize conv-epilogue sums -> scale/shift */
int pk_bn_finalize(const float* stats_partial, int tiles, int C, int count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                   float* scale, float* shift, float* save_mean, float* save_rstd, void* stream);
/* relu_mask (optional, rows*C/8 bytes): one byte per 16-byte chunk of y, bit j = channel j of the chunk is > 0 after the ReLU.  The
 * backward kernels read it instead of the activated output (1/16 of the bytes).                                                    */
int pk_bn_act(const void* x, const float* scale, const float* shift, const void* residual, void* y, int64_t rows, int C,
              int relu, uint8_t* relu_mask, void* stream);   /* y = relu?(x*scale + shift (+ residual)) */
/* Train-mode BatchNorm forward from the conv epilogue's partial statistics (nn.BatchNorm2d.forward in models/hrnet.py:24-52,
 * models/hrformer.py:309-344): pk_bn_finalize + pk_bn_act, as ONE launch for small tensors (tiles <= 128).  scale / shift: [C] workspaces. */
int pk_bn_train_fwd(const void* raw, const float* stats_partial, int tiles, int C, int64_t rows, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps, const void* residual,
                    void* y, float* save_mean, float* save_rstd, float* scale, float* shift, int relu, uint8_t* relu_mask, void* stream);
int pk_bn_bwd_blocks(int64_t rows);                          /* partial needs blocks*2*C floats */
/* pk_bn_bwd `relu`: bit 0 = the forward applied ReLU (mask from relu_mask when given, else from y_act); bit 1 = eval-mode BatchNorm (save_mean / save_rstd hold the
 * running statistics, which are constants: dx = gamma * rstd * g, no batch-mean terms; dgamma / dbeta as in training).           */
int pk_bn_bwd(const void* dy, const void* y_act, const void* raw, const float* save_mean, const float* save_rstd,
              const float* gamma, float* partial, float* sums, float* dgamma, float* dbeta, void* dx, void* dresidual,
              int64_t rows, int C, int relu, const uint8_t* relu_mask, void* stream);
int pk_relu_bwd(const void* dy, const void* y, void* dx, int64_t numel, void* stream);
/* nn.LayerNorm(C, eps 1e-5) over the channel dim of NHWC rows (hrformer.py:240,252,273,288).  C = row width (multiple of 8,
 * <= 1024); C_real (0 = C) = number of real channels when the rows carry zero padding: statistics over the real channels,
 * padded outputs / input gradients are zero.  gamma/beta (and dgamma/dbeta) have C entries, 16-byte aligned; the padded
 * entries are ignored.                                                                                                  */
int pk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* save_mean, float* save_rstd,
                     int64_t rows, int C, int C_real, float eps, void* stream);
int pk_ln_bwd_blocks(int64_t rows);                          /* partial needs blocks*2*C floats */
int pk_layernorm_bwd(const void* dy, const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                     const void* dresidual, void* dx, float* partial, float* dgamma, float* dbeta, int64_t rows, int C, int C_real,
                     void* stream);
/* Fused MLP half of the HRFormer block (hrformer.py:288-291 with Mlp :38-64 and DropPath :15-35), C = 32 / 64:
 *   y = x + row_scale[b] * ( fc2( gelu_erf( fc1( LayerNorm(x; gamma, beta, eps) ) + b1 ) ) + b2 )       x, y: [M][C] bf16
 * ONE launch; the 4C-wide hidden activation stays in registers.  w1 = fc1.weight [4C][C] bf16, w2 = fc2.weight [C][4C] bf16
 * (forward copies); w1_t = [C][4C], w2_t = [4C][C] (data-gradient copies).  row_scale (or NULL) is indexed by row / rows_per_sample.
 * Backward recomputes LayerNorm / fc1 / GELU from x (nothing but x is saved):
 *   pk_ln_mlp_bwd_dx: dx = dy + dLayerNorm(...), ln_partial[pk_ln_mlp_dx_blocks][2][C] = per-workgroup sums for dgamma | dbeta;
 *   pk_ln_mlp_bwd_dw: slabs[(4C / HS) slices][pk_ln_mlp_dw_blocks][pk_ln_mlp_slab_floats] with HS = pk_ln_mlp_hidden_slice(C);
 *                     one slab = [ dW1 rows h0..h0+HS [HS][C] | dW2 columns h0..h0+HS [C][HS] | db1[h0..] [HS] | db2 [C] ] fp32,
 *                     to be summed over the workgroup index (pk_reduce_many layouts 0 / 2); db2 is complete in every slice.   */
int pk_ln_mlp_supported(int C);
int pk_ln_mlp_hidden_slice(int C);
int pk_ln_mlp_slab_floats(int C);
int pk_ln_mlp_dx_blocks(int M, int C);
int pk_ln_mlp_dw_blocks(int M, int C);
int pk_ln_mlp_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2,
                  const float* b2, const float* row_scale, void* y, int M, int C, int rows_per_sample, float eps, void* stream);
/* The same half for wide channels (C = 80 / 128 / 160 / 256 / 320; hrformer.py:262-293 with Mlp :38-64 at stage-3/4 widths and the
 * 8-aligned twin of HRFormer-base :779-825), forward only: fc1 / fc2 weights streamed through LDS in hidden slices of 32.  `c_real` <= C:
 * channels the LayerNorm statistics run over (the padded channels hold zeros).  w1 [hidden][C], w2 [C][hidden] bf16 row-major. */
int pk_ln_mlp_wide_supported(int C, int hidden, int M);   /* M > 0: ... and the launch has enough workgroups to pay (M = 0: built for?) */
int pk_ln_mlp_wide_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2,
                       const float* b2, const float* row_scale, void* y, int M, int C, int c_real, int hidden, int rows_per_sample,
                       float eps, void* stream);
int pk_ln_mlp_bwd_dx(const void* dy, const void* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                     const void* w1_t, const void* w2_t, const float* row_scale, void* dx, float* ln_partial, int M, int C,
                     int rows_per_sample, float eps, void* stream);
int pk_ln_mlp_bwd_dw(const void* dy, const void* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                     const void* w2_t, const float* row_scale, float* slabs, int M, int C, int rows_per_sample, float eps,
                     void* stream);
/* Attention half for the 8-aligned twin of HRFormer-base (hrformer.py:779-825: C = heads x 39, here heads x 40 with zero padding; block
 * :262-286, WindowAttention :174-200), forward only: C = 80 with 2 heads.  wqkv [3][heads][40][C], wproj [C][heads][40] bf16, biases and
 * LayerNorm parameters zero in the padded entries; `c_real` = channels the LayerNorm statistics run over, `softmax_scale` = real
 * head_dim^-0.5.  `n_windows` > 0 in _supported additionally asks whether the launch has enough windows to pay. */
int pk_attn_block_wide_supported(int C, int heads, int n_windows);
int pk_attn_block_wide_fwd(const void* x, const int32_t* rowmap, const float* gamma, const float* beta, const float* rel_table,
                           const void* wqkv, const float* bqkv, const void* wproj, const float* bproj, const float* row_scale,
                           void* y, int n_windows, int windows_per_sample, int heads, int C, int c_real, float softmax_scale,
                           float eps, void* stream);
/* Fused attention half of the HRFormer block (hrformer.py:262-286 with WindowAttention :174-200, window_partition/reverse :67-114),
 * C = 32 / 64, head_dim 32, window 7:   y = x + row_scale[b] * proj( attention( qkv( LayerNorm(x) ) ) )     x, y: [M][C] bf16 pixel rows.
 * rowmap[windows*49]: pixel row of every window token, -1 for the reference's zero-pad tokens (LayerNorm output forced to 0, i.e.
 * q = b_q, k = b_k, v = b_v; attended without a mask, output dropped).  o_save ([windows*49][C] bf16) and lse ([windows][heads][49])
 * may be NULL (inference); training saves them for pk_attn_block_bwd.  row_scale is indexed by window / windows_per_sample.     */
int pk_attn_block_supported(int C, int heads);
int pk_attn_block_fwd(const void* x, const int32_t* rowmap, const float* gamma, const float* beta, const float* rel_table,
                      const void* wqkv, const float* bqkv, const void* wproj, const float* bproj, const float* row_scale,
                      void* y, void* o_save, float* lse, int n_windows, int windows_per_sample, int heads, int C, float eps,
                      void* stream);
/* Backward of the fused attention half: dx = dy + dLayerNorm(dqkv W_qkv) written to the pixel rows (pad tokens dropped); emits
 * dqkv [windows*49][3C] bf16 and the LayerNorm output u in window order [windows*49][C] (zero rows at the pad tokens) for the
 * weight-gradient GEMMs (pk_wgrad_bf16: dW_qkv = dqkv^T u, dW_proj = (s dy)^T o), ln_partial[pk_attn_block_blocks][2][C]
 * (dgamma | dbeta sums) and rpb_partial[pk_attn_block_blocks*4][heads][169] (rel-pos-bias gradient per wave).  wqkv_t = [C][3C],
 * wproj_t = [C][C] data-gradient copies; o_saved / lse from pk_attn_block_fwd.                                                */
int pk_attn_block_blocks(int n_windows);
int pk_attn_block_bwd(const void* dy, const void* x, const int32_t* rowmap, const float* gamma, const float* beta,
                      const float* rel_table, const void* wqkv, const float* bqkv, const void* wqkv_t, const void* wproj_t,
                      const float* row_scale, const void* o_saved, const float* lse, void* dx, void* dqkv, void* u_out,
                      float* ln_partial, float* rpb_partial, int n_windows, int windows_per_sample, int heads, int C, float eps,
                      void* stream);
int pk_colsum_bf16(const void* g, const int32_t* rowmap, const float* row_scale, int rows_per_sample, float* partial,
                   float* out, int64_t rows, int N, void* stream);
/* exchange unit sum (hrformer.py:471-489 == hrnet.py:207-225): out = relu?(sum_i bilinear_up_i(x_i)), F.interpolate
 * (mode='bilinear', align_corners=False) fused into the sum; and the up-sampling backward in gather form.            */
int pk_fuse_sum(const void* const* inputs, const int* in_h, const int* in_w, int n_inputs, void* out, int B, int H, int W,
                int C, int relu, void* stream);
int pk_upsample_bilinear_bwd(const void* dy, void* dsrc, int B, int H, int W, int Hs, int Ws, int C, void* stream);
int pk_nchw_f32_to_nhwc_bf16(const float* x, const float* softplus_out, void* y, int B, int Cin, int H, int W, int Cpad,
                             void* stream);   /* softplus_out != NULL: also multiply by d softplus/dz = 1-exp(-y) */
/* fp32 master weights -> bf16 compute copies, table driven, one launch for the whole model (see nnops.WeightCache).
 * desc_table: array of {const float* src; int64 dst_off; int N,C,T,mode,Cp,Np; int64 dst_numel} (48 bytes each);
 * mode 0: dst[n][t][cp]=src[n][c][t] (forward), 1: dst[c][T-1-t][np]=src[n][c][t] (conv data-grad), 2: dst[c][np]=src[n][c] */
/* Deferred parameter-gradient reductions: ONE launch for any number of slab sums  out[index(i)] = sum_{s<S} part[s*slab_stride + i], i < K.
 * desc_table rows (56 bytes): { const float* part; float* out; int64 slab_stride; int S, K, layout, N, T, Cin, out_stride, pad; };
 * layout 0: index(i) = i*out_stride; layout 1: i = (n*T + t)*Cin + c -> OIHW (n*Cin + c)*T + t; layout 2: i = a*Cin + b -> a*T + b*out_stride.  One 256-thread block per
 * pk_reduce_many_cols() outputs (block_desc = row, block_first = first block of that row).  Producers: pk_wgrad_bf16 (dw NULL), pk_layernorm_bwd
 * (dgamma/dbeta NULL: partial is [blocks][2C]), pk_window_attn_bwd (dtable NULL: partial is [groups][169]).                   */
int pk_reduce_many_cols(void);
int pk_reduce_many(const void* desc_table, const int* block_desc, const int* block_first, int n_blocks, void* stream);
int pk_pack_weights(void* flat_dst_bf16, const void* desc_table, const int32_t* block_desc, const int32_t* block_first,
                    int n_blocks, void* stream);
/* Padded twins (models whose channel counts are not multiples of 8, e.g. HRFormer-base C=78, head_dim 39 -- hrformer.py:779-825):
 * every real fp32 tensor is a box r[4] at the origin of the twin's box p[4] inside one flat twin buffer.
 * desc_table rows (56 bytes): { float* real; int64 twin_off; int r[4]; int p[4]; int64 numel; }; one 256-thread block per
 * 1024 real elements (block_desc = row index, block_first = first block of that row).
 * direction 0: twin <- real (parameters, buffers); 1: real <- twin (gradients into a sink, BatchNorm running statistics);
 * 2: real += twin (accumulate into .grad).                                                                              */
int pk_embed_boxes(float* twin_base, const void* desc_table, const int* block_desc, const int* block_first, int n_blocks,
                   int direction, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POSEKERNELS_H */
