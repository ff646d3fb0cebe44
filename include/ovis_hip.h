/*
 * ovis_hip.h -- C ABI of libovis_hip.so, the MI355X (gfx950) native-op library behind
 * the `maskrcnn_benchmark._C` operator surface of hbdat/cvpr22_cross_modal_pseudo_labeling.
 *
 * Conventions (every entry point):
 *   - plain C linkage, raw DEVICE pointers + explicit sizes, no torch / ATen types;
 *   - tensors are dense, contiguous, row-major (NCHW for feature maps);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it, nothing here synchronises the host unless stated;
 *   - return value: OVIS_OK (0) on success, a negative OVIS_E* code for an argument
 *     error detected on the host, or a positive hipError_t if a HIP call failed.
 *     Nothing throws across this boundary.
 *
 * Each declaration cites the reference interface (path:line under the reference repo,
 * `mb/` = maskrcnn_benchmark/) whose behaviour it reproduces.
 */
#ifndef OVIS_HIP_H_
#define OVIS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVIS_OK 0
#define OVIS_EINVAL (-1)    /* bad size / null pointer                     */
#define OVIS_ENOSPC (-2)    /* caller workspace too small                  */
#define OVIS_ERANGE (-3)    /* problem exceeds what the kernel supports    */

/* Library / build identification.  Returns a static NUL-terminated string
 * "ovis_hip <version> gfx950 flags: <the compiler flags of every object>".  The kernel sources have no compile-time
 * switch and a product build carries no -D; experiment variants live in their own libraries
 * (tools/experiments/build_variants.sh). */
const char* ovis_version(void);

/* ------------------------------------------------------------------------------------
 * RoIAlign                                   mb/csrc/ROIAlign.h:11-46
 *   forward : mb/csrc/cuda/ROIAlign_cuda.cu:65-122,257-299 (= cpu/ROIAlign_cpu.cpp:114-219)
 *   backward: mb/csrc/cuda/ROIAlign_cuda.cu:178-254,302-346
 * input  [batch, channels, height, width] f32
 * rois   [num_rois, 5] f32 = (batch_index, x1, y1, x2, y2) in image pixels
 * output [num_rois, channels, pooled_h, pooled_w] f32
 * sampling_ratio <= 0 selects the adaptive grid ceil(roi_size / pooled_size).
 * The forward follows the reference's operation order with FP contraction disabled, so
 * it is bit-identical to the reference CPU kernel on finite inputs.
 * ---------------------------------------------------------------------------------- */
int ovis_roi_align_forward_f32(const float* input, const float* rois, float* output,
                               int num_rois, int batch, int channels, int height,
                               int width, int pooled_h, int pooled_w,
                               float spatial_scale, int sampling_ratio, void* stream);

/* Workspace form of the forward (the fast path).  When
 * ovis_roi_align_forward_mfma_supported(height, width, pooled_h, pooled_w) is non-zero (map
 * at least 16 x 16, pooled sizes at most 16 x 16) the pooled tile of every (RoI, channel)
 * is computed as two 16 x 16 x 16 products on the bf16 matrix pipe with the operands split
 * into bf16 hi + lo parts: same values as the reference up to ~1e-5 relative to sum |w x|
 * (NOT bit-identical; ovis_roi_align_forward_f32 stays the bit-exact kernel).  `workspace`:
 * at least ovis_roi_align_forward_workspace_bytes(num_rois, height, width) bytes of device
 * scratch, 256-byte aligned (per-RoI weight tables, rebuilt by every call).  Unsupported
 * shapes run the exact kernel and ignore the workspace.
 * Replaces ROIAlign_forward_cuda, mb/csrc/cuda/ROIAlign_cuda.cu:257-299. */
size_t ovis_roi_align_forward_workspace_bytes(int num_rois, int height, int width);
int ovis_roi_align_forward_mfma_supported(int height, int width, int pooled_h, int pooled_w);
int ovis_roi_align_forward_ws_f32(const float* input, const float* rois, float* output,
                                  int num_rois, int batch, int channels, int height, int width,
                                  int pooled_h, int pooled_w, float spatial_scale,
                                  int sampling_ratio, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* Forward fused with the consumer's stride (extension): pools only bins (bin_stride*i,
 * bin_stride*j) and writes output [num_rois, ceil(pooled_h/bin_stride),
 * ceil(pooled_w/bin_stride), channels] (NHWC).  The res5 head's first 1x1 convolution has
 * stride 2 (mb/modeling/backbone/resnet.py:258-275, STRIDE_IN_1X1), so it never reads the other
 * three quarters of the 14x14 tile.  Per-bin arithmetic is ovis_roi_align_forward_f32's:
 * out[r,i,j,c] is bit-identical to that kernel's [r,c,bin_stride*i,bin_stride*j]. */
int ovis_roi_align_forward_strided_nhwc_f32(const float* input, const float* rois, float* output,
                                            int num_rois, int batch, int channels, int height,
                                            int width, int pooled_h, int pooled_w, int bin_stride,
                                            float spatial_scale, int sampling_ratio, void* stream);

/* The same bins written in PAIR layout (see ovis_split_pair_f32 below: per 32 channels 64 B bf16 hi | 64 B bf16 lo;
 * the values are the exact split of the fp32 results above), [num_rois * oh * ow, channels] pair rows: the operand of
 * the res5 head's first split GEMM without an fp32 copy and a split pass.  channels % 32 == 0. */
int ovis_roi_align_forward_strided_pair_f32(const float* input, const float* rois, void* output_pair,
                                            int num_rois, int batch, int channels, int height, int width,
                                            int pooled_h, int pooled_w, int bin_stride,
                                            float spatial_scale, int sampling_ratio, void* stream);
/* The same two poolers for a feature map that is already NHWC ([batch, height, width, channels] contiguous: what the
 * trunk of this library computes in): no window staging, every bilinear tap is one contiguous channel vector; samples
 * are visited in the reference's order, so the bins are bit-identical to the forms above.  pair_out != 0: pair rows
 * (channels % 32 == 0), else NHWC fp32 (channels % 4 == 0). */
int ovis_roi_align_forward_strided_from_nhwc_f32(const float* input_nhwc, const float* rois, void* output, int num_rois,
                                                 int batch, int channels, int height, int width, int pooled_h,
                                                 int pooled_w, int bin_stride, float spatial_scale, int sampling_ratio,
                                                 int pair_out, void* stream);

/* grad_input [batch, channels, height, width] is fully overwritten (zero-filled, then
 * accumulated into) by this call; the caller does not need to clear it. */
int ovis_roi_align_backward_f32(const float* grad_output, const float* rois,
                                float* grad_input, int num_rois, int batch,
                                int channels, int height, int width, int pooled_h,
                                int pooled_w, float spatial_scale, int sampling_ratio,
                                void* stream);

/* Workspace form of the backward (the fast path).  When
 * ovis_roi_align_backward_plane_supported(height, width, pooled_h, pooled_w) is non-zero
 * (pooled sizes <= 16 and an H x W f32 plane of at most 40 KB, e.g. the 50 x 84 C4 map) the
 * gradient is computed by the plane-owner matrix-core kernel: one wave owns one
 * (image, channel) plane in LDS, no atomics, no zero-fill pass, bit-reproducible run to
 * run, HBM traffic == grad_output once + grad_input once.  `workspace` is device scratch of
 * at least ovis_roi_align_backward_workspace_bytes(num_rois, batch, height, width) bytes,
 * 256-byte aligned (per-RoI separable weight tables + per-image RoI lists, rebuilt by every
 * call).  Unsupported shapes take the same path as ovis_roi_align_backward_f32 and ignore
 * the workspace.  Replaces ROIAlign_backward_cuda, mb/csrc/cuda/ROIAlign_cuda.cu:302-346. */
size_t ovis_roi_align_backward_workspace_bytes(int num_rois, int batch, int height, int width);
int ovis_roi_align_backward_plane_supported(int height, int width, int pooled_h, int pooled_w);
int ovis_roi_align_backward_ws_f32(const float* grad_output, const float* rois,
                                   float* grad_input, int num_rois, int batch, int channels,
                                   int height, int width, int pooled_h, int pooled_w,
                                   float spatial_scale, int sampling_ratio, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* Backward of ovis_roi_align_forward_strided_nhwc_f32 (the pooler fused with the stride of the res5 head's first
 * convolution): grad_output [num_rois, channels, ceil(pooled_h / bin_stride), ceil(pooled_w / bin_stride)] holds the
 * gradient of the bins (bin_stride * i, bin_stride * j) only -- the other bins were never computed, their gradient is
 * zero -- so a quarter of the bytes of the full tile are read at bin_stride 2.  Same workspace as the form above.
 * Plane-owner kernel only: OVIS_ERANGE for shapes it does not cover (scatter into a full tile and use the form above). */
int ovis_roi_align_backward_strided_ws_f32(const float* grad_output, const float* rois, float* grad_input,
                                           int num_rois, int batch, int channels, int height, int width,
                                           int pooled_h, int pooled_w, int bin_stride, float spatial_scale,
                                           int sampling_ratio, void* workspace, size_t workspace_bytes, void* stream);

/* The same backward from the gradient as the producing data-gradient GEMM leaves it: NHWC
 * [num_rois, th, tw, channels] fp32 with th, tw = ceil(pooled / bin_stride) <= 8 (the 7 x 7 tiles of the res5 head;
 * mb/csrc/cuda/ROIAlign_cuda.cu:178-254 restricted to the computed bins).  The layout change the plane-owner kernel
 * needs hands every value over already split into bf16 hi | lo (one 256-byte tile per RoI and channel, kept behind the
 * plan tables in `workspace`): no separate permute copy, no hi/lo arithmetic on the gradient operand per item and
 * channel, two matrix instructions per stage.  OVIS_ERANGE for shapes it does not cover. */
size_t ovis_roi_align_backward_strided_nhwc_workspace_bytes(int num_rois, int batch, int channels, int height, int width);
int ovis_roi_align_backward_strided_nhwc_ws_f32(const float* grad_output_nhwc, const float* rois, float* grad_input,
                                                int num_rois, int batch, int channels, int height, int width,
                                                int pooled_h, int pooled_w, int bin_stride, float spatial_scale,
                                                int sampling_ratio, void* workspace, size_t workspace_bytes,
                                                void* stream);

/* ------------------------------------------------------------------------------------
 * NMS                                        mb/csrc/nms.h:10-28, mb/csrc/cuda/nms.cu:13-131
 * boxes [K,4] f32 xyxy (+1 pixel area convention), scores [K] f32.
 * Greedy suppression in descending-score order (ties: lower index first) with the
 * CUDA comparison `IoU > threshold` (nms.cu:60); set `ge_mode` != 0 for the CPU
 * kernel's `>=` (cpu/nms_cpu.cpp:60).  Survivors are written to keep_out[0..n) as
 * ASCENDING original indices (int64) and n to *num_keep (device int32), entirely on the
 * device: no host round trip of the suppression mask.
 * `workspace` is caller-provided device scratch of at least
 * ovis_nms_workspace_bytes(K) bytes (256-byte aligned).
 * ---------------------------------------------------------------------------------- */
size_t ovis_nms_workspace_bytes(int num_boxes);
int ovis_nms_f32(const float* boxes, const float* scores, int num_boxes, float threshold,
                 int ge_mode, void* workspace, size_t workspace_bytes, int64_t* keep_out,
                 int32_t* num_keep, void* stream);
/* Grouped form: `groups` [K] int32; a box only suppresses boxes of its own group.  One launch replaces the
 * per-class loop of the detection post-processing (mb/modeling/roi_heads/box_head/inference.py:121-163:
 * `for j in range(1, num_classes): boxlist_nms(boxes of class j)`) -- same survivors as running ovis_nms_f32
 * on every group separately, returned as ascending indices into the K candidates. */
int ovis_nms_grouped_f32(const float* boxes, const float* scores, const int32_t* groups, int num_boxes,
                         float threshold, int ge_mode, void* workspace, size_t workspace_bytes,
                         int64_t* keep_out, int32_t* num_keep, void* stream);

/* Batched, score-sorted form for the RPN proposal pipeline (mb/modeling/rpn/inference.py:95-116,
 * mb/structures/boxlist_ops.py:9-49): boxes [num_images, K, 4] are each image's candidates in DESCENDING score order
 * (the top-k prefix: no sort happens here); `drop` [num_images, K] int32 or NULL: a negative entry marks a box the
 * small-box filter removed in front of the NMS -- it suppresses nothing and never survives.  Per image: survivors as
 * ascending indices (= ascending rank) in keep_out[image, 0..n), zero-filled behind them; num_keep[image] = {n, number
 * of survivors with index < `below`} (those form a prefix of keep_out: the NMS result of the first `below` candidates
 * alone, i.e. of a selector with a smaller pre-NMS top-n).  One mask launch + one reduce launch for the whole batch. */
size_t ovis_nms_presorted_workspace_bytes(int num_images, int num_boxes);
int ovis_nms_presorted_batched_f32(const float* boxes, const int32_t* drop, int num_images, int num_boxes,
                                   float threshold, int ge_mode, int below, void* workspace, size_t workspace_bytes,
                                   int64_t* keep_out, int32_t* num_keep, void* stream);

/* ------------------------------------------------------------------------------------
 * Sorted top-k of the RPN scores             mb/modeling/rpn/inference.py:95
 *   (`objectness.topk(pre_nms_top_n, dim=1, sorted=True)` on the sigmoid scores of the batch)
 * scores [num_rows, row_len] (row stride row_stride, in floats) -> out_scores / out_idx [num_rows, k]: the k largest
 * entries of every row in DESCENDING order and their positions in the row; equal scores come in ascending position
 * (torch.topk leaves the order of ties open).  One key-building launch, one device radix sort of the whole batch
 * (rocPRIM), one emitting launch.  num_rows <= 65535, num_rows * row_len < 2^31.
 * ---------------------------------------------------------------------------------- */
size_t ovis_topk_sorted_workspace_bytes(int num_rows, int row_len);
int ovis_topk_sorted_f32(const float* scores, long row_stride, int num_rows, int row_len, int k, float* out_scores,
                         int64_t* out_idx, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * RPN proposal decode                        mb/modeling/rpn/inference.py:95-114,
 *   mb/modeling/box_coder.py:49-95, mb/structures/bounding_box.py:214-225, mb/structures/boxlist_ops.py:34-49
 * For every image and every candidate k < num_candidates: anchor index topk_idx[image, k] = (y * feature_width + x) *
 * anchors_per_position + a; its regression deltas are box_regression[image * stride_image + (y * feature_width + x) *
 * stride_position + (4 * a + c) * stride_channel] (NCHW [N, 4A, H, W]: strides (4A*H*W, 1, H*W); NHWC rows of a GEMM:
 * (H*W*ld, ld, 1)); its anchor is cell_anchors[a] + (x, y, x, y) * anchor_stride.  Writes the decoded box, clipped to
 * the image (image_wh [num_images, 2] = width, height), to boxes[image, k] and drop[image, k] = 0, or -1 when the
 * clipped box is narrower / lower than min_size.  One launch for the batch.
 * ---------------------------------------------------------------------------------- */
int ovis_rpn_decode_f32(const float* box_regression, long reg_stride_image, long reg_stride_position,
                        long reg_stride_channel, const int64_t* topk_idx, const float* cell_anchors,
                        const float* image_wh, int num_images, int num_candidates, int anchors_per_position,
                        int feature_width, float anchor_stride, float weight_x, float weight_y, float weight_w,
                        float weight_h, float xform_clip, float min_size, float* boxes, int32_t* drop, void* stream);

/* ------------------------------------------------------------------------------------
 * Box decode of the box post-processor       mb/modeling/roi_heads/box_head/inference.py:40-88,
 *   mb/modeling/box_coder.py:49-95, mb/structures/bounding_box.py:214-225
 * rel_codes [num_rows, 4 * boxes_per_row] (row stride codes_row_stride, in floats), boxes [num_rows, 4] (row stride
 * boxes_row_stride) -> decoded [num_rows, 4 * boxes_per_row] dense: BoxCoder.decode with the given weights and
 * bbox_xform_clip, expression by expression as the reference writes it.  num_images > 0 also clips every row to its image
 * (clip_to_image): rows_per_image [num_images] int32 and image_wh [num_images, 2] float (width, height) are HOST arrays
 * (at most OVIS_BOX_DECODE_MAX_IMAGES images: they travel as kernel arguments), rows are image-major and
 * sum(rows_per_image) == num_rows.  One launch.
 * ---------------------------------------------------------------------------------- */
#define OVIS_BOX_DECODE_MAX_IMAGES 16
int ovis_box_decode_f32(const float* rel_codes, long codes_row_stride, const float* boxes, long boxes_row_stride,
                        long num_rows, int boxes_per_row, float weight_x, float weight_y, float weight_w,
                        float weight_h, float xform_clip, int num_images, const int32_t* rows_per_image,
                        const float* image_wh, float* decoded, void* stream);

/* Pooler.convert_to_roi_format (mb/modeling/poolers.py:73-86): boxes[i] (HOST array of device pointers) holds image i's
 * [boxes_per_image[i], 4] boxes (boxes_per_image: HOST array); rois [sum, 5] receives (id_i, x1, y1, x2, y2) rows in list
 * order, id_i = image_ids[i] (HOST array) or i when image_ids is NULL.  One launch per OVIS_BOX_DECODE_MAX_IMAGES images. */
int ovis_rois_from_boxes_f32(const float* const* boxes, const int32_t* boxes_per_image, const int32_t* image_ids,
                             int num_images, float* rois, void* stream);

/* ------------------------------------------------------------------------------------
 * Box-regression loss of the box head, forward + backward fused
 *   mb/modeling/roi_heads/box_head/loss.py:147-170, mb/layers/smooth_l1_loss.py:6-16
 * loss[0] = sum over the positives p = positives[0..num_positives) and c < 4 of
 *   smooth_l1(box_regression[p, col0(p) + c] - regression_targets[p, c]; beta) / denominator,
 * col0(p) = 4 * labels[p] (labels != NULL: class-specific regression) or column0 (class-agnostic: 4).
 * grad_regression (may be NULL) [num_rows, num_columns] dense is fully written (zero outside the picked entries).
 * Single workgroup, fixed summation order: deterministic.
 * ---------------------------------------------------------------------------------- */
int ovis_smooth_l1_picked_fwd_bwd_f32(const float* box_regression, long regression_row_stride, int num_rows,
                                      int num_columns, const float* regression_targets, long targets_row_stride,
                                      const int64_t* positives, const int64_t* labels, int num_positives, int column0,
                                      float beta, float denominator, float* loss, float* grad_regression, void* stream);

/* ------------------------------------------------------------------------------------
 * ROIPool                                    mb/csrc/ROIPool.h:11-48
 *   kernels: mb/csrc/cuda/ROIPool_cuda.cu:17-77 (fwd), :80-108 (bwd)
 * input [batch, channels, height, width] f32, rois [num_rois, 5] (batch index, x1, y1, x2, y2);
 * output / argmax [num_rois, channels, pooled_h, pooled_w] (f32 max per bin / int32 index into the
 * H*W plane, -1 for an empty bin whose value is 0).  Backward fully writes grad_input
 * [batch, channels, height, width] (zero fill + scatter-add to the argmax positions).
 * ---------------------------------------------------------------------------------- */
int ovis_roi_pool_forward_f32(const float* input, const float* rois, float* output, int32_t* argmax,
                              int num_rois, int batch, int channels, int height, int width,
                              int pooled_h, int pooled_w, float spatial_scale, void* stream);
int ovis_roi_pool_backward_f32(const float* grad_output, const int32_t* argmax, const float* rois,
                               float* grad_input, int num_rois, int batch, int channels, int height,
                               int width, int pooled_h, int pooled_w, void* stream);

/* ------------------------------------------------------------------------------------
 * Deformable position-sensitive RoI pooling  mb/csrc/deform_pool.h:11-70
 *   kernels: mb/csrc/cuda/deform_pool_kernel_cuda.cu:31-139 (fwd), :141-263 (bwd); host: deform_pool_cuda.cu:38-90
 * data [batch, channels, height, width] f32; rois [num_rois, 5] (batch index, x1, y1, x2, y2);
 * trans [num_rois, channels_trans, part_size, part_size] (channels_trans = 2 * classes; ignored when no_trans);
 * out / out_count [num_rois, output_dim, pooled_size, pooled_size]: the mean of the bilinear samples of every bin that
 * fall inside the map, read from plane (ctop * group_size + gh) * group_size + gw, and their number (f32, as upstream).
 * Backward ACCUMULATES into grad_data (and grad_trans unless no_trans): the caller zero-fills them, as the reference's
 * autograd function does (layers/dcn/deform_pool_func.py:68-70).
 * ---------------------------------------------------------------------------------- */
int ovis_deform_psroi_pool_forward_f32(const float* data, const float* rois, const float* trans, float* out,
                                       float* out_count, int num_rois, int batch, int channels, int height, int width,
                                       int channels_trans, int no_trans, float spatial_scale, int output_dim,
                                       int group_size, int pooled_size, int part_size, int sample_per_part,
                                       float trans_std, void* stream);
int ovis_deform_psroi_pool_backward_f32(const float* grad_out, const float* out_count, const float* data,
                                        const float* rois, const float* trans, float* grad_data, float* grad_trans,
                                        int num_rois, int batch, int channels, int height, int width,
                                        int channels_trans, int no_trans, float spatial_scale, int output_dim,
                                        int group_size, int pooled_size, int part_size, int sample_per_part,
                                        float trans_std, void* stream);

/* ------------------------------------------------------------------------------------
 * Sigmoid focal loss                         mb/csrc/SigmoidFocalLoss.h:10-41
 *   kernels: mb/csrc/cuda/SigmoidFocalLoss_cuda.cu:21-58 (fwd), :62-101 (bwd)
 * logits [num, num_classes] f32; targets [num] int32 (-1 ignore, 0 background,
 * j+1 = positive for column j); losses / d_logits [num, num_classes] f32.
 * ---------------------------------------------------------------------------------- */
int ovis_sigmoid_focal_loss_forward_f32(const float* logits, const int32_t* targets,
                                        float* losses, int num, int num_classes,
                                        float gamma, float alpha, void* stream);
int ovis_sigmoid_focal_loss_backward_f32(const float* logits, const int32_t* targets,
                                         const float* d_losses, float* d_logits, int num,
                                         int num_classes, float gamma, float alpha,
                                         void* stream);

/* ------------------------------------------------------------------------------------
 * bf16 hi/lo operand split for fp32-accurate GEMMs on the bf16 matrix pipe ("next" row f-1:
 * the res5 1x1 convolutions, mb/modeling/backbone/resnet.py:239-344, as NHWC GEMMs).
 * src [rows, cols] f32 (row stride in elements) -> dst [rows, 3*cols] bf16 with row layout
 * [hi | hi | lo] (mode 0, left operand) or [hi | lo | hi] (mode 1, right operand), hi =
 * bf16(x), lo = bf16(x - hi): a plain bf16 GEMM over the 3*cols columns of a mode-0 and a
 * mode-1 operand is the three-term product hi.hi + hi.lo + lo.hi (~4e-6 relative error).
 * cols and the row stride must be multiples of 4, src 16-byte and dst 8-byte aligned
 * (else OVIS_ERANGE: use the fp32 GEMM).
 * ---------------------------------------------------------------------------------- */
int ovis_split_bf16x3_f32(const float* src, long src_row_stride, void* dst_bf16, long rows,
                          int cols, int mode, void* stream);

/* Split + im2col of an NHWC tensor for odd-sized stride-1 "same" convolutions as bf16 GEMMs:
 * src [num, height, width, channels] f32 -> dst [num*height*width, 3*T*channels] bf16 with
 * T = kh*kw taps and row layout [hi(tap 0..T-1) | hi(tap 0..T-1) | lo(tap 0..T-1)]; tap
 * (ky, kx) reads pixel (y + ky - kh/2, x + kx - kw/2), zeros outside the map; flip != 0
 * takes the taps in reverse order (180-degree rotated kernel of the data gradient).
 * A bf16 GEMM of dst against the mode-1 split of the [Cout, T*channels] weight matrix is the
 * convolution (res5 conv2, mb/modeling/backbone/resnet.py:288-300).  channels % 4 == 0. */
int ovis_im2col_split_bf16x3_f32(const float* src, void* dst_bf16, long num, int height,
                                 int width, int channels, int kh, int kw, int flip, void* stream);

/* Fused GEMM epilogue of the NHWC res5 head: y[rows, cols] = act(y + bias[col] (+ residual)) in
 * place, act = ReLU when `relu` != 0 (Bottleneck.forward, mb/modeling/backbone/resnet.py:323-344:
 * FrozenBN shift, shortcut add and ReLU in one pass).  bias / residual may be NULL; cols % 4 == 0
 * and 16-byte aligned pointers (else OVIS_ERANGE). */
int ovis_bias_act_f32(float* y, const float* bias, const float* residual, long rows, int cols,
                      int relu, void* stream);

/* ------------------------------------------------------------------------------------
 * Pair-layout split GEMM / implicit-GEMM convolution (csrc/split_gemm.hip): the res5 head and the
 * frozen trunk (Bottleneck.forward, mb/modeling/backbone/resnet.py:323-344; ResNetHead :155-204)
 * as fp32-accurate products on the bf16 matrix cores, with the operand split shared by the three
 * hi/lo products inside the kernel.
 *
 * PAIR LAYOUT of a row of K fp32 values (K % 32 == 0): K/32 blocks of 128 bytes, each
 * [hi(32 x bf16) | lo(32 x bf16)], hi = bf16(x), lo = bf16(x - hi); 4*K bytes per row.
 * ---------------------------------------------------------------------------------- */
/* src [rows, cols] f32 (row stride in elements) -> dst pair rows (4*cols bytes each, dense).
 * cols % 32 == 0, stride % 4 == 0, 16-byte aligned pointers (else OVIS_ERANGE). */
int ovis_split_pair_f32(const float* src, long src_row_stride, void* dst_pair, long rows,
                        int cols, void* stream);

/* Backward ReLU gate fused with the operand split: g = dy * (y > 0) -> dst_pair (pair rows, dense) and, when
 * g_f32 != NULL, also as dense fp32.  dy [rows, cols] f32 (row stride in elements); gate = the saved forward
 * output y, dense: its pair form (gate_is_pair != 0; only the hi halves are read) or fp32, or NULL (no gate:
 * a plain split).  threshold_backward of mb/modeling/backbone/resnet.py:323-344's relu_ calls.  cols % 32 == 0.
 * g_pooled != NULL: [rows / pool_rows, cols] f32, the gradient of the mean over every pool_rows consecutive rows
 * (the 7x7 average pooling of FastRCNNPredictor.forward, roi_box_predictors.py:62-66, behind the res5 head) is
 * added on the fly, g = (dy + g_pooled[row / pool_rows] / pool_rows) * (y > 0); dy may then be NULL. */
int ovis_gate_split_pair_f32(const float* dy, long dy_row_stride, const void* gate, int gate_is_pair,
                             void* dst_pair, float* g_f32, long rows, int cols, const float* g_pooled,
                             int pool_rows, void* stream);
/* The same with the gradient of a ROW-GROUP GATHER added on the fly: the rows come in groups of pool_rows (the 7x7
 * positions of one RoI); group_slot [rows / pool_rows] int32 holds, for a group that a consumer gathered (the mask
 * head reads the res5 features of the positive RoIs only: mb/modeling/roi_heads/mask_head/mask_head.py:13-41,62-66),
 * its index in g_selected [num_selected * pool_rows, cols] f32 (dense), -1 otherwise:
 * g = (dy + g_pooled[group] / pool_rows + g_selected[group_slot[group] * pool_rows + row in group]) * (y > 0).
 * Replaces the zero-filled [rows, cols] tensor an index backward would build and this kernel would read back.
 * dy, g_pooled may be NULL; g_selected and group_slot are given together. */
int ovis_gate_split_pair_rows_f32(const float* dy, long dy_row_stride, const void* gate, int gate_is_pair,
                                  void* dst_pair, float* g_f32, long rows, int cols, const float* g_pooled,
                                  int pool_rows, const float* g_selected, const int32_t* group_slot, void* stream);

/* Weight preparation of a (trainable) convolution in one launch: weight [out_channels, in_channels, taps] f32 times the
 * folded FrozenBN scale [out_channels] (may be NULL; mb/layers/batch_norm.py:19-31) -> fwd_pair [out_channels, taps*in]
 * pair rows, k = tap*in + c (the forward / weight-gradient layout) and, when bwd_pair != NULL, [in_channels, taps*out]
 * pair rows, k' = tap*out + n (the data-gradient operand).  in_channels % 32 == 0 (out_channels too for bwd_pair). */
int ovis_weight_prep_pair_f32(const float* weight, const float* scale, void* fwd_pair, void* bwd_pair,
                              int out_channels, int in_channels, int taps, void* stream);

/* dweight [out, in, taps] = scale[out] * sum over the slabs of ovis_split_gemm_pair_tn ([slices, out, taps*in]). */
int ovis_slab_reduce_f32(const float* slabs, const float* scale, float* dweight, int slices, int out_channels,
                         int in_channels, int taps, void* stream);

/* Pair-layout im2col (the M-contracting weight gradient of a 3x3 needs the rows materialised):
 * src NHWC [num, height, width, channels] pair rows -> dst [num*height*width, kh*kw*channels]
 * pair rows, tap-major, zero rows for taps outside the map.  channels % 32 == 0. */
int ovis_im2col_pair(const void* src_pair, void* dst_pair, long num, int height, int width,
                     int channels, int kh, int kw, void* stream);

/* ovis_split_gemm_pair for a data gradient whose result passes a ReLU gate: C = (A B^T) * (y > 0) with y the forward
 * activation in pair layout (gate_pair, rows gate_row_bytes apart; only the hi halves are read), written as fp32 (c)
 * and / or in pair layout (c_pair) -- the gate and the operand split of the NEXT backward GEMM in this epilogue.
 * n % 32 == 0. */
int ovis_split_gemm_pair_gated(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                               float* c, long ldc, void* c_pair, long c_pair_row_bytes, const void* gate_pair,
                               long gate_row_bytes, long m, int n, int channels, int taps_h, int taps_w,
                               int height, int width, int flip, int config, void* stream);

/* ovis_split_gemm_pair_gated with a split-K workspace (size: ovis_split_gemm_pair_workspace_bytes(m, n, channels, 0,
 * taps_h, taps_w, width); NULL / too small = the un-split grid): grids that leave most CUs idle while every workgroup
 * walks a long K (the data gradients of a trainable trunk on 50 x 84 maps) are cut into K slices as in
 * ovis_split_gemm_pair; the slab reduction applies the gate and writes c / c_pair.  Same results as the un-split call up
 * to the fp32 summation order of the slices (fixed: reproducible). */
int ovis_split_gemm_pair_gated_ws(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                  float* c, long ldc, void* c_pair, long c_pair_row_bytes, const void* gate_pair,
                                  long gate_row_bytes, long m, int n, int channels, int taps_h, int taps_w,
                                  int height, int width, int flip, void* workspace, size_t workspace_bytes,
                                  int config, void* stream);

/* The input gradient of an identity bottleneck (mb/modeling/backbone/resnet.py:290-342: out = relu(conv3(..) + x)), handed
 * to the block below ready for use: C = (A B^T + r) * (x > 0) -- r the gradient arriving through the shortcut, given in
 * pair layout (residual_pair: hi + lo), x the block input in pair layout (gate_pair: the block below's ReLU output, only
 * the hi halves are read) -- written in pair layout (c_pair) and / or fp32 (c).  The block below then needs neither a
 * gate + split pass nor an fp32 gradient tensor.  Plain products only (1x1); n % 32 == 0, n >= 128 (else OVIS_ERANGE). */
int ovis_split_gemm_pair_rp_gated(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes, float* c,
                                  long ldc, void* c_pair, long c_pair_row_bytes, const void* residual_pair,
                                  long residual_pair_row_bytes, const void* gate_pair, long gate_row_bytes, long m, int n,
                                  int channels, int config, void* stream);

/* im2col of a strided convolution on an NCHW f32 image [num, channels, height, width] into pair rows
 * [num*ho*wo, k_padded]: k = (ky*kw + kx)*channels + c, columns >= kh*kw*channels are zero (k_padded % 32 == 0).
 * The 7x7 stride-2 stem (mb/modeling/backbone/resnet.py:347-366) then is one ovis_split_gemm_pair. */
int ovis_im2col_nchw_pair_f32(const float* src, void* dst_pair, int num, int channels, int height, int width,
                              int kh, int kw, int stride, int pad, int k_padded, void* stream);

/* C[m, n] = act( sum_{tap, c} A[row(m, tap), c] * B[n, tap*channels + c] (+ sum_c A2[m, c] * B[n, channels + c])
 *                + bias[n] + residual[m, n] )
 *   a_pair : pair rows of `channels` values, a_row_bytes apart.  taps_h = taps_w = 1: a plain
 *            [m, channels] matrix.  Otherwise an NHWC tensor [m / (height*width), height, width,
 *            channels]; tap (ty, tx) of row (r, y, x) reads pixel (y + ty - taps_h/2, x + tx -
 *            taps_w/2) (negated offsets when flip != 0: the data-gradient convolution), zeros
 *            outside the map -- stride-1 "same" convolution with no im2col matrix.  Maps whose largest
 *            tap shift (taps_h/2)*width + taps_w/2 is <= 8 rows (the 7x7 maps of the res5 head) are
 *            staged once per channel block with a halo and read by all taps from LDS.
 *   a2_pair: optional second operand (taps 1x1 only; NULL / channels2 = 0 otherwise): [m, channels2] pair
 *            rows whose products follow a_pair's along K, b_pair then holds channels + channels2 values
 *            per row -- conv3 + projection shortcut of a bottleneck (mb/modeling/backbone/resnet.py:
 *            323-344) as ONE product, the shortcut tensor is never written.
 *   b_pair : n pair rows of taps*channels (+ channels2) values (weights [n, tap, c]), b_row_bytes apart.
 *   c      : fp32 result, row stride ldc elements, or NULL;  c_pair: the result in pair layout
 *            (rows c_pair_row_bytes apart; needs n % 32 == 0), or NULL -- the operand split of the
 *            NEXT layer fused into this epilogue.  bias [n] / residual [m, n] (stride ldr) may be NULL.
 *   workspace : ovis_split_gemm_pair_workspace_bytes(...) bytes (16-byte aligned) or NULL.  Problems with few
 *            output tiles and a long K (the pseudo-box passes: m = a few hundred rows) are cut along K into
 *            slices whose partial tiles go to fp32 slabs in the workspace; a second launch sums them and
 *            applies the epilogue (deterministic).  Without a workspace the un-split grid runs.
 *   config : 0 = choose.  Otherwise a bit set for tests and A/B measurements: 1 = one LDS stage, 2 = two
 *            stages, 4 = never use the halo form, 8 = no K slices.
 * channels % 32 == 0, channels2 % 32 == 0, n % 4 == 0, 16-byte aligned pointers and strides (else OVIS_ERANGE). */
size_t ovis_split_gemm_pair_workspace_bytes(long m, int n, int channels, int channels2, int taps_h, int taps_w,
                                            int width);
/* The same query for a launch with a non-zero `config` (bit 16 = co-scheduled launch: slices chosen for least total work;
 * bits 8..15 = forced slice count): the plan, and with it the slabs' size, follows config. */
size_t ovis_split_gemm_pair_workspace_bytes_ex(long m, int n, int channels, int channels2, int taps_h, int taps_w,
                                               int width, int config);
int ovis_split_gemm_pair(const void* a_pair, long a_row_bytes, const void* a2_pair, long a2_row_bytes,
                         const void* b_pair, long b_row_bytes, float* c, long ldc, void* c_pair,
                         long c_pair_row_bytes, const float* bias, const float* residual, long ldr, long m,
                         int n, int channels, int channels2, int taps_h, int taps_w, int height, int width,
                         int flip, int relu, void* workspace, size_t workspace_bytes, int config, void* stream);
/* The plain (1x1) form with the shortcut operand in PAIR layout: residual_pair [m, n] pair rows (n % 32 == 0), added as
 * hi + lo (exact in fp32; 2^-17 relative to the value the pair was split from).  A bottleneck
 * (mb/modeling/backbone/resnet.py:323-344) holds its input as the pair operand of conv1 anyway: with this form the
 * identity shortcut needs no fp32 copy of the block input, so the producing block writes its result in pair layout only. */
int ovis_split_gemm_pair_rp(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes, float* c, long ldc,
                            void* c_pair, long c_pair_row_bytes, const float* bias, const void* residual_pair,
                            long residual_pair_row_bytes, long m, int n, int channels, int relu, void* workspace,
                            size_t workspace_bytes, int config, void* stream);

/* Weight gradient on pair operands: c_slabs[s][n][tap*channels + c] = sum over the rows m of slice s of
 * G[m, n] * X[row(m, tap), c]  (G = gated output gradient [m, n], X = the layer input [m, channels], both pair
 * rows; taps as in ovis_split_gemm_pair, X read shifted with zeros outside the map).  The rows are cut into
 * `slices` (ovis_split_gemm_tn_slices gives a count that fills the chip); the caller sums the slabs.
 * n % 128 == 0, channels % 128 == 0 (else OVIS_ERANGE); any map size. */
int ovis_split_gemm_tn_slices(long m, int n, int channels, int taps);
int ovis_split_gemm_pair_tn(const void* g_pair, long g_row_bytes, const void* x_pair, long x_row_bytes,
                            float* c_slabs, int slices, long m, int n, int channels, int taps_h,
                            int taps_w, int height, int width, void* stream);

/* ------------------------------------------------------------------------------------
 * Training-target construction on the device (csrc/targets.hip), one launch per image instead of ~100 tensor ops.
 *
 * ovis_match_encode_f32: boxlist_iou (mb/structures/boxlist_ops.py:53-89) of gt_boxes [num_gt,4] x proposals
 * [num_proposals,4] -> Matcher without low-quality matches (mb/modeling/matcher.py:42-81) -> matched_idx
 * [num_proposals] (argmax gt, 0 when below/between the thresholds, i.e. matched.clamp(min=0)), labels (gt label;
 * 0 below low_threshold; between the thresholds -1 [between_keeps_label == 0, box head: box_head/loss.py:60-70] or
 * the label of gt 0 [!= 0, mask head: mask_head/loss.py:60-77]) and, when regression_targets != NULL, BoxCoder.encode
 * (mb/modeling/box_coder.py:22-52) of the matched gt box with weights (wx, wy, ww, wh).
 *
 * ovis_project_masks_f32: project_masks_on_boxes (mb/modeling/roi_heads/mask_head/loss.py:11-42) for binary masks
 * [G, height, width] (1 byte per pixel; bool or uint8): crop to the rounded box, bilinear resize to resolution^2
 * (align_corners = False), cast back to the mask dtype -> out [num, resolution, resolution] f32.
 * ---------------------------------------------------------------------------------- */
int ovis_match_encode_f32(const float* gt_boxes, const int64_t* gt_labels, const float* proposals, int num_gt,
                          int num_proposals, float high_threshold, float low_threshold, int between_keeps_label,
                          float wx, float wy, float ww, float wh, int64_t* matched_idx, int64_t* labels,
                          float* regression_targets, void* stream);
/* ovis_rpn_match_encode_f32: the RPN's target building (mb/modeling/rpn/loss.py:21-89) for one image: boxlist_iou of
 * gt_boxes [num_gt, 4] x anchors [num_anchors, 4] -> Matcher WITH low-quality matches (mb/modeling/matcher.py:42-112: an
 * anchor whose IoU equals a ground truth's best IoU keeps its argmax; pass 1 takes the per-ground-truth maxima into
 * best_per_gt_scratch [num_gt] uint32, pass 2 compares bit patterns) -> labels [num_anchors] int64: 1 matched, 0 below the
 * low threshold, -1 ignored (between the thresholds, or visibility[a] == 0: anchors straddling the image border), in the
 * reference's order of the three rules; regression_targets [num_anchors, 4] = BoxCoder.encode of the matched ground truth
 * (index clamped at 0) with weights (wx, wy, ww, wh).  allow_low_quality_matches == 0: the plain Matcher. */
int ovis_rpn_match_encode_f32(const float* gt_boxes, const float* anchors, const uint8_t* visibility, int num_gt,
                              int num_anchors, float high_threshold, float low_threshold, int allow_low_quality_matches,
                              float wx, float wy, float ww, float wh, uint32_t* best_per_gt_scratch, int64_t* labels,
                              float* regression_targets, void* stream);
int ovis_project_masks_f32(const uint8_t* masks, const int64_t* gt_index, const float* boxes, int num, int height,
                           int width, int resolution, int masks_are_bool, float* out, void* stream);
/* ovis_sample_fg_bg: BalancedPositiveNegativeSampler of one image (mb/modeling/balanced_positive_negative_sampler.py:
 * 19-68) in one launch.  labels [num] int64 (>= 1 positive, 0 negative, < 0 ignored).  num_pos = min(#positives,
 * max_positives), num_neg = min(#negatives, batch_size - num_pos); a uniformly random subset of each size (keys =
 * splitmix64(seed, index): the reference draws torch.randperm()[:k], whose stream is version dependent) -> selected
 * [batch_size] int64: the chosen indices ASCENDING (the order of the reference's nonzero(pos_mask | neg_mask)), zeros
 * behind them; positive_slots [batch_size]: positions of the positives inside `selected`; counts[0] = number selected,
 * counts[1] = positives among them.  No host round trip.
 *
 * ovis_project_pasted_masks_f32: ovis_project_masks_f32 for ground truths whose binary image mask is DEFINED by a
 * prob_resolution^2 probability map and a box -- Masker.forward_single_image (mb/modeling/roi_heads/mask_head/
 * inference.py:100-160, padding 1: pad, expand the box by (M+2)/M, truncate to integers, bilinear resize, > threshold,
 * paste clipped to the image) -- as the pseudo labels' masks are (st_generalized_rcnn.py:266-271).  Every mask pixel
 * the crop + resize reads is evaluated from the map on the fly: same targets as pasting first, no H x W canvas.
 * mask_probs [G, prob_resolution, prob_resolution] f32, gt_boxes [G, 4], gt_index [num], boxes [num, 4]. */
int ovis_sample_fg_bg(const int64_t* labels, int num, int batch_size, int max_positives, uint64_t seed,
                      int64_t* selected, int64_t* positive_slots, int32_t* counts, void* stream);
int ovis_project_pasted_masks_f32(const float* mask_probs, const float* gt_boxes, const int64_t* gt_index,
                                  const float* boxes, int num, int image_height, int image_width, int prob_resolution,
                                  int resolution, float threshold, float* out, void* stream);
/* ovis_gather_rows: out[i] = src[index[i]] for up to two [P, 4] f32 arrays and two [P] int64 arrays at once (a NULL source
 * is skipped) -- the sampled proposals' boxes, regression targets, labels and matched ground truths
 * (mb/modeling/roi_heads/box_head/loss.py:112-121 indexes every field of the BoxList separately).  One launch. */
int ovis_gather_rows(const int64_t* index, int num, const float* boxes_a, float* boxes_a_out, const float* boxes_b,
                     float* boxes_b_out, const int64_t* ints_a, int64_t* ints_a_out, const int64_t* ints_b,
                     int64_t* ints_b_out, void* stream);

/* ------------------------------------------------------------------------------------
 * Cross-modal head: fp32 GEMM on the matrix cores
 *   mb/modeling/roi_heads/box_head/roi_box_predictors.py:66-71 (emb_pred Linear, einsum('pe,ce->pc'),
 *   bbox_pred Linear) and their autograd transposes.
 * C[m,n] = sum_k A(m,k) * B(n,k) (+ bias[n]);  A(m,k) = A[m*a_row_stride + k*a_k_stride], B likewise,
 * C row-major with row stride c_row_stride.  Exact fp32 (v_mfma_f32_32x32x2_f32): differs from a
 * BLAS fp32 GEMM by summation order only.
 * ---------------------------------------------------------------------------------- */
int ovis_gemm_f32(const float* A, long a_row_stride, long a_k_stride, const float* B,
                  long b_row_stride, long b_k_stride, const float* bias, float* C,
                  long c_row_stride, int M, int N, int K, void* stream);

/* General form: C = alpha * A.B^T (+ C when accumulate != 0) + bias, bias indexed by column n or, when
 * bias_per_row != 0, by row m (convolution bias in the deformable-conv products). */
int ovis_gemm_ex_f32(const float* A, long a_row_stride, long a_k_stride, const float* B,
                     long b_row_stride, long b_k_stride, const float* bias, int bias_per_row,
                     float alpha, int accumulate, float* C, long c_row_stride, int M, int N, int K,
                     void* stream);

/* The same product with K cut into slices when the problem has too few tiles to fill the chip (the head's small
 * GEMMs: a weight gradient contracts over ~1000 rows into a 768 x 2048 result): every slice writes its own fp32 slab
 * in `workspace` and one kernel adds the slabs in slice order -- no atomics, bit-reproducible run to run.
 * ovis_gemm_f32_workspace_bytes(M, N, K) = bytes the split needs (0: the problem is not cut); workspace NULL = never
 * cut (what ovis_gemm_f32 / ovis_gemm_ex_f32 do); too small a workspace: OVIS_ENOSPC. */
size_t ovis_gemm_f32_workspace_bytes(int M, int N, int K);
int ovis_gemm_ex_ws_f32(const float* A, long a_row_stride, long a_k_stride, const float* B,
                        long b_row_stride, long b_k_stride, const float* bias, int bias_per_row,
                        float alpha, int accumulate, float* C, long c_row_stride, int M, int N, int K,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Region <-> noun alignment of the teacher (mb/modeling/detector/st_generalized_rcnn.py:243-262):
 * for every noun w: raw_scores[w] = max_p <region_emb[p], noun_emb[w]>, best_region[w] = argmax_p
 * (lowest p on exact ties), sigmoid_scores[w] = sigmoid(raw).  region_emb [num_regions, dim],
 * noun_emb [num_nouns, dim]; the [num_regions, num_nouns] score matrix is never materialised. */
int ovis_region_noun_align_f32(const float* region_emb, const float* noun_emb, float* raw_scores,
                               float* sigmoid_scores, int64_t* best_region, int num_regions,
                               int num_nouns, int dim, void* stream);

/* ------------------------------------------------------------------------------------
 * Student losses, forward + backward fused
 *   weighted CE : mb/modeling/roi_heads/box_head/loss.py:172-174
 *     loss[0] = sum_p w[label_p] * CE(logits_p, label_p) / num_rows, w[0] = bg_weight, w[c>0] = 1;
 *     dlogits (may be NULL) receives d loss / d logits.  row_scratch: num_rows floats.
 *   stochastic mask BCE : mb/modeling/roi_heads/mask_head/roi_mask_predictors.py:41-65,
 *                         mb/modeling/roi_heads/mask_head/loss.py:117-148
 *     z = mu + eps * sigma on logit channel `channel` of the positive RoIs pos_index[0..num_pos);
 *     loss[0] = mean BCEWithLogits(z, targets); dmu [num_rois, C, pixels] and dsigma [num_rois, pixels]
 *     (either may be NULL) are fully written (zero outside the positives).  sigma / eps may be NULL
 *     (plain BCE).  mu [num_rois, C, pixels], sigma [num_rois, pixels], eps [num_rois, C, pixels],
 *     targets [num_pos, pixels].  row_scratch: num_pos floats.
 *     `_classes_`: the logit channel of positive i is channels[i] (class-specific masks,
 *     CLS_AGNOSTIC_MASK False: mask_logits[positive_inds, labels_pos], mask_head/loss.py:131-141;
 *     values outside [0, C) are clamped); eps, when given, is read at the same channel.
 * ---------------------------------------------------------------------------------- */
int ovis_weighted_ce_fwd_bwd_f32(const float* logits, const int64_t* labels, float bg_weight,
                                 float* loss, float* dlogits, float* row_scratch, int num_rows,
                                 int num_classes, void* stream);
int ovis_mask_bce_stochastic_fwd_bwd_f32(const float* mu, const float* sigma, const float* eps,
                                         const int64_t* pos_index, const float* targets, float* loss,
                                         float* dmu, float* dsigma, float* row_scratch, int num_rois,
                                         int num_pos, int num_channels, int mask_pixels, int channel,
                                         void* stream);
int ovis_mask_bce_stochastic_classes_fwd_bwd_f32(const float* mu, const float* sigma, const float* eps,
                                                 const int64_t* pos_index, const int64_t* channels,
                                                 const float* targets, float* loss, float* dmu, float* dsigma,
                                                 float* row_scratch, int num_rois, int num_pos,
                                                 int num_channels, int mask_pixels, void* stream);

/* ------------------------------------------------------------------------------------
 * Deformable convolution v1 / v2 building blocks      mb/csrc/deform_conv.h:11-190
 *   kernels: mb/csrc/cuda/deform_conv_kernel_cuda.cu:198-250 (im2col), :287-342 (col2im),
 *            :381-443 (col2im_coord), modulated variants :578-774; host loops deform_conv_cuda.cu:161-694
 * columns [channels*kernel_h*kernel_w, num_images, H_out, W_out]; input [num_images, channels, H, W];
 * offset [num_images, deformable_group*2*kernel_h*kernel_w, H_out, W_out];
 * mask   [num_images, deformable_group*kernel_h*kernel_w, H_out, W_out] or NULL (v1).
 * col2im ACCUMULATES into grad_input (caller zero-fills, as the reference does); col2im_coord overwrites
 * grad_offset (and grad_mask when non-NULL).  The surrounding GEMMs are ovis_gemm_ex_f32 on this layout.
 * ---------------------------------------------------------------------------------- */
int ovis_deform_im2col_f32(const float* input, const float* offset, const float* mask, float* columns,
                           int num_images, int channels, int height, int width, int kernel_h,
                           int kernel_w, int pad_h, int pad_w, int stride_h, int stride_w,
                           int dilation_h, int dilation_w, int deformable_group, void* stream);
int ovis_deform_col2im_f32(const float* columns, const float* offset, const float* mask,
                           float* grad_input, int num_images, int channels, int height, int width,
                           int kernel_h, int kernel_w, int pad_h, int pad_w, int stride_h,
                           int stride_w, int dilation_h, int dilation_w, int deformable_group,
                           void* stream);
int ovis_deform_col2im_coord_f32(const float* columns, const float* input, const float* offset,
                                 const float* mask, float* grad_offset, float* grad_mask,
                                 int num_images, int channels, int height, int width, int kernel_h,
                                 int kernel_w, int pad_h, int pad_w, int stride_h, int stride_w,
                                 int dilation_h, int dilation_w, int deformable_group, void* stream);

/* Deformable convolution forward WITHOUT the column buffer (mb/csrc/cuda/deform_conv_cuda.cu:161-260, :491-560;
 * sampling: deform_conv_kernel_cuda.cu:198-250, 578-644): one implicit GEMM over rows = output pixels (batch, out_h,
 * out_w) and K = (tap, channel) on the pair-layout split GEMM of the res5 head -- the A tile of every k-step is bilinear-
 * sampled in the kernel (zero padding, the reference's four-term order, times the mask when given) from input_nhwc
 * [batch, height, width, channels] fp32, split into bf16 hi / lo and multiplied with weight_pair [out_channels,
 * taps * channels] (tap-major pair rows: ovis_weight_prep_pair_f32).  offset [batch, dg*2*taps, out_h, out_w], mask
 * [batch, dg*taps, out_h, out_w] or NULL (v1), bias [out_channels] or NULL; output [batch*out_h*out_w, out_channels]
 * fp32 rows (NHWC), row stride ldc.  Convolution groups = 1; (channels / deformable_group) % 32 == 0,
 * out_channels % 4 == 0 (else OVIS_ERANGE: use the column route). */
int ovis_deform_conv_implicit_f32(const float* input_nhwc, const float* offset, const float* mask,
                                  const void* weight_pair, long weight_row_bytes, const float* bias, float* output,
                                  long ldc, int batch, int channels, int height, int width, int out_channels, int out_h,
                                  int out_w, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h,
                                  int pad_w, int dil_h, int dil_w, int deformable_group, void* stream);

/* Text side of the cross-modal head: mb/modeling/language_backbone/transformers.py:27-68 (BERT.forward: the frozen
 * word-embedding table indexed by the token ids, no transformer forward) + mb/modeling/detector/st_generalized_rcnn.py:
 * 202-209 (extract_emb): out[n] = normalize( sum_l (1 - special[n,l]) * table[ids[n,l]] / sum_l (1 - special[n,l]) ),
 * normalize = x / max(||x||_2, 1e-12).  table [table_rows, dim] f32 (dim % 4 == 0), input_ids / special_tokens_mask
 * [num_words, max_tokens] int32 as the tokenizer pads them ([CLS], [SEP], [PAD] carry special = 1), out [num_words,
 * dim].  The [num_words, max_tokens, dim] tensor the reference builds is never materialised.  A token id outside the
 * table makes that word's row NaN. */
int ovis_text_embed_f32(const float* table, long table_rows, int dim, const int32_t* input_ids,
                        const int32_t* special_tokens_mask, int num_words, int max_tokens, float* out, void* stream);

/* Polygon ground truth -> mask-head targets: project_masks_on_boxes (mb/modeling/roi_heads/mask_head/loss.py:11-42)
 * for SegmentationMask(mode='poly') targets -- PolygonInstance.crop / resize / convert_to_binarymask
 * (mb/structures/segmentation_mask.py:270-334), whose rasteriser is pycocotools' rleFrPoly + merge + decode
 * (pycocotools==2.0, requirements.txt:35; restated, see csrc/polygons.hip).  coords: all polygons' (x, y) float32
 * pairs back to back; polygon q owns coords[polygon_start[q] .. polygon_start[q+1]) (offsets in floats, even);
 * ground-truth instance g owns polygons [instance_start[g], instance_start[g+1]) (polygons with < 3 vertices are
 * skipped, as PolygonInstance.__init__ drops them); gt_index [num] int64 picks the instance of every box; boxes
 * [num, 4] xyxy in image pixels; out [num, resolution, resolution] f32 in {0, 1}.  resolution <= 64. */
int ovis_project_polygon_masks_f32(const float* coords, const int32_t* polygon_start, const int32_t* instance_start,
                                   const int64_t* gt_index, const float* boxes, int num, int image_width,
                                   int image_height, int resolution, float* out, void* stream);

/* Deformable convolution BACKWARD on NHWC rows (csrc/deform_conv_rows.hip; mb/csrc/cuda/deform_conv_cuda.cu:271-497,
 * 580-694): rows m = (image, h_out, w_out), k = (tap, channel), the layout of ovis_deform_conv_implicit_f32.
 * ovis_deform_im2col_pair_rows_f32: the sampled (x mask) rows col[m, (t, c)] written ONCE, in pair layout (row stride
 *   columns_row_bytes >= 4 * taps * channels) -- the X operand of ovis_split_gemm_pair_tn for the weight gradient.
 * ovis_deform_col2im_rows_f32: one pass over dcol_rows [rows, dcol_ld >= taps * channels] fp32 (= dY . W, an
 *   ovis_split_gemm_pair product): grad_input_nhwc [batch, height, width, channels] += scatter (fp32 atomics; the caller
 *   zero-fills), grad_offset [batch, dg*2*taps, out_h, out_w] and grad_mask [batch, dg*taps, out_h, out_w] += the
 *   coordinate / mask gradients (atomics: the caller zero-fills; grad_mask / mask NULL = v1).  Either output group may be
 *   NULL.  (channels / deformable_group) % 32 == 0, else OVIS_ERANGE. */
int ovis_deform_im2col_pair_rows_f32(const float* input_nhwc, const float* offset, const float* mask, void* columns_pair,
                                     long columns_row_bytes, int batch, int channels, int height, int width, int out_h,
                                     int out_w, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                                     int dil_h, int dil_w, int deformable_group, void* stream);
int ovis_deform_col2im_rows_f32(const float* dcol_rows, long dcol_ld, const float* input_nhwc, const float* offset,
                                const float* mask, float* grad_input_nhwc, float* grad_offset, float* grad_mask, int batch,
                                int channels, int height, int width, int out_h, int out_w, int kernel_h, int kernel_w,
                                int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int deformable_group,
                                void* stream);

/* Last bottleneck of a res5 chain with the head's average pooling in the GEMM epilogue (mb/modeling/roi_heads/box_head/
 * roi_box_predictors.py:62-66: AdaptiveAvgPool2d over the 7 x 7 map of every RoI): act(A.B^T + bias + shortcut) as
 * ovis_split_gemm_pair_rp, and pooled[m / pool_rows][n] += pool_scale * result summed over the pool_rows rows of every map
 * (32 <= pool_rows <= 64; `pooled` [ceil(m / pool_rows), n] fp32 must be ZERO-FILLED by the caller; every pooled value
 * receives at most two fp32 atomic addends: bit-reproducible).  c and c_pair may BOTH be NULL: then nothing but the pooled
 * rows is written (the no-grad teacher pass).  residual_pair may be NULL.  Plain un-split 128-column launches only:
 * ovis_split_gemm_pair_pool_supported(m, n, channels, pool_rows) != 0, else OVIS_ERANGE. */
int ovis_split_gemm_pair_pool_supported(long m, int n, int channels, int pool_rows);
int ovis_split_gemm_pair_rp_pool(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes, float* c,
                                 long ldc, void* c_pair, long c_pair_row_bytes, const float* bias,
                                 const void* residual_pair, long residual_pair_row_bytes, long m, int n, int channels,
                                 int relu, float* pooled, int pool_rows, float pool_scale, void* stream);

/* SGD with momentum / weight decay over many tensors in one launch (mb/solver/build.py:8-37 + torch.optim.SGD's update;
 * engine/solver.py): items = device array of {float* p; const float* g; float* buf; long n} (32 bytes each), blocks = device
 * array of int2 (item, chunk): block b updates elements [chunk * ovis_sgd_chunk_elements(), ...) of its item with
 *   d = g + wd * p (when apply_weight_decay);  buf = momentum * buf + d (when momentum != 0);  p += (-lr) * (buf or d)
 * in exactly that fp32 operation order (bit-identical to the six multi-tensor passes it replaces). */
int ovis_sgd_momentum_multi_f32(const void* items, const void* blocks, int num_blocks, float lr, float weight_decay,
                                float momentum, int apply_weight_decay, void* stream);
int ovis_sgd_chunk_elements(void);

/* The backward of an identity bottleneck (mb/modeling/backbone/resnet.py:290-342: out = relu(conv3(relu(conv2(relu(conv1(x)))))
 * + x); autograd derives the backward) inside a pair-only chain, behind ONE call: g3 [m, channels] = the gradient w.r.t. the
 * block's output, already gated and in pair layout (what ovis_split_gemm_pair_rp_gated of the block above wrote); x / o1 / o2 =
 * the block's input and its conv1 / conv2 outputs (post-ReLU) in pair layout; t1 [channels, mid], t2 [mid, taps * mid],
 * t3 [mid, channels] = the transposed pair forms of the folded weights (ovis_weight_prep_pair_f32's second result);
 * scale1..3 = the folded FrozenBN scales the raw weights' gradients are multiplied with (NULL: none).  Writes
 *   g2 [m, mid], g1 [m, mid] (pair), gx [m, channels] (pair: the gated gradient for the block below),
 *   dw1 [mid, channels, 1, 1], dw2 [mid, mid, taps_h, taps_w], dw3 [channels, mid, 1, 1] (fp32)
 * by exactly the launches the separate entry points make (ovis_split_gemm_pair_gated_ws x 2, ovis_split_gemm_pair_rp_gated,
 * ovis_split_gemm_pair_tn + ovis_slab_reduce_f32 x 3: bit-identical results), the data-gradient chain on `stream`, the three
 * weight gradients on `side_stream` (NULL = `stream`) ordered against the chain by events; `stream` waits for them before
 * whatever the caller enqueues next.  channels % 128 == 0, mid_channels % 128 == 0, odd taps. */
size_t ovis_bottleneck_identity_backward_workspace_bytes(long m, int channels, int mid_channels, int taps_h, int taps_w,
                                                         int width, int config);
int ovis_bottleneck_identity_backward(
    const void* g3_pair, long g3_row_bytes, const void* x_pair, long x_row_bytes, const void* o1_pair, long o1_row_bytes,
    const void* o2_pair, long o2_row_bytes, const void* t1_pair, long t1_row_bytes, const void* t2_pair, long t2_row_bytes,
    const void* t3_pair, long t3_row_bytes, const float* scale1, const float* scale2, const float* scale3, long m, int channels,
    int mid_channels, int taps_h, int taps_w, int height, int width, void* g2_pair, void* g1_pair, void* gx_pair, float* dw1,
    float* dw2, float* dw3, void* workspace, size_t workspace_bytes, int config, void* stream, void* side_stream);

/* Weight preparation of MANY convolutions in one launch, behind the optimizer step (mb/engine/trainer.py:139; what
 * ovis_weight_prep_pair_f32 does per convolution): items = device array of 64-byte records
 *   {const float* w [N, C, T]; const float* scale [N] or NULL; void* fwd; void* bwd or NULL; long fwd_row_bytes;
 *    long bwd_row_bytes; int N, C, T, pad}
 * (fwd row n at fwd + n * fwd_row_bytes holds the pair form of w[n, :, :] * scale[n] tap-major, k = t * C + c; bwd row c at
 * bwd + c * bwd_row_bytes the transposed matrix, k' = t * N + n; a row stride above 4 * T * C writes into a wider matrix:
 * [w3 | wd] of a projection block), blocks = device array of int2 (item, tile): workgroup b converts tile (tile / (C / 32),
 * tile % (C / 32)) of 32 output x 32 input channels (ovis_weight_prep_tile() = 32) of its item, all T taps; an item has
 * (N / 32) * (C / 32) tiles.  max_taps = the largest T in the table (<= 15: the tile is staged in LDS).  N % 32 == 0,
 * C % 32 == 0, buffers 16-byte aligned: the CALLER checks (the table is opaque here).  Bit-identical to the per-convolution
 * entry. */
int ovis_weight_prep_pair_multi_f32(const void* items, const void* blocks, int num_blocks, int max_taps, void* stream);
int ovis_weight_prep_tile(void);

#ifdef __cplusplus
}
#endif
#endif /* OVIS_HIP_H_ */
