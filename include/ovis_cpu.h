/* C ABI of libovis_cpu.so: the host-side twins of the ops the REFERENCE itself implements on the CPU, for the reference's
 * CPU-only configuration (MODEL.DEVICE=cpu: BASELINE.json configs[0], "plumbing, runs without GPU").
 *
 * The reference dispatches on the tensor's device (maskrcnn_benchmark/csrc/ROIAlign.h:11-25, csrc/nms.h:10-28):
 * device tensors go to the CUDA kernels, host tensors to csrc/cpu/ROIAlign_cpu.cpp:114-219 and csrc/cpu/nms_cpu.cpp:6-65;
 * everything else raises "Not implemented on the CPU" (csrc/ROIAlign.h:44).  `_C.py` keeps exactly that contract: a HOST
 * tensor is served here, a DEVICE tensor only ever by libovis_hip.so -- there is no fallback in either direction, and a
 * missing library raises at the first call that needs it.
 *
 * Plain pointers and sizes, int status (0 = ok, OVIS_CPU_EINVAL = bad argument).  fp32, contiguous NCHW.  `threads` <= 0
 * = all cores (OpenMP); the reference kernels are single-threaded -- the results do not depend on the thread count. */
#ifndef OVIS_CPU_H
#define OVIS_CPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OVIS_CPU_OK 0
#define OVIS_CPU_EINVAL (-1)

/* csrc/cpu/ROIAlign_cpu.cpp:114-219 (ROIAlignForward_cpu_kernel): out [R, C, PH, PW]; rois [R, 5] = (batch index, x1, y1, x2, y2)
 * in image pixels; sampling_ratio <= 0 = adaptive ceil(roi extent / bins).  Bit-identical to the reference kernel (same
 * operation order per output, no FMA contraction). */
int ovis_cpu_roi_align_forward_f32(const float* input, const float* rois, float* out, int num_rois, int batch, int channels,
                                   int height, int width, int pooled_h, int pooled_w, float spatial_scale,
                                   int sampling_ratio, int threads);

/* The transpose of the forward (the reference has NO host backward, csrc/ROIAlign.h:44; arithmetic of
 * csrc/cuda/ROIAlign_cuda.cu:178-254 with the atomics replaced by an owner-computes loop: a thread owns whole channel planes
 * and visits the RoIs in order, so the result is deterministic).  grad_input [batch, C, H, W] is overwritten. */
int ovis_cpu_roi_align_backward_f32(const float* grad_out, const float* rois, float* grad_input, int num_rois, int batch,
                                    int channels, int height, int width, int pooled_h, int pooled_w, float spatial_scale,
                                    int sampling_ratio, int threads);

/* csrc/cpu/nms_cpu.cpp:6-65: greedy suppression in descending score order (ties: lower index first), IoU with the "+1"
 * pixel convention, a box is suppressed when IoU >= threshold.  keep [K] receives the ASCENDING original indices of the
 * survivors; returns their number (< 0: bad argument). */
int ovis_cpu_nms_f32(const float* boxes, const float* scores, int num_boxes, float threshold, int64_t* keep);

/* Polygon ground truth -> mask targets: project_masks_on_boxes (mb/modeling/roi_heads/mask_head/loss.py:11-42) for
 * SegmentationMask(mode='poly') -- PolygonInstance.crop / resize / convert_to_binarymask (mb/structures/segmentation_mask.py:270-334,
 * pycocotools==2.0 rleFrPoly + merge + decode).  Host twin of ovis_project_polygon_masks_f32 (include/ovis_hip.h): coords float32
 * (x, y pairs of all polygons), polygon_start int32 [NP + 1] (float offsets), instance_start int32 [G + 1] (polygon ranges),
 * gt_index [num] int64, boxes [num, 4]; out [num, M, M] (1.0 inside). */
int ovis_cpu_project_polygon_masks_f32(const float* coords, const int32_t* polygon_start, const int32_t* instance_start,
                                       const int64_t* gt_index, const float* boxes, int num, int image_width, int image_height,
                                       int resolution, float* out, int threads);

/* Whole-image masks of polygon instances, SegmentationMask(mode='poly').convert('mask') (mb/structures/segmentation_mask.py:326-334:
 * pycocotools frPyObjects -> merge -> decode at the image size).  out [num_instances, height, width] uint8 (1 inside). */
int ovis_cpu_polygons_to_masks_u8(const float* coords, const int32_t* polygon_start, const int32_t* instance_start, int num_instances,
                                  int width, int height, uint8_t* out, int threads);

const char* ovis_cpu_version(void);

#ifdef __cplusplus
}
#endif
#endif
