// Device-side training-target construction for the RoI heads: the chains of ~100 tiny tensor ops per
// image that the reference runs for box matching / delta encoding and for the mask targets, as ONE launch each.
// Both follow the reference's fp32 operation order (FP contraction is off for this library), so they reproduce the
// tensor-op formulation value for value (log / division are the device's correctly rounded forms).
//
//   match_encode : boxlist_iou (mb/structures/boxlist_ops.py:53-89) -> Matcher (mb/modeling/matcher.py:42-81, no
//                  low-quality matches) -> labels (box_head/loss.py:46-77, mask_head/loss.py:60-77) -> BoxCoder.encode
//                  (mb/modeling/box_coder.py:22-52).  HBM-trivial; latency-bound: one thread per proposal, the G
//                  ground-truth boxes are broadcast reads.
//   project_masks: project_masks_on_boxes (mask_head/loss.py:11-42) for binary 'mask'-mode targets: crop to the
//                  rounded box (segmentation_mask.py:117-136) and bilinear resize to MxM (F.interpolate,
//                  align_corners=False; :138-156), cast back to the mask dtype.  One thread per output pixel.
#include "ovis_common.h"

namespace {
__global__ __launch_bounds__(256) void match_encode_kernel(const float* __restrict__ gt, const long* __restrict__ gt_labels,
                                                          const float* __restrict__ prop, int G, int P, float high,
                                                          float low, int between_label_mode, float wx, float wy,
                                                          float ww, float wh, long* __restrict__ matched_idx,
                                                          long* __restrict__ labels, float* __restrict__ reg) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const float4 b = *(const float4*)(prop + 4 * p);
  const float area_b = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  float best = -1.f;
  int arg = 0;
  for (int g = 0; g < G; ++g) {
    const float4 a = *(const float4*)(gt + 4 * g);
    const float area_a = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x) + 1.f, 0.f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y) + 1.f, 0.f);
    const float inter = w * h;
    const float iou = inter / (area_a + area_b - inter);
    if (iou > best) {  // first maximum wins
      best = iou;
      arg = g;
    }
  }
  long lab = gt_labels[arg];
  int idx = arg;
  if (best < low) {
    lab = 0;
    idx = 0;  // matched.clamp(min=0) of BELOW_LOW_THRESHOLD
  } else if (best < high) {
    idx = 0;  // clamp(min=0) of BETWEEN_THRESHOLDS
    lab = between_label_mode == 0 ? -1 : gt_labels[0];  // box head: ignore (-1); mask head: the label of gt 0, as upstream
  }
  matched_idx[p] = idx;
  labels[p] = lab;
  if (reg) {
    const float4 a = *(const float4*)(gt + 4 * idx);
    const float ex_w = b.z - b.x + 1.f, ex_h = b.w - b.y + 1.f;
    const float ex_cx = b.x + 0.5f * ex_w, ex_cy = b.y + 0.5f * ex_h;
    const float gt_w = a.z - a.x + 1.f, gt_h = a.w - a.y + 1.f;
    const float gt_cx = a.x + 0.5f * gt_w, gt_cy = a.y + 0.5f * gt_h;
    float4 r;
    r.x = wx * (gt_cx - ex_cx) / ex_w;
    r.y = wy * (gt_cy - ex_cy) / ex_h;
    r.z = ww * logf(gt_w / ex_w);
    r.w = wh * logf(gt_h / ex_h);
    *(float4*)(reg + 4 * p) = r;
  }
}

__global__ __launch_bounds__(256) void project_masks_kernel(const unsigned char* __restrict__ masks,
                                                           const long* __restrict__ gt_index,
                                                           const float* __restrict__ boxes, int P, int H, int W, int M,
                                                           int is_bool, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)P * M * M;
  if (i >= total) return;
  const int dx = (int)(i % M), dy = (int)((i / M) % M), p = (int)(i / ((long)M * M));
  const float4 bx = *(const float4*)(boxes + 4 * p);
  const float b0 = rintf(bx.x), b1 = rintf(bx.y), b2 = rintf(bx.z), b3 = rintf(bx.w);  // python round(): half to even
  const float xmin = fminf(fmaxf(b0, 0.f), (float)(W - 1)), ymin = fminf(fmaxf(b1, 0.f), (float)(H - 1));
  const float xmax = fmaxf(fminf(fmaxf(b2, 0.f), (float)W), xmin + 1.f);
  const float ymax = fmaxf(fminf(fmaxf(b3, 0.f), (float)H), ymin + 1.f);
  const float w = xmax - xmin, h = ymax - ymin;
  // "size / M" with a scalar divisor is a multiplication by the reciprocal in the tensor-op formulation
  const float inv_m = 1.f / (float)M;
  const float sy = fmaxf(((float)dy + 0.5f) * (h * inv_m) - 0.5f, 0.f);
  const float sx = fmaxf(((float)dx + 0.5f) * (w * inv_m) - 0.5f, 0.f);
  float y0 = floorf(sy), x0 = floorf(sx);
  const float ly = sy - y0, lx = sx - x0;
  y0 = fminf(y0, h - 1.f);
  x0 = fminf(x0, w - 1.f);
  const float y1 = fminf(y0 + 1.f, h - 1.f), x1 = fminf(x0 + 1.f, w - 1.f);
  const unsigned char* m = masks + (long)gt_index[p] * H * W;
  const long r0 = (long)(y0 + ymin) * W, r1 = (long)(y1 + ymin) * W;
  const long c0 = (long)(x0 + xmin), c1 = (long)(x1 + xmin);
  const float v00 = (float)m[r0 + c0], v01 = (float)m[r0 + c1], v10 = (float)m[r1 + c0], v11 = (float)m[r1 + c1];
  const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  out[i] = is_bool ? (v != 0.f ? 1.f : 0.f) : (float)(unsigned char)v;
}
}  // namespace

extern "C" int ovis_match_encode_f32(const float* gt_boxes, const int64_t* gt_labels, const float* proposals,
                                     int num_gt, int num_proposals, float high_threshold, float low_threshold,
                                     int between_keeps_label, float wx, float wy, float ww, float wh,
                                     int64_t* matched_idx, int64_t* labels, float* regression_targets, void* stream) {
  if (num_gt <= 0 || num_proposals < 0) return OVIS_EINVAL;
  if (num_proposals == 0) return OVIS_OK;
  if (!gt_boxes || !gt_labels || !proposals || !matched_idx || !labels) return OVIS_EINVAL;
  if (((uintptr_t)gt_boxes & 15) || ((uintptr_t)proposals & 15) || ((uintptr_t)regression_targets & 15)) return OVIS_ERANGE;
  hipLaunchKernelGGL(match_encode_kernel, dim3((num_proposals + 255) / 256), dim3(256), 0, (hipStream_t)stream, gt_boxes,
                     (const long*)gt_labels, proposals, num_gt, num_proposals, high_threshold, low_threshold,
                     between_keeps_label, wx, wy, ww, wh, (long*)matched_idx, (long*)labels, regression_targets);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_project_masks_f32(const uint8_t* masks, const int64_t* gt_index, const float* boxes, int num,
                                      int height, int width, int resolution, int masks_are_bool, float* out,
                                      void* stream) {
  if (num < 0 || height <= 0 || width <= 0 || resolution <= 0) return OVIS_EINVAL;
  if (num == 0) return OVIS_OK;
  if (!masks || !gt_index || !boxes || !out) return OVIS_EINVAL;
  if ((uintptr_t)boxes & 15) return OVIS_ERANGE;
  const long total = (long)num * resolution * resolution;
  hipLaunchKernelGGL(project_masks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, masks,
                     (const long*)gt_index, boxes, num, height, width, resolution, masks_are_bool, out);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
