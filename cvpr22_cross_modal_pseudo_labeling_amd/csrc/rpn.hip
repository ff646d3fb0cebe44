// RPN proposal decode for gfx950 (MI355X): the per-candidate arithmetic of
// maskrcnn_benchmark/modeling/rpn/inference.py:95-114 (RPNPostProcessor.forward_for_single_feature_map) in ONE launch
// for the whole batch -- gather of the top-k candidates' regression deltas and anchors, BoxCoder.decode
// (modeling/box_coder.py:49-95), clip_to_image (structures/bounding_box.py:214-225) and the remove_small_boxes test
// (structures/boxlist_ops.py:34-49), which becomes a drop flag for the NMS instead of a compaction.  The reference
// spends ~40 tensor-op launches per image on this chain.
//
// Arithmetic follows the reference expression by expression (this library is built with -ffp-contract=off):
//   anchors:  shift + cell anchor in fp32 (rpn/anchor_generator.py:104-128: arange(0, W*stride, stride) + base)
//   decode:   widths = x2 - x1 + 1; ctr = x1 + 0.5 * widths; d = code / weight; dw, dh clamped to xform_clip;
//             pred_ctr = d * widths + ctr; pred_w = exp(dw) * widths; x1' = pred_ctr - 0.5 * pred_w;
//             x2' = pred_ctr + 0.5 * pred_w - 1
//   clip:     clamp to [0, image_w - 1] x [0, image_h - 1]   (ternary clamps: non-finite deltas stay non-finite, as torch.clamp)
//   small:    (x2 - x1 + 1 >= min_size) && (y2 - y1 + 1 >= min_size)   else drop = -1
#include "ovis_common.h"

namespace {

// clamp(min=0, max=hi) as torch evaluates it: a NaN stays a NaN (same helper as boxes.hip::box_decode_kernel)
__device__ __forceinline__ float clamp_like_torch(float v, float hi) { return v < 0.f ? 0.f : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void rpn_decode_kernel(
    const float* __restrict__ reg, long reg_sn, long reg_spos, long reg_sch, const long long* __restrict__ topk_idx,
    const float* __restrict__ cell_anchors, const float* __restrict__ image_wh, int K, int A, int feat_w, float stride,
    float wx, float wy, float ww, float wh, float xform_clip, float min_size, float4* __restrict__ boxes,
    int* __restrict__ drop) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = blockIdx.y;
  if (k >= K) return;
  const long long idx = topk_idx[(size_t)n * K + k];   // (y * feat_w + x) * A + a
  const int a = (int)(idx % A);
  const long pos = (long)(idx / A);
  const int y = (int)(pos / feat_w), x = (int)(pos % feat_w);
  const float4 ca = ((const float4*)cell_anchors)[a];
  const float sx = (float)(x * (int)stride), sy = (float)(y * (int)stride);
  const float ax1 = sx + ca.x, ay1 = sy + ca.y, ax2 = sx + ca.z, ay2 = sy + ca.w;
  const float* r = reg + (size_t)n * reg_sn + (size_t)pos * reg_spos + (size_t)(a * 4) * reg_sch;
  const float c0 = r[0], c1 = r[reg_sch], c2 = r[2 * reg_sch], c3 = r[3 * reg_sch];
  const float widths = ax2 - ax1 + 1.f, heights = ay2 - ay1 + 1.f;
  const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;
  const float dx = c0 / wx, dy = c1 / wy;
  // torch.clamp propagates NaN (fminf / fmaxf would return the finite operand and turn a diverged delta into a valid box)
  const float dw0 = c2 / ww, dh0 = c3 / wh;
  const float dw = dw0 > xform_clip ? xform_clip : dw0, dh = dh0 > xform_clip ? xform_clip : dh0;
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths, ph = expf(dh) * heights;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph;
  float x2 = pcx + 0.5f * pw - 1.f, y2 = pcy + 0.5f * ph - 1.f;
  const float iw = image_wh[2 * n] - 1.f, ih = image_wh[2 * n + 1] - 1.f;
  x1 = clamp_like_torch(x1, iw);
  y1 = clamp_like_torch(y1, ih);
  x2 = clamp_like_torch(x2, iw);
  y2 = clamp_like_torch(y2, ih);
  boxes[(size_t)n * K + k] = make_float4(x1, y1, x2, y2);
  const bool keep = (x2 - x1 + 1.f >= min_size) && (y2 - y1 + 1.f >= min_size);
  drop[(size_t)n * K + k] = keep ? 0 : -1;
}

}  // namespace

extern "C" int ovis_rpn_decode_f32(const float* box_regression, long reg_stride_image, long reg_stride_position,
                                   long reg_stride_channel, const int64_t* topk_idx, const float* cell_anchors,
                                   const float* image_wh, int num_images, int num_candidates, int anchors_per_position,
                                   int feature_width, float anchor_stride, float weight_x, float weight_y, float weight_w,
                                   float weight_h, float xform_clip, float min_size, float* boxes, int32_t* drop,
                                   void* stream) {
  if (num_images < 0 || num_candidates < 0 || anchors_per_position <= 0 || feature_width <= 0) return OVIS_EINVAL;
  if (num_images == 0 || num_candidates == 0) return OVIS_OK;
  if (!box_regression || !topk_idx || !cell_anchors || !image_wh || !boxes || !drop) return OVIS_EINVAL;
  if (anchor_stride != (float)(int)anchor_stride || num_images > 65535) return OVIS_ERANGE;
  if (((uintptr_t)cell_anchors & 15) != 0 || ((uintptr_t)boxes & 15) != 0) return OVIS_EINVAL;
  dim3 grid(ovis_ceil_div(num_candidates, 256), num_images);
  hipLaunchKernelGGL(rpn_decode_kernel, grid, dim3(256), 0, (hipStream_t)stream, box_regression, reg_stride_image,
                     reg_stride_position, reg_stride_channel, (const long long*)topk_idx, cell_anchors, image_wh,
                     num_candidates, anchors_per_position, feature_width, anchor_stride, weight_x, weight_y, weight_w,
                     weight_h, xform_clip, min_size, (float4*)boxes, drop);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
