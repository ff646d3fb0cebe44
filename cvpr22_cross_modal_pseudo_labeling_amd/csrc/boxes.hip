// Box-head box arithmetic for gfx950 (MI355X), one launch each where the reference spends dozens of tensor ops:
//
//  * box_decode_kernel: BoxCoder.decode (maskrcnn_benchmark/modeling/box_coder.py:49-95) of [R, 4K] regression deltas
//    against [R, 4] boxes, optionally followed by clip_to_image (structures/bounding_box.py:214-225) against the size of
//    the image a row belongs to -- the chain of the box post-processor (roi_heads/box_head/inference.py:40-88).
//  * smooth_l1_picked_kernel: the box-regression loss of the student heads (roi_heads/box_head/loss.py:147-170):
//    gather of the positives' 4 regression columns, smooth_l1_loss(beta, size_average=False) / denominator
//    (layers/smooth_l1_loss.py:6-16), forward and gradient in one pass.
//  * rois_from_boxes_kernel: Pooler.convert_to_roi_format (modeling/poolers.py:73-86), the [R, 5] RoI rows of a batch.
//
// The decode follows the reference expression by expression (this library is built with -ffp-contract=off), so it agrees
// bit for bit with the tensor-op form on the same device.
#include "ovis_common.h"

namespace {

struct ClipImages {      // rows [start[i], start[i+1]) belong to image i
  int count;
  int start[OVIS_BOX_DECODE_MAX_IMAGES + 1];
  float w[OVIS_BOX_DECODE_MAX_IMAGES], h[OVIS_BOX_DECODE_MAX_IMAGES];
};

__device__ __forceinline__ float clamp_like_torch(float v, float hi) { return v < 0.f ? 0.f : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void box_decode_kernel(const float* __restrict__ codes, long codes_rs,
                                                        const float* __restrict__ boxes, long boxes_rs,
                                                        float4* __restrict__ out, long R, int K, float wx, float wy,
                                                        float ww, float wh, float xform_clip, ClipImages clip) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * K) return;
  const long r = i / K;
  const int k = (int)(i - r * K);
  const float* b = boxes + r * boxes_rs;
  const float bx1 = b[0], by1 = b[1], bx2 = b[2], by2 = b[3];
  const float* c = codes + r * codes_rs + 4 * k;
  const float widths = bx2 - bx1 + 1.f, heights = by2 - by1 + 1.f;
  const float ctr_x = bx1 + 0.5f * widths, ctr_y = by1 + 0.5f * heights;
  const float dx = c[0] / wx, dy = c[1] / wy;
  float dw = c[2] / ww, dh = c[3] / wh;
  dw = dw > xform_clip ? xform_clip : dw;  // torch.clamp(max=): a NaN stays a NaN
  dh = dh > xform_clip ? xform_clip : dh;
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths, ph = expf(dh) * heights;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph;
  float x2 = pcx + 0.5f * pw - 1.f, y2 = pcy + 0.5f * ph - 1.f;
  if (clip.count > 0) {
    int img = 0;
    while (img + 1 < clip.count && r >= clip.start[img + 1]) ++img;
    const float iw = clip.w[img] - 1.f, ih = clip.h[img] - 1.f;
    x1 = clamp_like_torch(x1, iw);
    y1 = clamp_like_torch(y1, ih);
    x2 = clamp_like_torch(x2, iw);
    y2 = clamp_like_torch(y2, ih);
  }
  out[i] = make_float4(x1, y1, x2, y2);
}

struct ImageBoxes {      // image i: boxes[i] points at [start[i+1] - start[i], 4] floats
  int count;
  int start[OVIS_BOX_DECODE_MAX_IMAGES + 1];
  int id[OVIS_BOX_DECODE_MAX_IMAGES];
  const float* boxes[OVIS_BOX_DECODE_MAX_IMAGES];
};

// Pooler.convert_to_roi_format (modeling/poolers.py:73-86): rows (image index, x1, y1, x2, y2) of every image's boxes
__global__ __launch_bounds__(256) void rois_from_boxes_kernel(ImageBoxes in, float* __restrict__ rois) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= in.start[in.count]) return;
  int img = 0;
  while (img + 1 < in.count && r >= in.start[img + 1]) ++img;
  const float* b = in.boxes[img] + 4 * (size_t)(r - in.start[img]);
  float* o = rois + 5 * (size_t)r;
  o[0] = (float)in.id[img];
  o[1] = b[0];
  o[2] = b[1];
  o[3] = b[2];
  o[4] = b[3];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ONE workgroup: a few hundred positives x 4 columns; fixed summation order -> deterministic
__global__ __launch_bounds__(1024) void smooth_l1_picked_kernel(
    const float* __restrict__ reg, long reg_rs, const float* __restrict__ targets, long tgt_rs,
    const long long* __restrict__ pos, const long long* __restrict__ labels, int num_pos, int num_columns, int column0,
    float beta, float denominator, float* __restrict__ loss, float* __restrict__ dreg) {
  __shared__ float part[16];
  const float inv_denominator = 1.f / denominator;
  float acc = 0.f;
  for (int e = threadIdx.x; e < num_pos * 4; e += 1024) {
    const long p = pos[e >> 2];
    const int c = e & 3;
    const long col = (labels ? 4 * labels[p] : (long)column0) + c;
    const float d = reg[p * reg_rs + col] - targets[p * tgt_rs + c];
    const float n = fabsf(d);
    const bool quad = n < beta;
    acc += quad ? 0.5f * n * n / beta : n - 0.5f * beta;
    if (dreg) {
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      dreg[p * num_columns + col] = (quad ? d / beta : sgn) * inv_denominator;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += part[k];
    loss[0] = t / denominator;
  }
}

}  // namespace

extern "C" int ovis_box_decode_f32(const float* rel_codes, long codes_row_stride, const float* boxes,
                                   long boxes_row_stride, long num_rows, int boxes_per_row, float weight_x,
                                   float weight_y, float weight_w, float weight_h, float xform_clip, int num_images,
                                   const int32_t* rows_per_image, const float* image_wh, float* decoded, void* stream) {
  if (num_rows < 0 || boxes_per_row <= 0 || num_images < 0 || num_images > OVIS_BOX_DECODE_MAX_IMAGES)
    return num_images > OVIS_BOX_DECODE_MAX_IMAGES ? OVIS_ERANGE : OVIS_EINVAL;
  if (num_rows == 0) return OVIS_OK;
  if (!rel_codes || !boxes || !decoded || ((uintptr_t)decoded & 15) != 0) return OVIS_EINVAL;
  if (num_rows * (long)boxes_per_row > 0x7fffffffL * 256) return OVIS_ERANGE;
  ClipImages clip;
  clip.count = num_images;
  if (num_images > 0) {
    if (!rows_per_image || !image_wh) return OVIS_EINVAL;  // HOST arrays: a handful of scalars that travel as arguments
    long at = 0;
    for (int i = 0; i < num_images; ++i) {
      if (rows_per_image[i] < 0) return OVIS_EINVAL;
      clip.start[i] = (int)at;
      clip.w[i] = image_wh[2 * i];
      clip.h[i] = image_wh[2 * i + 1];
      at += rows_per_image[i];
    }
    clip.start[num_images] = (int)at;
    if (at != num_rows) return OVIS_EINVAL;
  }
  hipLaunchKernelGGL(box_decode_kernel, dim3(ovis_ceil_div(num_rows * boxes_per_row, 256)), dim3(256), 0,
                     (hipStream_t)stream, rel_codes, codes_row_stride, boxes, boxes_row_stride, (float4*)decoded,
                     num_rows, boxes_per_row, weight_x, weight_y, weight_w, weight_h, xform_clip, clip);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_smooth_l1_picked_fwd_bwd_f32(const float* box_regression, long regression_row_stride, int num_rows,
                                                 int num_columns, const float* regression_targets,
                                                 long targets_row_stride, const int64_t* positives,
                                                 const int64_t* labels, int num_positives, int column0, float beta,
                                                 float denominator, float* loss, float* grad_regression,
                                                 void* stream) {
  if (num_rows < 0 || num_columns < 4 || num_positives < 0 || !loss || !(beta > 0.f) || !(denominator > 0.f))
    return OVIS_EINVAL;
  if (!labels && (column0 < 0 || column0 + 4 > num_columns)) return OVIS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (grad_regression)  // a dense [num_rows, num_columns], whatever the row stride of box_regression
    OVIS_HIP_TRY(hipMemsetAsync(grad_regression, 0, sizeof(float) * (size_t)num_rows * num_columns, s));
  if (num_positives == 0) {
    OVIS_HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), s));
    return OVIS_OK;
  }
  if (!box_regression || !regression_targets || !positives) return OVIS_EINVAL;
  hipLaunchKernelGGL(smooth_l1_picked_kernel, dim3(1), dim3(1024), 0, s, box_regression, regression_row_stride,
                     regression_targets, targets_row_stride, (const long long*)positives, (const long long*)labels,
                     num_positives, num_columns, column0, beta, denominator, loss, grad_regression);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_rois_from_boxes_f32(const float* const* boxes, const int32_t* boxes_per_image,
                                        const int32_t* image_ids, int num_images, float* rois, void* stream) {
  if (num_images < 0 || (num_images > 0 && (!boxes || !boxes_per_image))) return OVIS_EINVAL;
  size_t row = 0;
  for (int first = 0; first < num_images; first += OVIS_BOX_DECODE_MAX_IMAGES) {  // one launch per 16 images
    ImageBoxes in;
    in.count = num_images - first < OVIS_BOX_DECODE_MAX_IMAGES ? num_images - first : OVIS_BOX_DECODE_MAX_IMAGES;
    long at = 0;
    for (int i = 0; i < in.count; ++i) {
      if (boxes_per_image[first + i] < 0 || (boxes_per_image[first + i] > 0 && !boxes[first + i])) return OVIS_EINVAL;
      in.start[i] = (int)at;
      in.boxes[i] = boxes[first + i];
      in.id[i] = image_ids ? image_ids[first + i] : first + i;
      at += boxes_per_image[first + i];
    }
    in.start[in.count] = (int)at;
    if (at > 0) {
      if (!rois) return OVIS_EINVAL;
      hipLaunchKernelGGL(rois_from_boxes_kernel, dim3(ovis_ceil_div(at, 256)), dim3(256), 0, (hipStream_t)stream, in,
                         rois + 5 * row);
      OVIS_LAUNCH_CHECK();
    }
    row += (size_t)at;
  }
  return OVIS_OK;
}
