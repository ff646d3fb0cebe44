// ROIPool (max pooling per RoI bin) for gfx950 -- maskrcnn_benchmark/csrc/cuda/ROIPool_cuda.cu:17-77 (forward),
// :80-108 (backward); exported by the reference's layers API (layers/roi_pool.py) but reached by no shipped config
// (SURVEY section 8f-4), so this is an API-surface op: correct, coalesced, not tuned further.
//
// Forward: one thread per output element (n, c, ph, pw), pw fastest -> neighbouring lanes scan neighbouring bins of one
// feature row (the map is L2-resident); RoI rounding (roundf = CUDA round: half away from zero), bin extents
// floor / ceil, clipping and the empty-bin convention (value 0, argmax -1) follow the reference line by line, and the
// scan order with a strict `>` keeps the FIRST maximum, so values and argmax indices are exact.
// Backward: grad_input is zero-filled on the stream, then one hardware fp32 atomic per pooled element with a valid
// argmax (overlapping RoIs collide, as in the reference).
#include <float.h>

#include "ovis_common.h"

namespace {
__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(const float* __restrict__ in, const float* __restrict__ rois,
                                                          float* __restrict__ out, int* __restrict__ argmax, long total,
                                                          int C, int H, int W, int PH, int PW, float scale) {
  for (long index = (long)blockIdx.x * 256 + threadIdx.x; index < total; index += (long)gridDim.x * 256) {
    const int pw = (int)(index % PW), ph = (int)((index / PW) % PH);
    const int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    const float* r = rois + (long)n * 5;
    const int b = (int)r[0];
    const int sw = (int)roundf(r[1] * scale), sh = (int)roundf(r[2] * scale);
    const int ew = (int)roundf(r[3] * scale), eh = (int)roundf(r[4] * scale);
    const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);  // malformed RoIs become 1x1
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    int hs = (int)floorf((float)ph * bh), ws = (int)floorf((float)pw * bw);
    int he = (int)ceilf((float)(ph + 1) * bh), we = (int)ceilf((float)(pw + 1) * bw);
    hs = min(max(hs + sh, 0), H);
    he = min(max(he + sh, 0), H);
    ws = min(max(ws + sw, 0), W);
    we = min(max(we + sw, 0), W);
    const bool empty = (he <= hs) || (we <= ws);
    float maxval = empty ? 0.f : -FLT_MAX;
    int maxidx = -1;
    const float* p = in + ((long)b * C + c) * H * W;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w) {
        const float v = p[h * W + w];
        if (v > maxval) {
          maxval = v;
          maxidx = h * W + w;
        }
      }
    out[index] = maxval;
    argmax[index] = maxidx;
  }
}

__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(const float* __restrict__ grad, const int* __restrict__ argmax,
                                                          const float* __restrict__ rois, float* __restrict__ gin,
                                                          long total, int C, int H, int W, int PH, int PW) {
  for (long index = (long)blockIdx.x * 256 + threadIdx.x; index < total; index += (long)gridDim.x * 256) {
    const int a = argmax[index];
    if (a == -1) continue;
    const int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    const int b = (int)rois[(long)n * 5];
    atomicAdd(gin + ((long)b * C + c) * H * W + a, grad[index]);
  }
}
}  // namespace

extern "C" int ovis_roi_pool_forward_f32(const float* input, const float* rois, float* output, int32_t* argmax,
                                         int num_rois, int batch, int channels, int height, int width, int pooled_h,
                                         int pooled_w, float spatial_scale, void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return OVIS_EINVAL;
  if (num_rois == 0 || channels == 0) return OVIS_OK;
  if (!input || !rois || !output || !argmax) return OVIS_EINVAL;
  const long total = (long)num_rois * channels * pooled_h * pooled_w;
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 16L * OVIS_NUM_CU ? blocks : 16L * OVIS_NUM_CU);
  hipLaunchKernelGGL(roi_pool_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, input, rois, output, argmax,
                     total, channels, height, width, pooled_h, pooled_w, spatial_scale);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_roi_pool_backward_f32(const float* grad_output, const int32_t* argmax, const float* rois,
                                          float* grad_input, int num_rois, int batch, int channels, int height,
                                          int width, int pooled_h, int pooled_w, void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return OVIS_EINVAL;
  if (batch == 0 || channels == 0) return OVIS_OK;
  if (!grad_input) return OVIS_EINVAL;
  OVIS_HIP_TRY(hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)batch * channels * height * width, (hipStream_t)stream));
  if (num_rois == 0) return OVIS_OK;
  if (!grad_output || !argmax || !rois) return OVIS_EINVAL;
  const long total = (long)num_rois * channels * pooled_h * pooled_w;
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 16L * OVIS_NUM_CU ? blocks : 16L * OVIS_NUM_CU);
  hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_output, argmax, rois,
                     grad_input, total, channels, height, width, pooled_h, pooled_w);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
