// Deformable convolution v1 / v2 (modulated) building blocks for gfx950, fp32.
//
// Semantics: maskrcnn_benchmark/csrc/cuda/deform_conv_kernel_cuda.cu
//   :92-122  bilinear sample with zero outside the map      :125-195 gradient / coordinate weights
//   :198-250 deformable_im2col    :287-342 col2im    :381-443 col2im_coord
//   :578-640, :643-700, :703-774  the modulated (mask) variants
// One set of kernels serves both versions: `mask == nullptr` is v1.  Column layout (the reference's):
//   columns[(c*KH*KW + i*KW + j)][b][h_out][w_out],  b in [0, nb) images of the current chunk.
// The GEMMs around these kernels run on ovis_gemm_f32 (fp32 matrix cores) directly on this layout with
// strided operands, so the reference's output_buffer / transpose / copy_ steps (deform_conv_cuda.cu:218-252)
// do not exist here.  All three kernels are HBM-bound on the column buffer.
#include "ovis_common.h"

namespace {

constexpr int kThreads = 256;

struct DcnGeom {
  int C, H, W, KH, KW, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, nb, Ho, Wo;
};

__device__ __forceinline__ float bilinear_zero(const float* __restrict__ im, int H, int W, float h, float w) {
  const int hl = (int)floorf(h), wl = (int)floorf(w);
  const int hh = hl + 1, wh = wl + 1;
  const float lh = h - hl, lw = w - wl, uh = 1.f - lh, uw = 1.f - lw;
  const float v1 = (hl >= 0 && wl >= 0) ? im[hl * W + wl] : 0.f;
  const float v2 = (hl >= 0 && wh <= W - 1) ? im[hl * W + wh] : 0.f;
  const float v3 = (hh <= H - 1 && wl >= 0) ? im[hh * W + wl] : 0.f;
  const float v4 = (hh <= H - 1 && wh <= W - 1) ? im[hh * W + wh] : 0.f;
  return uh * uw * v1 + uh * lw * v2 + lh * uw * v3 + lh * lw * v4;
}

// d(sample)/d(h) (dir 0) or d/d(w) (dir 1): get_coordinate_weight, :150-195
__device__ __forceinline__ float coord_weight(const float* __restrict__ im, int H, int W, float h, float w,
                                              int dir) {
  if (h <= -1.f || h >= (float)H || w <= -1.f || w >= (float)W) return 0.f;
  const int hl = (int)floorf(h), wl = (int)floorf(w);
  const int hh = hl + 1, wh = wl + 1;
  float r = 0.f;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) r += -1.f * (wl + 1 - w) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) r += -1.f * (w - wl) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) r += (wl + 1 - w) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) r += (w - wl) * im[hh * W + wh];
  } else {
    if (hl >= 0 && wl >= 0) r += -1.f * (hl + 1 - h) * im[hl * W + wl];
    if (hl >= 0 && wh <= W - 1) r += (hl + 1 - h) * im[hl * W + wh];
    if (hh <= H - 1 && wl >= 0) r += -1.f * (h - hl) * im[hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) r += (h - hl) * im[hh * W + wh];
  }
  return r;
}

// one lane per (c, b, h_out, w_out); loops the KH*KW taps.  Stores are lane-contiguous in w_out.
__global__ __launch_bounds__(kThreads) void dcn_im2col_kernel(const float* __restrict__ im,
                                                             const float* __restrict__ offset,
                                                             const float* __restrict__ mask,
                                                             float* __restrict__ col, DcnGeom g, long total) {
  const int K = g.KH * g.KW;
  const long plane = (long)g.Ho * g.Wo;
  const int cpg = g.C / g.dg;
  for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
    const int w = (int)(idx % g.Wo);
    const int h = (int)((idx / g.Wo) % g.Ho);
    const int b = (int)((idx / plane) % g.nb);
    const int c = (int)(idx / (plane * g.nb));
    const int grp = c / cpg;
    const float* imc = im + ((long)b * g.C + c) * g.H * g.W;
    const float* off = offset + ((long)b * g.dg + grp) * 2 * K * plane + (long)h * g.Wo + w;
    const float* msk = mask ? mask + ((long)b * g.dg + grp) * K * plane + (long)h * g.Wo + w : nullptr;
    float* dst = col + (((long)c * K) * g.nb + b) * plane + (long)h * g.Wo + w;
    const int h_in = h * g.stride_h - g.pad_h, w_in = w * g.stride_w - g.pad_w;
    for (int i = 0; i < g.KH; ++i)
      for (int j = 0; j < g.KW; ++j) {
        const int k = i * g.KW + j;
        const float hi = h_in + i * g.dil_h + off[(long)(2 * k) * plane];
        const float wi = w_in + j * g.dil_w + off[(long)(2 * k + 1) * plane];
        float v = 0.f;
        if (hi > -1.f && wi > -1.f && hi < (float)g.H && wi < (float)g.W) v = bilinear_zero(imc, g.H, g.W, hi, wi);
        if (msk) v *= msk[(long)k * plane];
        dst[(long)k * g.nb * plane] = v;
      }
  }
}

// one lane per column element: scatters d(col) onto the (up to) four input cells it was sampled from.
__global__ __launch_bounds__(kThreads) void dcn_col2im_kernel(const float* __restrict__ col,
                                                             const float* __restrict__ offset,
                                                             const float* __restrict__ mask,
                                                             float* __restrict__ grad_im, DcnGeom g, long total) {
  const int K = g.KH * g.KW;
  const long plane = (long)g.Ho * g.Wo;
  const int cpg = g.C / g.dg;
  for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
    const int w = (int)(idx % g.Wo);
    const int h = (int)((idx / g.Wo) % g.Ho);
    const int b = (int)((idx / plane) % g.nb);
    const int k = (int)((idx / (plane * g.nb)) % K);
    const int c = (int)(idx / (plane * g.nb * K));
    const int i = k / g.KW, j = k - i * g.KW;
    const int grp = c / cpg;
    const float* off = offset + ((long)b * g.dg + grp) * 2 * K * plane + (long)h * g.Wo + w;
    const float hi = h * g.stride_h - g.pad_h + i * g.dil_h + off[(long)(2 * k) * plane];
    const float wi = w * g.stride_w - g.pad_w + j * g.dil_w + off[(long)(2 * k + 1) * plane];
    if (hi <= -1.f || wi <= -1.f || hi >= (float)g.H || wi >= (float)g.W) continue;
    float top = col[idx];
    if (mask) top *= mask[((long)b * g.dg + grp) * K * plane + (long)k * plane + (long)h * g.Wo + w];
    const int hl = (int)floorf(hi), wl = (int)floorf(wi);
    float* gim = grad_im + ((long)b * g.C + c) * g.H * g.W;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int y = hl + dy, x = wl + dx;
        if (y < 0 || y >= g.H || x < 0 || x >= g.W) continue;
        const float wy = dy ? (hi - hl) : (hl + 1 - hi), wx = dx ? (wi - wl) : (wl + 1 - wi);
        atomicAdd(gim + y * g.W + x, wy * wx * top);
      }
  }
}

// one lane per (b, offset channel, h_out, w_out): d/d(offset) and, for v2, d/d(mask)
__global__ __launch_bounds__(kThreads) void dcn_col2im_coord_kernel(
    const float* __restrict__ col, const float* __restrict__ im, const float* __restrict__ offset,
    const float* __restrict__ mask, float* __restrict__ grad_offset, float* __restrict__ grad_mask, DcnGeom g,
    long total) {
  const int K = g.KH * g.KW;
  const long plane = (long)g.Ho * g.Wo;
  const int cpg = g.C / g.dg;
  const int och = g.dg * 2 * K;
  for (long idx = (long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long)gridDim.x * kThreads) {
    const int w = (int)(idx % g.Wo);
    const int h = (int)((idx / g.Wo) % g.Ho);
    const int oc = (int)((idx / plane) % och);
    const int b = (int)(idx / (plane * och));
    const int grp = oc / (2 * K);
    const int rem = oc - grp * 2 * K;
    const int k = rem >> 1, dir = rem & 1;
    const int i = k / g.KW, j = k - i * g.KW;
    const float* off = offset + ((long)b * g.dg + grp) * 2 * K * plane + (long)h * g.Wo + w;
    float hi = h * g.stride_h - g.pad_h + i * g.dil_h + off[(long)(2 * k) * plane];
    float wi = w * g.stride_w - g.pad_w + j * g.dil_w + off[(long)(2 * k + 1) * plane];
    const bool inside = !(hi <= -1.f || wi <= -1.f || hi >= (float)g.H || wi >= (float)g.W);
    const float m = mask ? mask[((long)b * g.dg + grp) * K * plane + (long)k * plane + (long)h * g.Wo + w] : 1.f;
    float val = 0.f, mval = 0.f;
    if (inside) {
      for (int cc = 0; cc < cpg; ++cc) {
        const int c = grp * cpg + cc;
        const float* imc = im + ((long)b * g.C + c) * g.H * g.W;
        const float cv = col[(((long)c * K + k) * g.nb + b) * plane + (long)h * g.Wo + w];
        val += coord_weight(imc, g.H, g.W, hi, wi, dir) * cv * m;
        if (grad_mask && dir == 0) mval += cv * bilinear_zero(imc, g.H, g.W, hi, wi);
      }
    }
    grad_offset[idx] = val;
    if (grad_mask && dir == 0)
      grad_mask[((long)b * g.dg + grp) * K * plane + (long)k * plane + (long)h * g.Wo + w] = mval;
  }
}

int check(const DcnGeom& g) {
  if (g.C <= 0 || g.H <= 0 || g.W <= 0 || g.KH <= 0 || g.KW <= 0 || g.stride_h <= 0 || g.stride_w <= 0 ||
      g.dil_h <= 0 || g.dil_w <= 0 || g.dg <= 0 || g.C % g.dg != 0 || g.nb < 0 || g.Ho <= 0 || g.Wo <= 0)
    return OVIS_EINVAL;
  return OVIS_OK;
}

DcnGeom make(int C, int H, int W, int KH, int KW, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
             int dil_w, int dg, int nb) {
  DcnGeom g{C, H, W, KH, KW, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, nb, 0, 0};
  g.Ho = (H + 2 * pad_h - (dil_h * (KH - 1) + 1)) / stride_h + 1;
  g.Wo = (W + 2 * pad_w - (dil_w * (KW - 1) + 1)) / stride_w + 1;
  return g;
}

unsigned grid_for(long total) {
  long b = (total + kThreads - 1) / kThreads;
  return (unsigned)(b < OVIS_NUM_CU * 32 ? b : OVIS_NUM_CU * 32);
}

}  // namespace

extern "C" int ovis_deform_im2col_f32(const float* input, const float* offset, const float* mask, float* columns,
                                      int num_images, int channels, int height, int width, int kernel_h,
                                      int kernel_w, int pad_h, int pad_w, int stride_h, int stride_w,
                                      int dilation_h, int dilation_w, int deformable_group, void* stream) {
  const DcnGeom g = make(channels, height, width, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w,
                         dilation_h, dilation_w, deformable_group, num_images);
  if (check(g)) return OVIS_EINVAL;
  if (num_images == 0) return OVIS_OK;
  if (!input || !offset || !columns) return OVIS_EINVAL;
  const long total = (long)channels * num_images * g.Ho * g.Wo;
  hipLaunchKernelGGL(dcn_im2col_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, input,
                     offset, mask, columns, g, total);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_deform_col2im_f32(const float* columns, const float* offset, const float* mask,
                                      float* grad_input, int num_images, int channels, int height, int width,
                                      int kernel_h, int kernel_w, int pad_h, int pad_w, int stride_h,
                                      int stride_w, int dilation_h, int dilation_w, int deformable_group,
                                      void* stream) {
  const DcnGeom g = make(channels, height, width, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w,
                         dilation_h, dilation_w, deformable_group, num_images);
  if (check(g)) return OVIS_EINVAL;
  if (num_images == 0) return OVIS_OK;
  if (!columns || !offset || !grad_input) return OVIS_EINVAL;
  const long total = (long)channels * kernel_h * kernel_w * num_images * g.Ho * g.Wo;
  hipLaunchKernelGGL(dcn_col2im_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, columns,
                     offset, mask, grad_input, g, total);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_deform_col2im_coord_f32(const float* columns, const float* input, const float* offset,
                                            const float* mask, float* grad_offset, float* grad_mask,
                                            int num_images, int channels, int height, int width, int kernel_h,
                                            int kernel_w, int pad_h, int pad_w, int stride_h, int stride_w,
                                            int dilation_h, int dilation_w, int deformable_group,
                                            void* stream) {
  const DcnGeom g = make(channels, height, width, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w,
                         dilation_h, dilation_w, deformable_group, num_images);
  if (check(g)) return OVIS_EINVAL;
  if (num_images == 0) return OVIS_OK;
  if (!columns || !input || !offset || !grad_offset || (grad_mask && !mask)) return OVIS_EINVAL;
  const long total = (long)num_images * deformable_group * 2 * kernel_h * kernel_w * g.Ho * g.Wo;
  hipLaunchKernelGGL(dcn_col2im_coord_kernel, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream,
                     columns, input, offset, mask, grad_offset, grad_mask, g, total);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
