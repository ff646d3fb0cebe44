// Deformable position-sensitive RoI pooling for gfx950 -- maskrcnn_benchmark/csrc/deform_pool.h:11-70, kernels
// csrc/cuda/deform_pool_kernel_cuda.cu:31-139 (forward), :141-263 (backward).  Exported by the reference's layers API
// (layers/dcn/deform_pool_func.py) but reached by no shipped config: an API-surface op -- correct,
// coalesced on the output side, not tuned further.
//
// One thread per output element (n, ctop, ph, pw), pw fastest.  The bin geometry follows the reference's fp32 expression
// order including its promotions to double (`- 0.5`, `max(.., 0.1)`, the border tests and the clamp are evaluated in
// double there because the literals are doubles), so the discrete decisions (floor / ceil cells, samples inside the map,
// sample count) are the reference's.  Backward: one hardware fp32 atomic per touched input cell and per offset
// component, like the reference (the offset gradient deliberately ignores the border clamp, as the reference does).
#include "ovis_common.h"

namespace {

struct PsGeom {
  int b, gw, gh, class_id, part_h, part_w;
  float roi_w, roi_h, wstart, hstart, sub_w, sub_h;
};

__device__ __forceinline__ PsGeom ps_geom(const float* __restrict__ rois, const float* __restrict__ trans, int n, int ctop,
                                          int ph, int pw, float scale, int P, int no_trans, float trans_std,
                                          int sample_per_part, int group_size, int part_size, int num_classes,
                                          int channels_each_class) {
  PsGeom g;
  const float* r = rois + (long)n * 5;
  g.b = (int)r[0];
  const float start_w = (float)((double)(roundf(r[1]) * scale) - 0.5);
  const float start_h = (float)((double)(roundf(r[2]) * scale) - 0.5);
  const float end_w = (float)((double)((float)((double)roundf(r[3]) + 1.) * scale) - 0.5);
  const float end_h = (float)((double)((float)((double)roundf(r[4]) + 1.) * scale) - 0.5);
  g.roi_w = (float)fmax((double)(end_w - start_w), 0.1);
  g.roi_h = (float)fmax((double)(end_h - start_h), 0.1);
  const float bin_h = g.roi_h / (float)P, bin_w = g.roi_w / (float)P;
  g.sub_h = bin_h / (float)sample_per_part;
  g.sub_w = bin_w / (float)sample_per_part;
  g.part_h = (int)floorf((float)ph / (float)P * (float)part_size);
  g.part_w = (int)floorf((float)pw / (float)P * (float)part_size);
  g.class_id = ctop / channels_each_class;
  float tx = 0.f, ty = 0.f;
  if (!no_trans) {
    const long base = ((long)n * num_classes + g.class_id) * 2;
    tx = trans[((base)*part_size + g.part_h) * part_size + g.part_w] * trans_std;
    ty = trans[((base + 1) * part_size + g.part_h) * part_size + g.part_w] * trans_std;
  }
  g.wstart = (float)pw * bin_w + start_w;
  g.wstart += tx * g.roi_w;
  g.hstart = (float)ph * bin_h + start_h;
  g.hstart += ty * g.roi_h;
  int gw = (int)floorf((float)pw * (float)group_size / (float)P);
  int gh = (int)floorf((float)ph * (float)group_size / (float)P);
  g.gw = min(max(gw, 0), group_size - 1);
  g.gh = min(max(gh, 0), group_size - 1);
  return g;
}

// sample (iw, ih) of the bin: false when it falls outside the map, else the clamped position
__device__ __forceinline__ bool ps_sample(const PsGeom& g, int iw, int ih, int W, int H, float* w_out, float* h_out) {
  const float w = g.wstart + (float)iw * g.sub_w;
  const float h = g.hstart + (float)ih * g.sub_h;
  if ((double)w < -0.5 || (double)w > (double)W - 0.5 || (double)h < -0.5 || (double)h > (double)H - 0.5) return false;
  *w_out = (float)fmin(fmax((double)w, 0.), (double)W - 1.);
  *h_out = (float)fmin(fmax((double)h, 0.), (double)H - 1.);
  return true;
}

__global__ __launch_bounds__(256) void deform_psroi_fwd_kernel(
    const float* __restrict__ data, const float* __restrict__ rois, const float* __restrict__ trans,
    float* __restrict__ out, float* __restrict__ out_count, long total, float scale, int C, int H, int W, int P,
    int no_trans, float trans_std, int sample_per_part, int output_dim, int group_size, int part_size, int num_classes,
    int channels_each_class) {
  for (long index = (long)blockIdx.x * 256 + threadIdx.x; index < total; index += (long)gridDim.x * 256) {
    const int pw = (int)(index % P), ph = (int)((index / P) % P);
    const int ctop = (int)((index / P / P) % output_dim), n = (int)(index / P / P / output_dim);
    const PsGeom g = ps_geom(rois, trans, n, ctop, ph, pw, scale, P, no_trans, trans_std, sample_per_part, group_size,
                             part_size, num_classes, channels_each_class);
    const int c = (ctop * group_size + g.gh) * group_size + g.gw;
    const float* plane = data + ((long)g.b * C + c) * H * W;
    float sum = 0.f;
    int count = 0;
    for (int ih = 0; ih < sample_per_part; ++ih)
      for (int iw = 0; iw < sample_per_part; ++iw) {
        float w, h;
        if (!ps_sample(g, iw, ih, W, H, &w, &h)) continue;
        const int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
        const float dx = w - (float)x1, dy = h - (float)y1;
        const float v11 = plane[y1 * W + x1], v12 = plane[y2 * W + x1], v21 = plane[y1 * W + x2], v22 = plane[y2 * W + x2];
        const float val = (1.f - dx) * (1.f - dy) * v11 + (1.f - dx) * dy * v12 + dx * (1.f - dy) * v21 + dx * dy * v22;
        sum += val;
        ++count;
      }
    out[index] = count == 0 ? 0.f : sum / (float)count;
    out_count[index] = (float)count;
  }
}

__global__ __launch_bounds__(256) void deform_psroi_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ out_count, const float* __restrict__ data,
    const float* __restrict__ rois, const float* __restrict__ trans, float* __restrict__ gdata, float* __restrict__ gtrans,
    long total, float scale, int C, int H, int W, int P, int no_trans, float trans_std, int sample_per_part,
    int output_dim, int group_size, int part_size, int num_classes, int channels_each_class) {
  for (long index = (long)blockIdx.x * 256 + threadIdx.x; index < total; index += (long)gridDim.x * 256) {
    const float cnt = out_count[index];
    if (cnt <= 0.f) continue;
    const int pw = (int)(index % P), ph = (int)((index / P) % P);
    const int ctop = (int)((index / P / P) % output_dim), n = (int)(index / P / P / output_dim);
    const PsGeom g = ps_geom(rois, trans, n, ctop, ph, pw, scale, P, no_trans, trans_std, sample_per_part, group_size,
                             part_size, num_classes, channels_each_class);
    const float diff = gout[index] / cnt;
    const int c = (ctop * group_size + g.gh) * group_size + g.gw;
    const long plane_off = ((long)g.b * C + c) * H * W;
    const float* plane = data + plane_off;
    float* gplane = gdata + plane_off;
    float acc_x = 0.f, acc_y = 0.f;
    for (int ih = 0; ih < sample_per_part; ++ih)
      for (int iw = 0; iw < sample_per_part; ++iw) {
        float w, h;
        if (!ps_sample(g, iw, ih, W, H, &w, &h)) continue;
        const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        const float dx = w - (float)x0, dy = h - (float)y0;
        atomicAdd(gplane + y0 * W + x0, (1.f - dx) * (1.f - dy) * diff);
        atomicAdd(gplane + y1 * W + x0, (1.f - dx) * dy * diff);
        atomicAdd(gplane + y0 * W + x1, dx * (1.f - dy) * diff);
        atomicAdd(gplane + y1 * W + x1, dx * dy * diff);
        if (no_trans) continue;
        const float u00 = plane[y0 * W + x0], u01 = plane[y1 * W + x0], u10 = plane[y0 * W + x1], u11 = plane[y1 * W + x1];
        acc_x += (u11 * dy + u10 * (1.f - dy) - u01 * dy - u00 * (1.f - dy)) * trans_std * diff * g.roi_w;
        acc_y += (u11 * dx + u01 * (1.f - dx) - u10 * dx - u00 * (1.f - dx)) * trans_std * diff * g.roi_h;
      }
    if (!no_trans) {  // one atomic per component and bin (the reference issues one per sample)
      const long base = ((long)n * num_classes + g.class_id) * 2;
      atomicAdd(gtrans + ((base)*part_size + g.part_h) * part_size + g.part_w, acc_x);
      atomicAdd(gtrans + ((base + 1) * part_size + g.part_h) * part_size + g.part_w, acc_y);
    }
  }
}

int check_args(int num_rois, int batch, int channels, int height, int width, int channels_trans, int no_trans,
               int output_dim, int group_size, int pooled_size, int part_size, int sample_per_part) {
  if (num_rois < 0 || batch < 0 || channels <= 0 || height <= 0 || width <= 0 || output_dim <= 0 || group_size <= 0 ||
      pooled_size <= 0 || part_size <= 0 || sample_per_part <= 0)
    return OVIS_EINVAL;
  if ((long)output_dim * group_size * group_size > channels) return OVIS_EINVAL;  // position-sensitive planes must exist
  if (!no_trans && (channels_trans < 2 || channels_trans % 2 != 0 || output_dim % (channels_trans / 2) != 0))
    return OVIS_EINVAL;
  return OVIS_OK;
}

}  // namespace

extern "C" int ovis_deform_psroi_pool_forward_f32(const float* data, const float* rois, const float* trans, float* out,
                                                  float* out_count, int num_rois, int batch, int channels, int height,
                                                  int width, int channels_trans, int no_trans, float spatial_scale,
                                                  int output_dim, int group_size, int pooled_size, int part_size,
                                                  int sample_per_part, float trans_std, void* stream) {
  int rc = check_args(num_rois, batch, channels, height, width, channels_trans, no_trans, output_dim, group_size,
                      pooled_size, part_size, sample_per_part);
  if (rc != OVIS_OK) return rc;
  const long total = (long)num_rois * output_dim * pooled_size * pooled_size;
  if (total == 0) return OVIS_OK;
  if (!data || !rois || !out || !out_count || (!no_trans && !trans)) return OVIS_EINVAL;
  const int num_classes = no_trans ? 1 : channels_trans / 2;
  const int cec = no_trans ? output_dim : output_dim / num_classes;
  const unsigned blocks = (unsigned)min((total + 255) / 256, (long)OVIS_NUM_CU * 32);
  hipLaunchKernelGGL(deform_psroi_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, data, rois, trans, out,
                     out_count, total, spatial_scale, channels, height, width, pooled_size, no_trans ? 1 : 0, trans_std,
                     sample_per_part, output_dim, group_size, part_size, num_classes, cec);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_deform_psroi_pool_backward_f32(const float* grad_out, const float* out_count, const float* data,
                                                   const float* rois, const float* trans, float* grad_data,
                                                   float* grad_trans, int num_rois, int batch, int channels, int height,
                                                   int width, int channels_trans, int no_trans, float spatial_scale,
                                                   int output_dim, int group_size, int pooled_size, int part_size,
                                                   int sample_per_part, float trans_std, void* stream) {
  int rc = check_args(num_rois, batch, channels, height, width, channels_trans, no_trans, output_dim, group_size,
                      pooled_size, part_size, sample_per_part);
  if (rc != OVIS_OK) return rc;
  const long total = (long)num_rois * output_dim * pooled_size * pooled_size;
  if (total == 0) return OVIS_OK;
  if (!grad_out || !out_count || !data || !rois || !grad_data || (!no_trans && (!trans || !grad_trans))) return OVIS_EINVAL;
  const int num_classes = no_trans ? 1 : channels_trans / 2;
  const int cec = no_trans ? output_dim : output_dim / num_classes;
  const unsigned blocks = (unsigned)min((total + 255) / 256, (long)OVIS_NUM_CU * 32);
  hipLaunchKernelGGL(deform_psroi_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grad_out, out_count, data,
                     rois, trans, grad_data, grad_trans, total, spatial_scale, channels, height, width, pooled_size,
                     no_trans ? 1 : 0, trans_std, sample_per_part, output_dim, group_size, part_size, num_classes, cec);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
