// RoI geometry shared by the RoIAlign forward / backward translation units (gfx950).
// Follows the reference's bilinear set-up, maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:16-62,86-113.
#pragma once
#include <hip/hip_runtime.h>

namespace ovis_roi {

struct RoiGeom {
  int b;
  float start_w, start_h, bin_w, bin_h;
  int gh, gw;
  float count;
  // When the sampling grid is a power of two per axis (every RoI up to 448 px at 14x14 / scale 1/16),
  // x / g == x * (1/g) and acc / count == acc * (1/count) EXACTLY, so the IEEE divisions of the reference
  // formula (10+ VALU each) can be replaced by multiplies without changing a single bit.
  bool pow2;
  float inv_gh, inv_gw, inv_count;
  int wy0, wy1, wx0, wx1;  // inclusive feature-map window touched by valid samples
  bool empty;
};

// One axis of the reference's bilinear set-up (ROIAlign_cuda.cu:22-50): returns false for a
// coordinate outside [-1, size]; otherwise low/high cell and the two lerp weights.
static __device__ __forceinline__ bool axis_sample(float v, int size, int& lo, int& hi, float& l,
                                            float& h) {
  if (v < -1.0f || v > (float)size) return false;
  if (v <= 0.f) v = 0.f;
  lo = (int)v;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    v = (float)lo;
  } else {
    hi = lo + 1;
  }
  l = v - (float)lo;
  h = 1.f - l;
  return true;
}

static __device__ __forceinline__ int axis_low(float v, int size) {
  if (v <= 0.f) return 0;
  int lo = (int)v;
  return lo >= size - 1 ? size - 1 : lo;
}

static __device__ __forceinline__ float sample_coord(float start, int p, float bin, int i, int g) {
  // ROIAlign_cuda.cu:109,112: start + p*bin + (i + .5f)*bin / g
  return start + (float)p * bin + ((float)i + .5f) * bin / (float)g;
}

template <bool FAST>
static __device__ __forceinline__ float sample_coord_t(float start, int p, float bin, int i, int g, float inv_g) {
  const float t = ((float)i + .5f) * bin;
  return start + (float)p * bin + (FAST ? t * inv_g : t / (float)g);
}

static __device__ __forceinline__ RoiGeom make_geom(const float* __restrict__ roi, float scale, int H,
                                             int W, int PH, int PW, int sampling_ratio,
                                             int batch) {
  RoiGeom g;
  g.b = (int)roi[0];
  g.start_w = roi[1] * scale;
  g.start_h = roi[2] * scale;
  float end_w = roi[3] * scale;
  float end_h = roi[4] * scale;
  float roi_w = fmaxf(end_w - g.start_w, 1.f);
  float roi_h = fmaxf(end_h - g.start_h, 1.f);
  g.bin_h = roi_h / (float)PH;
  g.bin_w = roi_w / (float)PW;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)PH);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)PW);
  g.count = (float)(g.gh * g.gw);
  g.pow2 = g.gh > 0 && g.gw > 0 && (g.gh & (g.gh - 1)) == 0 && (g.gw & (g.gw - 1)) == 0 && g.gh <= 1024 &&
           g.gw <= 1024;
  g.inv_gh = 1.f / (float)g.gh;
  g.inv_gw = 1.f / (float)g.gw;
  g.inv_count = 1.f / g.count;
  // Sample coordinates are monotone in (p, i), so the first / last sample bound them all.
  float y_first = sample_coord(g.start_h, 0, g.bin_h, 0, g.gh);
  float y_last = sample_coord(g.start_h, PH - 1, g.bin_h, g.gh - 1, g.gh);
  float x_first = sample_coord(g.start_w, 0, g.bin_w, 0, g.gw);
  float x_last = sample_coord(g.start_w, PW - 1, g.bin_w, g.gw - 1, g.gw);
  bool finite = isfinite(y_first) && isfinite(y_last) && isfinite(x_first) && isfinite(x_last);
  g.empty = !finite || g.b < 0 || g.b >= batch || g.gh <= 0 || g.gw <= 0 ||
            y_last < -1.f || y_first > (float)H || x_last < -1.f || x_first > (float)W;
  if (!g.empty) {
    g.wy0 = axis_low(fmaxf(y_first, -1.f), H);
    g.wx0 = axis_low(fmaxf(x_first, -1.f), W);
    g.wy1 = min(axis_low(fminf(y_last, (float)H), H) + 1, H - 1);
    g.wx1 = min(axis_low(fminf(x_last, (float)W), W) + 1, W - 1);
  } else {
    g.wy0 = g.wx0 = 0;
    g.wy1 = g.wx1 = -1;
  }
  return g;
}


}  // namespace ovis_roi
