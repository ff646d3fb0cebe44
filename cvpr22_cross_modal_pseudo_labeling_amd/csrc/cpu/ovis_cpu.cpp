// libovis_cpu.so: host twins of the two native ops the reference implements on the CPU (RoIAlign forward, NMS) plus the
// RoIAlign transpose -- see include/ovis_cpu.h for the contract and the reference lines.  Product code for HOST tensors of the
// reference's CPU-only configuration; device tensors never come here.
//
// RoIAlign is evaluated SEPARABLY: the sampling positions of a RoI along y and along x do not depend on each other
// (the reference's joint table of pooled_h * pooled_w * grid_h * grid_w entries is the outer product of two short per-axis
// tables), so a RoI builds pooled_h * grid_h + pooled_w * grid_w axis samples once and every channel re-uses them.  The
// floating-point expressions per output are the reference's, in its order.
#include "ovis_cpu.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

struct AxisSample {  // one sampling coordinate along one axis
  int lo, hi;        // the two cells it interpolates between
  float frac, rest;  // frac = distance from cell lo (the weight of cell hi), rest = 1 - frac (the weight of cell lo)
  bool inside;       // false: the coordinate lies outside [-1, size] and contributes nothing
};

AxisSample axis_sample(float v, int size) {
  AxisSample s{0, 0, 0.f, 0.f, false};
  if (v < -1.0f || v > (float)size) return s;
  if (v <= 0.f) v = 0.f;
  s.lo = (int)v;
  if (s.lo >= size - 1) {
    s.hi = s.lo = size - 1;
    v = (float)s.lo;
  } else {
    s.hi = s.lo + 1;
  }
  s.frac = v - (float)s.lo;
  s.rest = 1.f - s.frac;
  s.inside = true;
  return s;
}

struct RoiGeometry {
  int image;
  int grid_h, grid_w;
  float count;
  std::vector<AxisSample> ys, xs;  // [pooled * grid] per axis, bin-major
};

RoiGeometry roi_geometry(const float* roi, int height, int width, int pooled_h, int pooled_w, float scale, int sampling_ratio) {
  RoiGeometry g;
  g.image = (int)roi[0];
  const float start_w = roi[1] * scale, start_h = roi[2] * scale, end_w = roi[3] * scale, end_h = roi[4] * scale;
  const float roi_w = std::max(end_w - start_w, 1.f), roi_h = std::max(end_h - start_h, 1.f);
  const float bin_h = roi_h / (float)pooled_h, bin_w = roi_w / (float)pooled_w;
  g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)std::ceil(roi_h / pooled_h);
  g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)std::ceil(roi_w / pooled_w);
  g.count = (float)(g.grid_h * g.grid_w);
  g.ys.resize((size_t)pooled_h * g.grid_h);
  g.xs.resize((size_t)pooled_w * g.grid_w);
  for (int ph = 0; ph < pooled_h; ++ph)
    for (int iy = 0; iy < g.grid_h; ++iy)
      g.ys[(size_t)ph * g.grid_h + iy] = axis_sample(start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)g.grid_h, height);
  for (int pw = 0; pw < pooled_w; ++pw)
    for (int ix = 0; ix < g.grid_w; ++ix)
      g.xs[(size_t)pw * g.grid_w + ix] = axis_sample(start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)g.grid_w, width);
  return g;
}

int thread_count(int threads) {
#ifdef _OPENMP
  return threads > 0 ? threads : omp_get_max_threads();
#else
  (void)threads;
  return 1;
#endif
}

}  // namespace

extern "C" int ovis_cpu_roi_align_forward_f32(const float* input, const float* rois, float* out, int num_rois, int batch,
                                              int channels, int height, int width, int pooled_h, int pooled_w,
                                              float spatial_scale, int sampling_ratio, int threads) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0) return OVIS_CPU_EINVAL;
  if (num_rois == 0 || channels == 0) return OVIS_CPU_OK;
  if (!input || !rois || !out) return OVIS_CPU_EINVAL;
  for (int r = 0; r < num_rois; ++r) {
    const int b = (int)rois[(size_t)r * 5];
    if (b < 0 || b >= batch) return OVIS_CPU_EINVAL;
  }
  const size_t plane = (size_t)height * width, bins = (size_t)pooled_h * pooled_w;
  const int nt = thread_count(threads);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
  for (int r = 0; r < num_rois; ++r) {
    const RoiGeometry g = roi_geometry(rois + (size_t)r * 5, height, width, pooled_h, pooled_w, spatial_scale, sampling_ratio);
    for (int c = 0; c < channels; ++c) {
      const float* src = input + ((size_t)g.image * channels + c) * plane;
      float* dst = out + ((size_t)r * channels + c) * bins;
      for (int ph = 0; ph < pooled_h; ++ph) {
        for (int pw = 0; pw < pooled_w; ++pw) {
          float acc = 0.f;
          for (int iy = 0; iy < g.grid_h; ++iy) {
            const AxisSample& y = g.ys[(size_t)ph * g.grid_h + iy];
            if (!y.inside) continue;
            const float* row_lo = src + (size_t)y.lo * width;
            const float* row_hi = src + (size_t)y.hi * width;
            for (int ix = 0; ix < g.grid_w; ++ix) {
              const AxisSample& x = g.xs[(size_t)pw * g.grid_w + ix];
              if (!x.inside) continue;
              const float w1 = y.rest * x.rest, w2 = y.rest * x.frac, w3 = y.frac * x.rest, w4 = y.frac * x.frac;
              acc += w1 * row_lo[x.lo] + w2 * row_lo[x.hi] + w3 * row_hi[x.lo] + w4 * row_hi[x.hi];
            }
          }
          dst[(size_t)ph * pooled_w + pw] = acc / g.count;
        }
      }
    }
  }
  return OVIS_CPU_OK;
}

extern "C" int ovis_cpu_roi_align_backward_f32(const float* grad_out, const float* rois, float* grad_input, int num_rois,
                                               int batch, int channels, int height, int width, int pooled_h, int pooled_w,
                                               float spatial_scale, int sampling_ratio, int threads) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0) return OVIS_CPU_EINVAL;
  const size_t plane = (size_t)height * width, bins = (size_t)pooled_h * pooled_w;
  if ((size_t)batch * channels > 0) {
    if (!grad_input) return OVIS_CPU_EINVAL;
    std::memset(grad_input, 0, sizeof(float) * (size_t)batch * channels * plane);
  }
  if (num_rois == 0 || channels == 0) return OVIS_CPU_OK;
  if (!grad_out || !rois) return OVIS_CPU_EINVAL;
  std::vector<RoiGeometry> geo;
  geo.reserve(num_rois);
  for (int r = 0; r < num_rois; ++r) {
    geo.push_back(roi_geometry(rois + (size_t)r * 5, height, width, pooled_h, pooled_w, spatial_scale, sampling_ratio));
    if (geo.back().image < 0 || geo.back().image >= batch) return OVIS_CPU_EINVAL;
  }
  const int nt = thread_count(threads);
  // a thread owns channel c of every image: no two threads touch the same plane, RoIs are visited in order
#pragma omp parallel for schedule(static) num_threads(nt)
  for (int c = 0; c < channels; ++c) {
    for (int r = 0; r < num_rois; ++r) {
      const RoiGeometry& g = geo[r];
      float* dst = grad_input + ((size_t)g.image * channels + c) * plane;
      const float* src = grad_out + ((size_t)r * channels + c) * bins;
      for (int ph = 0; ph < pooled_h; ++ph) {
        for (int pw = 0; pw < pooled_w; ++pw) {
          const float top = src[(size_t)ph * pooled_w + pw];
          for (int iy = 0; iy < g.grid_h; ++iy) {
            const AxisSample& y = g.ys[(size_t)ph * g.grid_h + iy];
            if (!y.inside) continue;
            float* row_lo = dst + (size_t)y.lo * width;
            float* row_hi = dst + (size_t)y.hi * width;
            for (int ix = 0; ix < g.grid_w; ++ix) {
              const AxisSample& x = g.xs[(size_t)pw * g.grid_w + ix];
              if (!x.inside) continue;
              row_lo[x.lo] += top * (y.rest * x.rest) / g.count;
              row_lo[x.hi] += top * (y.rest * x.frac) / g.count;
              row_hi[x.lo] += top * (y.frac * x.rest) / g.count;
              row_hi[x.hi] += top * (y.frac * x.frac) / g.count;
            }
          }
        }
      }
    }
  }
  return OVIS_CPU_OK;
}

extern "C" int ovis_cpu_nms_f32(const float* boxes, const float* scores, int num_boxes, float threshold, int64_t* keep) {
  if (num_boxes < 0) return OVIS_CPU_EINVAL;
  if (num_boxes == 0) return 0;
  if (!boxes || !scores || !keep) return OVIS_CPU_EINVAL;
  std::vector<int> order(num_boxes);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[a] > scores[b]; });
  std::vector<float> area(num_boxes);
  for (int i = 0; i < num_boxes; ++i) {
    const float* q = boxes + (size_t)i * 4;
    area[i] = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
  }
  std::vector<char> gone(num_boxes, 0);
  for (int oi = 0; oi < num_boxes; ++oi) {
    const int i = order[oi];
    if (gone[i]) continue;
    const float* a = boxes + (size_t)i * 4;
    for (int oj = oi + 1; oj < num_boxes; ++oj) {
      const int j = order[oj];
      if (gone[j]) continue;
      const float* b = boxes + (size_t)j * 4;
      const float w = std::max(0.f, std::min(a[2], b[2]) - std::max(a[0], b[0]) + 1);
      const float h = std::max(0.f, std::min(a[3], b[3]) - std::max(a[1], b[1]) + 1);
      const float inter = w * h;
      if (inter / (area[i] + area[j] - inter) >= threshold) gone[j] = 1;
    }
  }
  int n = 0;
  for (int i = 0; i < num_boxes; ++i)
    if (!gone[i]) keep[n++] = i;
  return n;
}

// ---- polygon ground truth -> M x M mask targets (mask_head/loss.py:11-42 on SegmentationMask(mode='poly') targets) ------------
// PolygonInstance.crop (segmentation_mask.py:270-296: python-float clamps of the box) -> resize (:298-324: float32 tensor x
// python scalar) -> convert_to_binarymask (:326-334: pycocotools.mask.frPyObjects -> merge -> decode; pycocotools==2.0,
// common/maskApi.c rleFrPoly / rleMerge / rleDecode -- a third-party dependency outside /root/reference, its published
// algorithm followed here as in csrc/polygons.hip).  Host form: one positive at a time, a dense walk of every edge on the
// x5 grid; a boundary point TOGGLES its column-major position and the decoded mask is the running parity of the toggles
// (equal positions cancel, as equal run boundaries do in the RLE) -- no sort, no run lengths.
static inline void edge_point(int xs, int ys, int xe, int ye, int d, int& u, int& v) {
  const int dx = std::abs(xe - xs), dy = std::abs(ys - ye);
  const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
  if (flip) { std::swap(xs, xe); std::swap(ys, ye); }
  if (dx >= dy) {
    const int t = flip ? dx - d : d;
    u = t + xs;
    v = dx == 0 ? ys : (int)(ys + ((double)(ye - ys) / dx) * t + .5);
  } else {
    const int t = flip ? dy - d : d;
    v = t + ys;
    u = (int)(xs + ((double)(xe - xs) / dy) * t + .5);
  }
}

// Walks the closed polygon whose x5-grid vertices are (xs[j], ys[j]) and toggles the column-major position of every boundary point
// (rleFrPoly steps 2-4) on a w x h grid: toggles has w * h + 1 entries.
static void toggle_boundary(const std::vector<int>& xs, const std::vector<int>& ys, int w, int h, std::vector<int>& toggles) {
  const int k = (int)xs.size();
  for (int j = 0; j < k; ++j) {
    const int jn = j + 1 == k ? 0 : j + 1;
    const int steps = std::max(std::abs(xs[jn] - xs[j]), std::abs(ys[jn] - ys[j]));
    int u0, v0;
    edge_point(xs[j], ys[j], xs[jn], ys[jn], 0, u0, v0);
    for (int d = 1; d <= steps; ++d) {
      int u1, v1;
      edge_point(xs[j], ys[j], xs[jn], ys[jn], d, u1, v1);
      if (u1 != u0) {
        const double xd = ((double)(u1 < u0 ? u1 : u1 - 1) + .5) / 5.0 - .5;
        if (std::floor(xd) == xd && xd >= 0 && xd <= w - 1) {
          double yd = ((double)std::min(v1, v0) + .5) / 5.0 - .5;
          yd = std::ceil(yd < 0 ? 0.0 : (yd > h ? (double)h : yd));
          toggles[(size_t)xd * h + (size_t)yd] ^= 1;
        }
      }
      u0 = u1;
      v0 = v1;
    }
  }
}

// Whole-image masks of polygon instances: SegmentationMask(mode='poly').convert('mask') (mb/structures/segmentation_mask.py:326-334:
// frPyObjects -> merge -> decode at the image size, no crop / resize).  out [num_instances, height, width] uint8.
extern "C" int ovis_cpu_polygons_to_masks_u8(const float* coords, const int32_t* polygon_start, const int32_t* instance_start,
                                             int num_instances, int width, int height, uint8_t* out, int threads) {
  if (num_instances < 0 || width <= 0 || height <= 0) return OVIS_CPU_EINVAL;
  if (num_instances == 0) return OVIS_CPU_OK;
  if (!polygon_start || !instance_start || !out) return OVIS_CPU_EINVAL;
  const size_t npos = (size_t)width * height;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic)
#endif
  for (int g = 0; g < num_instances; ++g) {
    std::vector<int> toggles(npos + 1);
    uint8_t* o = out + (size_t)g * npos;
    std::memset(o, 0, npos);
    for (int poly = instance_start[g]; poly < instance_start[g + 1]; ++poly) {
      const int c0 = polygon_start[poly], k = (polygon_start[poly + 1] - c0) / 2;
      if (k < 3) continue;
      std::fill(toggles.begin(), toggles.end(), 0);
      std::vector<int> xs(k), ys(k);
      for (int j = 0; j < k; ++j) {  // _mask.pyx hands the coordinates over as doubles: float32 -> double is exact
        xs[j] = (int)(5.0 * (double)coords[c0 + 2 * j] + .5);
        ys[j] = (int)(5.0 * (double)coords[c0 + 2 * j + 1] + .5);
      }
      toggle_boundary(xs, ys, width, height, toggles);
      int parity = 0;
      for (int x = 0; x < width; ++x)
        for (int y = 0; y < height; ++y) {
          parity ^= toggles[(size_t)x * height + y];
          if (parity) o[(size_t)y * width + x] = 1;
        }
    }
  }
  return OVIS_CPU_OK;
}

extern "C" int ovis_cpu_project_polygon_masks_f32(const float* coords, const int32_t* polygon_start, const int32_t* instance_start,
                                                  const int64_t* gt_index, const float* boxes, int num, int image_width,
                                                  int image_height, int resolution, float* out, int threads) {
  if (num < 0 || resolution <= 0 || image_width <= 0 || image_height <= 0) return OVIS_CPU_EINVAL;
  if (num == 0) return OVIS_CPU_OK;
  if (!polygon_start || !instance_start || !gt_index || !boxes || !out) return OVIS_CPU_EINVAL;
  const int M = resolution, npos = M * M;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic)
#endif
  for (int p = 0; p < num; ++p) {
    std::vector<int> toggles(npos + 1);
    std::vector<char> acc(npos, 0);
    double xmin = boxes[4 * (size_t)p], ymin = boxes[4 * (size_t)p + 1], xmax = boxes[4 * (size_t)p + 2], ymax = boxes[4 * (size_t)p + 3];
    xmin = std::min(std::max(xmin, 0.0), (double)(image_width - 1));
    ymin = std::min(std::max(ymin, 0.0), (double)(image_height - 1));
    xmax = std::min(std::max(xmax, 0.0), (double)image_width);
    ymax = std::min(std::max(ymax, 0.0), (double)image_height);
    xmax = std::max(xmax, xmin + 1.0);
    ymax = std::max(ymax, ymin + 1.0);
    const float fxmin = (float)xmin, fymin = (float)ymin;
    const float rw = (float)((double)M / (xmax - xmin)), rh = (float)((double)M / (ymax - ymin));
    const int64_t g = gt_index[p];
    for (int poly = instance_start[g]; poly < instance_start[g + 1]; ++poly) {
      const int c0 = polygon_start[poly], k = (polygon_start[poly + 1] - c0) / 2;
      if (k < 3) continue;  // PolygonInstance.__init__ drops polygons with fewer than 6 numbers
      std::fill(toggles.begin(), toggles.end(), 0);
      std::vector<int> xs(k), ys(k);
      for (int j = 0; j < k; ++j) {  // crop + resize in float32, then rleFrPoly's scale-by-5 rounding in double
        const float ax = (coords[c0 + 2 * j] - fxmin) * rw, ay = (coords[c0 + 2 * j + 1] - fymin) * rh;
        xs[j] = (int)(5.0 * (double)ax + .5);
        ys[j] = (int)(5.0 * (double)ay + .5);
      }
      toggle_boundary(xs, ys, M, M, toggles);
      int parity = 0;
      for (int i = 0; i < npos; ++i) {
        parity ^= toggles[i];
        if (parity) acc[i] = 1;  // union over the instance's polygons (rleMerge, intersect = 0)
      }
    }
    float* o = out + (size_t)p * npos;  // out[p][y][x]; positions are column-major (x * M + y), as the RLE counts
    for (int y = 0; y < M; ++y)
      for (int x = 0; x < M; ++x) o[y * M + x] = (float)acc[x * M + y];
  }
  return OVIS_CPU_OK;
}

extern "C" const char* ovis_cpu_version(void) { return "ovis_cpu 3 (RoIAlign fwd/bwd, NMS, polygon masks and mask targets; fp32, OpenMP)"; }
