/*
 * ovis_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's native hot ops.  It is the
 * checker for the HIP kernels (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
 * leg); nothing in the product package may import, link or call it.
 *
 * Each function cites the reference source it follows (paths relative to the reference
 * repo, maskrcnn_benchmark/csrc/...).  Pinning status (see DESIGN.md "Oracle"):
 *   - roi_align_forward, nms       : pinned against the reference's own CPU kernels
 *                                    (oracle/_ref build + tests/golden fixtures).
 *   - sigmoid focal forward        : pinned against the reference's Python formula
 *                                    layers/sigmoid_focal_loss.py:40-50 (fixtures); its
 *                                    backward against autograd of that formula.
 *   - roi_align_backward           : the reference has NO CPU implementation
 *                                    (csrc/ROIAlign.h:44) and no tests -> "parity
 *                                    unpinned" by reference vectors; pinned instead by the
 *                                    adjoint identity <fwd(x), g> == <x, bwd(g)> in fp64.
 *
 * Build: `make -C oracle` -> oracle/libovis_oracle.so (gcc -O2, no -ffast-math, no FMA
 * contraction so fp32 results follow the reference's operation order exactly).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* RoIAlign. cuda/ROIAlign_cuda.cu:16-62 (bilinear_interpolate), :65-122 (forward),       */
/* :125-175 (gradient weights), :178-254 (backward); cpu/ROIAlign_cpu.cpp:18-219.         */
/* ------------------------------------------------------------------------------------ */

#define DEFINE_ROI_ALIGN(T, SUFFIX, CEIL)                                                    \
  static int axis_##SUFFIX(T v, int size, int* lo, int* hi, T* l, T* h) {                    \
    if (v < (T)-1.0 || v > (T)size) return 0;                                                \
    if (v <= 0) v = 0;                                                                       \
    *lo = (int)v;                                                                            \
    if (*lo >= size - 1) {                                                                   \
      *hi = *lo = size - 1;                                                                  \
      v = (T)*lo;                                                                            \
    } else {                                                                                 \
      *hi = *lo + 1;                                                                         \
    }                                                                                        \
    *l = v - (T)*lo;                                                                         \
    *h = (T)1. - *l;                                                                         \
    return 1;                                                                                \
  }                                                                                          \
                                                                                             \
  void oracle_roi_align_forward_##SUFFIX(const T* in, const T* rois, T* out, int R, int N,   \
                                         int C, int H, int W, int PH, int PW, T scale,       \
                                         int sampling_ratio) {                               \
    (void)N;                                                                                 \
    for (int n = 0; n < R; ++n) {                                                            \
      const T* roi = rois + (size_t)n * 5;                                                   \
      int b = (int)roi[0];                                                                   \
      T sw = roi[1] * scale, sh = roi[2] * scale, ew = roi[3] * scale, eh = roi[4] * scale;  \
      T rw = ew - sw > (T)1. ? ew - sw : (T)1.;                                              \
      T rh = eh - sh > (T)1. ? eh - sh : (T)1.;                                              \
      T bh = rh / (T)PH, bw = rw / (T)PW;                                                    \
      int gh = sampling_ratio > 0 ? sampling_ratio : (int)CEIL(rh / PH);                     \
      int gw = sampling_ratio > 0 ? sampling_ratio : (int)CEIL(rw / PW);                     \
      T count = (T)(gh * gw);                                                                \
      for (int c = 0; c < C; ++c) {                                                          \
        const T* plane = in + ((size_t)b * C + c) * H * W;                                   \
        T* o = out + ((size_t)n * C + c) * PH * PW;                                          \
        for (int ph = 0; ph < PH; ++ph)                                                      \
          for (int pw = 0; pw < PW; ++pw) {                                                  \
            T acc = 0;                                                                       \
            for (int iy = 0; iy < gh; ++iy) {                                                \
              T y = sh + ph * bh + (T)(iy + .5f) * bh / (T)gh;                               \
              int yl, yh;                                                                    \
              T ly, hy;                                                                      \
              if (!axis_##SUFFIX(y, H, &yl, &yh, &ly, &hy)) continue;                        \
              for (int ix = 0; ix < gw; ++ix) {                                              \
                T x = sw + pw * bw + (T)(ix + .5f) * bw / (T)gw;                             \
                int xl, xh;                                                                  \
                T lx, hx;                                                                    \
                if (!axis_##SUFFIX(x, W, &xl, &xh, &lx, &hx)) continue;                      \
                T w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;                    \
                acc += w1 * plane[yl * W + xl] + w2 * plane[yl * W + xh] +                   \
                       w3 * plane[yh * W + xl] + w4 * plane[yh * W + xh];                    \
              }                                                                              \
            }                                                                                \
            o[ph * PW + pw] = acc / count;                                                   \
          }                                                                                  \
      }                                                                                      \
    }                                                                                        \
  }                                                                                          \
                                                                                             \
  void oracle_roi_align_backward_##SUFFIX(const T* gout, const T* rois, T* gin, int R,       \
                                          int N, int C, int H, int W, int PH, int PW,        \
                                          T scale, int sampling_ratio) {                     \
    memset(gin, 0, sizeof(T) * (size_t)N * C * H * W);                                       \
    for (int n = 0; n < R; ++n) {                                                            \
      const T* roi = rois + (size_t)n * 5;                                                   \
      int b = (int)roi[0];                                                                   \
      T sw = roi[1] * scale, sh = roi[2] * scale, ew = roi[3] * scale, eh = roi[4] * scale;  \
      T rw = ew - sw > (T)1. ? ew - sw : (T)1.;                                              \
      T rh = eh - sh > (T)1. ? eh - sh : (T)1.;                                              \
      T bh = rh / (T)PH, bw = rw / (T)PW;                                                    \
      int gh = sampling_ratio > 0 ? sampling_ratio : (int)CEIL(rh / PH);                     \
      int gw = sampling_ratio > 0 ? sampling_ratio : (int)CEIL(rw / PW);                     \
      T count = (T)(gh * gw);                                                                \
      for (int c = 0; c < C; ++c) {                                                          \
        T* plane = gin + ((size_t)b * C + c) * H * W;                                        \
        const T* g = gout + ((size_t)n * C + c) * PH * PW;                                   \
        for (int ph = 0; ph < PH; ++ph)                                                      \
          for (int pw = 0; pw < PW; ++pw) {                                                  \
            T top = g[ph * PW + pw];                                                         \
            for (int iy = 0; iy < gh; ++iy) {                                                \
              T y = sh + ph * bh + (T)(iy + .5f) * bh / (T)gh;                               \
              int yl, yh;                                                                    \
              T ly, hy;                                                                      \
              if (!axis_##SUFFIX(y, H, &yl, &yh, &ly, &hy)) continue;                        \
              for (int ix = 0; ix < gw; ++ix) {                                              \
                T x = sw + pw * bw + (T)(ix + .5f) * bw / (T)gw;                             \
                int xl, xh;                                                                  \
                T lx, hx;                                                                    \
                if (!axis_##SUFFIX(x, W, &xl, &xh, &lx, &hx)) continue;                      \
                plane[yl * W + xl] += top * (hy * hx) / count;                               \
                plane[yl * W + xh] += top * (hy * lx) / count;                               \
                plane[yh * W + xl] += top * (ly * hx) / count;                               \
                plane[yh * W + xh] += top * (ly * lx) / count;                               \
              }                                                                              \
            }                                                                                \
          }                                                                                  \
      }                                                                                      \
    }                                                                                        \
  }

DEFINE_ROI_ALIGN(float, f32, ceilf)
DEFINE_ROI_ALIGN(double, f64, ceil)

/* ------------------------------------------------------------------------------------ */
/* NMS. cuda/nms.cu:13-21 (devIoU), :23-67 (mask `>`), :106-130 (greedy reduce, ascending */
/* index output); cpu/nms_cpu.cpp:37-66 (`>=`).  ge_mode selects the comparison.          */
/* Returns the number of survivors; keep_out receives ascending original indices.         */
/* ------------------------------------------------------------------------------------ */

static void stable_order_desc(const float* s, int n, int* order) {
  /* bottom-up merge sort on indices: descending score, ties keep the lower index first */
  int* tmp = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int width = 1; width < n; width *= 2) {
    for (int lo = 0; lo < n; lo += 2 * width) {
      int mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
      int a = lo, b = mid, k = lo;
      while (a < mid && b < hi) tmp[k++] = (s[order[b]] > s[order[a]]) ? order[b++] : order[a++];
      while (a < mid) tmp[k++] = order[a++];
      while (b < hi) tmp[k++] = order[b++];
    }
    memcpy(order, tmp, sizeof(int) * (size_t)n);
  }
  free(tmp);
}

int oracle_nms_f32(const float* boxes, const float* scores, int K, float thr, int ge_mode,
                   int64_t* keep_out) {
  if (K <= 0) return 0;
  int* order = (int*)malloc(sizeof(int) * (size_t)K);
  unsigned char* dead = (unsigned char*)calloc((size_t)K, 1);
  stable_order_desc(scores, K, order);
  for (int _i = 0; _i < K; ++_i) {
    int i = order[_i];
    if (dead[i]) continue;
    const float* a = boxes + (size_t)i * 4;
    float sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    for (int _j = _i + 1; _j < K; ++_j) {
      int j = order[_j];
      if (dead[j]) continue;
      const float* b = boxes + (size_t)j * 4;
      float left = a[0] > b[0] ? a[0] : b[0], right = a[2] < b[2] ? a[2] : b[2];
      float top = a[1] > b[1] ? a[1] : b[1], bottom = a[3] < b[3] ? a[3] : b[3];
      float w = right - left + 1 > 0.f ? right - left + 1 : 0.f;
      float h = bottom - top + 1 > 0.f ? bottom - top + 1 : 0.f;
      float inter = w * h;
      float sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
      float ovr = inter / (sa + sb - inter);
      if (ge_mode ? (ovr >= thr) : (ovr > thr)) dead[j] = 1;
    }
  }
  int n = 0;
  for (int i = 0; i < K; ++i)
    if (!dead[i]) keep_out[n++] = i;
  free(order);
  free(dead);
  return n;
}

/* ------------------------------------------------------------------------------------ */
/* Sigmoid focal loss. cuda/SigmoidFocalLoss_cuda.cu:21-58 (forward), :62-101 (backward). */
/* ------------------------------------------------------------------------------------ */

void oracle_sigmoid_focal_loss_forward_f32(const float* logits, const int32_t* targets,
                                           float* losses, int num, int C, float gamma,
                                           float alpha) {
  for (long i = 0; i < (long)num * C; ++i) {
    int n = (int)(i / C), d = (int)(i % C), t = targets[n];
    float c1 = (t == d + 1), c2 = (t >= 0 && t != d + 1);
    float zn = 1.0f - alpha, zp = alpha, x = logits[i];
    float p = 1.f / (1.f + expf(-x));
    float term1 = powf(1.f - p, gamma) * logf(p > FLT_MIN ? p : FLT_MIN);
    float pos = x >= 0;
    float term2 = powf(p, gamma) * (-1.f * x * pos - logf(1.f + expf(x - 2.f * x * pos)));
    float l = 0.f;
    l += -c1 * term1 * zp;
    l += -c2 * term2 * zn;
    losses[i] = l;
  }
}

void oracle_sigmoid_focal_loss_backward_f32(const float* logits, const int32_t* targets,
                                            const float* d_losses, float* d_logits, int num,
                                            int C, float gamma, float alpha) {
  for (long i = 0; i < (long)num * C; ++i) {
    int n = (int)(i / C), d = (int)(i % C), t = targets[n];
    float c1 = (t == d + 1), c2 = (t >= 0 && t != d + 1);
    float zn = 1.0f - alpha, zp = alpha, x = logits[i];
    float p = 1.f / (1.f + expf(-x));
    float term1 = powf(1.f - p, gamma) * (1.f - p - (p * gamma * logf(p > FLT_MIN ? p : FLT_MIN)));
    float pos = x >= 0;
    float term2 = powf(p, gamma) *
                  ((-1.f * x * pos - logf(1.f + expf(x - 2.f * x * pos))) * (1.f - p) * gamma - p);
    float g = 0.f;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    d_logits[i] = g * d_losses[i];
  }
}

/* ------------------------------------------------------------------------------------ */
/* ROIPool (max pooling per bin).  cuda/ROIPool_cuda.cu:17-77 (forward: rounded RoI, bin  */
/* extents floor/ceil, clip to the map, empty bin -> 0 / argmax -1, first maximum in      */
/* (h, w) scan order), :80-108 (backward: scatter-add of the bin gradient to its argmax). */
/* The reference has no CPU implementation (csrc/ROIPool.h:20,37) and no tests: "parity    */
/* unpinned" by reference vectors; pinned by an independent tensor-op formulation in      */
/* tests/test_oracle.py.                                                                  */
/* ------------------------------------------------------------------------------------ */

void oracle_roi_pool_forward_f32(const float* in, const float* rois, float* out, int32_t* argmax,
                                 int R, int C, int H, int W, int PH, int PW, float scale) {
  for (long index = 0; index < (long)R * C * PH * PW; ++index) {
    int pw = (int)(index % PW), ph = (int)((index / PW) % PH);
    int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    const float* r = rois + (long)n * 5;
    int b = (int)r[0];
    int sw = (int)roundf(r[1] * scale), sh = (int)roundf(r[2] * scale);
    int ew = (int)roundf(r[3] * scale), eh = (int)roundf(r[4] * scale);
    int rw = ew - sw + 1 > 1 ? ew - sw + 1 : 1, rh = eh - sh + 1 > 1 ? eh - sh + 1 : 1;
    float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    int hs = (int)floorf((float)ph * bh), ws = (int)floorf((float)pw * bw);
    int he = (int)ceilf((float)(ph + 1) * bh), we = (int)ceilf((float)(pw + 1) * bw);
    hs = hs + sh < 0 ? 0 : (hs + sh > H ? H : hs + sh);
    he = he + sh < 0 ? 0 : (he + sh > H ? H : he + sh);
    ws = ws + sw < 0 ? 0 : (ws + sw > W ? W : ws + sw);
    we = we + sw < 0 ? 0 : (we + sw > W ? W : we + sw);
    int empty = (he <= hs) || (we <= ws);
    float maxval = empty ? 0.f : -FLT_MAX;
    int maxidx = -1;
    const float* p = in + ((long)b * C + c) * H * W;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w)
        if (p[h * W + w] > maxval) { maxval = p[h * W + w]; maxidx = h * W + w; }
    out[index] = maxval;
    argmax[index] = maxidx;
  }
}

void oracle_roi_pool_backward_f32(const float* grad, const int32_t* argmax, const float* rois,
                                  float* gin, int R, int N, int C, int H, int W, int PH, int PW) {
  for (long i = 0; i < (long)N * C * H * W; ++i) gin[i] = 0.f;
  for (long index = 0; index < (long)R * C * PH * PW; ++index) {
    int c = (int)((index / PW / PH) % C), n = (int)(index / PW / PH / C);
    int b = (int)rois[(long)n * 5];
    if (argmax[index] != -1) gin[((long)b * C + c) * H * W + argmax[index]] += grad[index];
  }
}

/* ------------------------------------------------------------------------------------ */
/* Polygon -> binary mask.  THIRD-PARTY ALGORITHM, PARITY UNPINNED: the reference calls  */
/* pycocotools.mask.frPyObjects / merge / decode (structures/segmentation_mask.py:326-334, */
/* pycocotools==2.0 per requirements.txt:35), which is not under /root/reference and not   */
/* installed here.  This restates the published algorithm of its common/maskApi.c:         */
/* rleFrPoly (polygon -> run lengths: x5 upsampling, dense edge walk, y-boundary points at  */
/* integer pixel columns, sort, zero-run merging), rleMerge with intersect = 0 (union) and   */
/* rleDecode (column-major runs, starting with a 0-run) -- literally, sort and runs         */
/* included, so that the device kernel's sort-free "toggle + prefix parity" formulation is  */
/* checked against the original control flow.  Anchors: the reference's call site above and */
/* known-answer cases in tests/test_polygons.py (rectangles, triangles, out-of-image        */
/* vertices) worked out by hand from the algorithm's definition.                             */
/* ------------------------------------------------------------------------------------ */
static int cmp_uint(const void* a, const void* b) {
  uint32_t c = *(const uint32_t*)a, d = *(const uint32_t*)b;
  return c > d ? 1 : (c < d ? -1 : 0);
}

/* xy: k vertices (x0, y0, x1, y1, ...) in double; mask [h][w] row-major uint8, OR-ed in (union of polygons). */
void oracle_polygon_to_mask(const double* xy, int k, int h, int w, uint8_t* mask) {
  const double scale = 5;
  int j, m = 0;
  if (k < 1) return;
  int* x = (int*)malloc(sizeof(int) * (k + 1));
  int* y = (int*)malloc(sizeof(int) * (k + 1));
  for (j = 0; j < k; j++) x[j] = (int)(scale * xy[j * 2 + 0] + .5);
  x[k] = x[0];
  for (j = 0; j < k; j++) y[j] = (int)(scale * xy[j * 2 + 1] + .5);
  y[k] = y[0];
  for (j = 0; j < k; j++) {
    int a = abs(x[j] - x[j + 1]), b = abs(y[j] - y[j + 1]);
    m += (a > b ? a : b) + 1;
  }
  int* u = (int*)malloc(sizeof(int) * m);
  int* v = (int*)malloc(sizeof(int) * m);
  m = 0;
  for (j = 0; j < k; j++) {
    int xs = x[j], xe = x[j + 1], ys = y[j], ye = y[j + 1], dx, dy, t, d, flip;
    double s;
    dx = abs(xe - xs);
    dy = abs(ys - ye);
    flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) { t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
    s = dx >= dy ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;
    if (dx >= dy) for (d = 0; d <= dx; d++) {
      t = flip ? dx - d : d;
      u[m] = t + xs;
      /* dx == 0 (a repeated vertex): the original evaluates (int)(NaN); that point never meets a neighbour with a
         different x, so its y is never read -- any value will do */
      v[m] = dx == 0 ? ys : (int)(ys + s * t + .5);
      m++;
    } else for (d = 0; d <= dy; d++) {
      t = flip ? dy - d : d;
      v[m] = t + ys;
      u[m] = (int)(xs + s * t + .5);
      m++;
    }
  }
  /* points along the y-boundary, downsampled */
  int n = m;
  free(x); free(y);
  x = (int*)malloc(sizeof(int) * (n + 1));
  y = (int*)malloc(sizeof(int) * (n + 1));
  m = 0;
  for (j = 1; j < n; j++) if (u[j] != u[j - 1]) {
    double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1), yd;
    xd = (xd + .5) / scale - .5;
    if (floor(xd) != xd || xd < 0 || xd > w - 1) continue;
    yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]);
    yd = (yd + .5) / scale - .5;
    if (yd < 0) yd = 0; else if (yd > h) yd = h;
    yd = ceil(yd);
    x[m] = (int)xd;
    y[m] = (int)yd;
    m++;
  }
  /* run lengths from the sorted boundary positions */
  int cnt = m;
  uint32_t* a = (uint32_t*)malloc(sizeof(uint32_t) * (cnt + 1));
  for (j = 0; j < cnt; j++) a[j] = (uint32_t)(x[j] * (int)h + y[j]);
  a[cnt++] = (uint32_t)(h * w);
  free(u); free(v); free(x); free(y);
  qsort(a, cnt, sizeof(uint32_t), cmp_uint);
  uint32_t p = 0;
  for (j = 0; j < cnt; j++) { uint32_t t = a[j]; a[j] -= p; p = t; }
  uint32_t* b = (uint32_t*)malloc(sizeof(uint32_t) * cnt);
  j = 0; m = 0;
  b[m++] = a[j++];
  while (j < cnt) if (a[j] > 0) b[m++] = a[j++]; else { j++; if (j < cnt) b[m - 1] += a[j++]; }
  /* rleDecode (column-major) OR-ed into the row-major mask: rleMerge(intersect = 0) of the instance's polygons */
  long pos = 0;
  uint8_t val = 0;
  for (j = 0; j < m; j++) {
    for (uint32_t c = 0; c < b[j] && pos < (long)h * w; c++, pos++)
      if (val) mask[(pos % h) * w + (pos / h)] = 1;
    val = !val;
  }
  free(a); free(b);
}
