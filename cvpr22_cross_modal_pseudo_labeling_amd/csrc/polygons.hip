// Polygon ground-truth masks -> mask-head targets on the device, gfx950 (MI355X).
//
// Reference path: maskrcnn_benchmark/modeling/roi_heads/mask_head/loss.py:11-42 (project_masks_on_boxes) on
// SegmentationMask(mode='poly') targets: per positive proposal
//     PolygonInstance.crop(box)      structures/segmentation_mask.py:270-296   clamp the box (python floats), p -= (xmin, ymin)
//     PolygonInstance.resize((M,M))  :298-324                                  p *= M / w, M / h   (float32 tensors)
//     convert_to_binarymask()        :326-334    pycocotools.mask.frPyObjects -> merge -> decode at M x M
// The rasteriser lives in a third-party dependency that is NOT under /root/reference: pycocotools==2.0
// (requirements.txt:35), common/maskApi.c -- rleFrPoly (polygon -> RLE), rleMerge (union), rleDecode.  Its published
// algorithm is restated here (and in oracle/ovis_oracle.c, which this kernel is tested against bit for bit):
//   1. vertices are scaled by 5 and rounded ((int)(5 * v + .5), truncation) -- double arithmetic on the float32 inputs;
//   2. every edge is walked densely along its major axis: d = 0 .. max(|dx|, |dy|), the minor coordinate
//      (int)(start + slope * t + .5), points emitted from the edge's first vertex to its second;
//   3. wherever consecutive points differ in x, a boundary point is emitted at xd = (min-side x + .5) / 5 - .5 if that is
//      an integer column inside [0, w - 1], yd = ceil(clamp((min(y, y_prev) + .5) / 5 - .5, 0, h));
//   4. the boundary points' column-major positions x * h + y, sorted, are the run boundaries of the RLE; equal positions
//      cancel.  Decoded, pixel (x, y) is set iff an ODD number of boundary points lie at positions <= x * h + y.
// Step 4 is evaluated as "toggle a counter per position, then prefix parity", which needs no sort.  Polygons of one
// instance are merged by union (rleMerge, intersect = 0).
//
// One wave per positive proposal; its lanes share the points of an edge; toggles are LDS integer atomics; the prefix
// parity is a wave scan.  Nothing here is bandwidth: a step has <= 512 positives of ~10-100 vertices each.
#include "ovis_common.h"

namespace {

constexpr int kMaxPositions = 64 * 64 + 1;  // resolution up to 64

__device__ __forceinline__ void edge_point(int xs, int ys, int xe, int ye, int d, int* u, int* v) {
  // point number d (0 .. max(dx, dy)) of the edge (xs, ys) -> (xe, ye) in the order rleFrPoly emits them
  const int dx = abs(xe - xs), dy = abs(ys - ye);
  const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
  int a_xs = xs, a_ys = ys, a_xe = xe, a_ye = ye;
  if (flip) { a_xs = xe; a_xe = xs; a_ys = ye; a_ye = ys; }
  if (dx >= dy) {
    const double s = (double)(a_ye - a_ys) / dx;   // dx == 0 only when the edge is a point: d == 0, s unused (0 * inf guarded)
    const int t = flip ? dx - d : d;
    *u = t + a_xs;
    *v = dx == 0 ? a_ys : (int)(a_ys + s * t + .5);
  } else {
    const double s = (double)(a_xe - a_xs) / dy;
    const int t = flip ? dy - d : d;
    *v = t + a_ys;
    *u = (int)(a_xs + s * t + .5);
  }
}

__global__ __launch_bounds__(64) void project_polygons_kernel(
    const float* __restrict__ coords, const int* __restrict__ poly_start, const int* __restrict__ inst_start,
    const long* __restrict__ gt_index, const float* __restrict__ boxes, int P, int img_w, int img_h, int M,
    float* __restrict__ out) {
  __shared__ int toggles[kMaxPositions];
  __shared__ unsigned char acc[kMaxPositions];
  const int p = blockIdx.x;
  const int lane = threadIdx.x;
  const int npos = M * M;
  for (int i = lane; i < npos; i += 64) acc[i] = 0;
  // PolygonInstance.crop: python-float (double) clamps of the float32 box
  const float4 bb = *(const float4*)(boxes + 4 * (long)p);
  double xmin = bb.x, ymin = bb.y, xmax = bb.z, ymax = bb.w;
  xmin = fmin(fmax(xmin, 0.0), (double)(img_w - 1));
  ymin = fmin(fmax(ymin, 0.0), (double)(img_h - 1));
  xmax = fmin(fmax(xmax, 0.0), (double)img_w);
  ymax = fmin(fmax(ymax, 0.0), (double)img_h);
  xmax = fmax(xmax, xmin + 1.0);
  ymax = fmax(ymax, ymin + 1.0);
  const double w = xmax - xmin, h = ymax - ymin;
  // tensor (float32) - python scalar / * python scalar: the scalar is rounded to float32, the operation is float32
  const float fxmin = (float)xmin, fymin = (float)ymin;
  const float rw = (float)((double)M / w), rh = (float)((double)M / h);
  const long g = gt_index[p];
  for (int poly = inst_start[g]; poly < inst_start[g + 1]; ++poly) {
    const int c0 = poly_start[poly];
    const int k = (poly_start[poly + 1] - c0) / 2;   // vertices
    if (k < 3) continue;                              // PolygonInstance.__init__ drops polygons with < 6 numbers
    for (int i = lane; i <= npos; i += 64) toggles[i] = 0;
    __syncthreads();
    for (int j = 0; j < k; ++j) {
      const int jn = j + 1 == k ? 0 : j + 1;
      // crop + resize in float32, then rleFrPoly's scale-by-5 rounding in double
      const float ax = (coords[c0 + 2 * j] - fxmin) * rw, ay = (coords[c0 + 2 * j + 1] - fymin) * rh;
      const float bx = (coords[c0 + 2 * jn] - fxmin) * rw, by = (coords[c0 + 2 * jn + 1] - fymin) * rh;
      const int xs = (int)(5.0 * (double)ax + .5), ys = (int)(5.0 * (double)ay + .5);
      const int xe = (int)(5.0 * (double)bx + .5), ye = (int)(5.0 * (double)by + .5);
      const int steps = max(abs(xe - xs), abs(ye - ys));
      // consecutive points (d - 1, d), d = 1 .. steps; the pair across two edges is the shared vertex twice: no boundary
      for (int d = 1 + lane; d <= steps; d += 64) {
        int u0, v0, u1, v1;
        edge_point(xs, ys, xe, ye, d - 1, &u0, &v0);
        edge_point(xs, ys, xe, ye, d, &u1, &v1);
        if (u1 == u0) continue;
        double xd = (double)(u1 < u0 ? u1 : u1 - 1);
        xd = (xd + .5) / 5.0 - .5;
        if (floor(xd) != xd || xd < 0 || xd > M - 1) continue;
        double yd = (double)(v1 < v0 ? v1 : v0);
        yd = (yd + .5) / 5.0 - .5;
        if (yd < 0) yd = 0; else if (yd > M) yd = M;
        yd = ceil(yd);
        atomicAdd(&toggles[(int)xd * M + (int)yd], 1);
      }
    }
    __syncthreads();
    // prefix parity over the column-major positions, OR-ed into the instance's mask
    int carry = 0;
    for (int base = 0; base < npos; base += 64) {
      const int i = base + lane;
      int v = i < npos ? (toggles[i] & 1) : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      const int par = (carry + incl) & 1;
      if (i < npos && par) acc[i] = 1;
      carry = (carry + __shfl(incl, 63, 64)) & 1;
    }
    __syncthreads();
  }
  __syncthreads();
  // out[p][y][x]; position = x * M + y (column-major, as the RLE counts)
  float* o = out + (long)p * npos;
  for (int i = lane; i < npos; i += 64) {
    const int y = i / M, x = i - y * M;
    o[i] = (float)acc[x * M + y];
  }
}

}  // namespace

extern "C" int ovis_project_polygon_masks_f32(const float* coords, const int32_t* polygon_start,
                                              const int32_t* instance_start, const int64_t* gt_index, const float* boxes,
                                              int num, int image_width, int image_height, int resolution, float* out,
                                              void* stream) {
  if (num < 0 || resolution <= 0 || image_width <= 0 || image_height <= 0) return OVIS_EINVAL;
  if (num == 0) return OVIS_OK;
  if (!polygon_start || !instance_start || !gt_index || !boxes || !out) return OVIS_EINVAL;
  if (resolution > 64) return OVIS_ERANGE;
  if (((uintptr_t)boxes & 15) != 0) return OVIS_EINVAL;
  hipLaunchKernelGGL(project_polygons_kernel, dim3((unsigned)num), dim3(64), 0, (hipStream_t)stream, coords, polygon_start,
                     instance_start, (const long*)gt_index, boxes, num, image_width, image_height, resolution, out);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
