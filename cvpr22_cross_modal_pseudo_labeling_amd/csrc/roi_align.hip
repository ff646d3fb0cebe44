// RoIAlign forward / backward for gfx950 (MI355X), fp32, NCHW.
//
// Semantics follow the reference kernels
//   forward : maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:16-122  (CPU twin cpu/ROIAlign_cpu.cpp:18-219)
//   backward: maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:125-254
// but the work decomposition is MI355X-first:
//
//  * one workgroup owns (RoI r, a tile of CPB channels). Blocks are ordered channel-tile
//    major so the ~2000 co-resident blocks all read the same few channel planes, which
//    stay in the XCD L2s; HBM sees each feature byte once.
//  * the RoI's bounding window of the feature map is staged into LDS for a batch of up to
//    16 channels with row-contiguous global reads; the four bilinear taps of every sample
//    are LDS gathers, never global gathers.
//  * lane <-> output bin. The sample geometry (tap offsets + weights) is computed once per
//    (lane, sample) and reused across the channel batch held in registers, so the inner
//    loop is 4 ds_read + 8 VALU per (sample, channel).
//  * each output plane (pooled_h*pooled_w floats) is written by consecutive lanes:
//    stores are wave-contiguous; the 0.8 MB/RoI output stream is the HBM roofline term.
//
// The translation unit is compiled with -ffp-contract=off: the forward then performs the
// exact IEEE operation sequence of the reference CPU kernel and is bit-identical to it.
#include <stdlib.h>

#include "ovis_common.h"
#include "roi_geom.h"

namespace {
using namespace ovis_roi;

constexpr int kThreads = 256;
constexpr int kCPB = 32;            // channels per block (forward and backward)
constexpr int kMaxBatch = 16;       // channels staged per LDS batch
constexpr int kFwdLdsFloats = 4352; // 17 KB window budget (>= 50*84 C4 map for 1 channel)

// ---------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------

// Pool NCS channels whose window (origin oy,ox; row stride rs; channel stride cs) starts at
// `src` (LDS window or, for windows too large for LDS, the global plane itself).
template <int NCS>
__device__ __forceinline__ void fwd_pool(const float* src, int cs, int rs, int oy, int ox,
                                         const RoiGeom& g, int H, int W, int PH, int PW,
                                         float* __restrict__ out_c0) {
  const int PHPW = PH * PW;
  for (int bin = threadIdx.x; bin < PHPW; bin += kThreads) {
    const int ph = bin / PW;
    const int pw = bin - ph * PW;
    float acc[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) acc[c] = 0.f;
    for (int iy = 0; iy < g.gh; ++iy) {
      const float y = sample_coord(g.start_h, ph, g.bin_h, iy, g.gh);
      int yl, yh;
      float ly, hy;
      if (!axis_sample(y, H, yl, yh, ly, hy)) continue;
      const int ryl = (yl - oy) * rs, ryh = (yh - oy) * rs;
      for (int ix = 0; ix < g.gw; ++ix) {
        const float x = sample_coord(g.start_w, pw, g.bin_w, ix, g.gw);
        int xl, xh;
        float lx, hx;
        if (!axis_sample(x, W, xl, xh, lx, hx)) continue;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const int o1 = ryl + (xl - ox), o2 = ryl + (xh - ox);
        const int o3 = ryh + (xl - ox), o4 = ryh + (xh - ox);
#pragma unroll
        for (int c = 0; c < NCS; ++c) {
          const float* p = src + c * cs;
          acc[c] += w1 * p[o1] + w2 * p[o2] + w3 * p[o3] + w4 * p[o4];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCS; ++c) out_c0[(long)c * PHPW + bin] = acc[c] / g.count;
  }
}

template <int NCS>
__device__ __forceinline__ void fwd_batch_lds(float* win, const float* __restrict__ plane_c0,
                                              int HW, int W, int H, const RoiGeom& g, int wh,
                                              int ww, int PH, int PW,
                                              float* __restrict__ out_c0) {
  const int warea = wh * ww;
  const float inv_ww = 1.f / (float)ww;
  for (int idx = threadIdx.x; idx < warea; idx += kThreads) {
    const int y = (int)(((float)idx + 0.5f) * inv_ww);
    const int x = idx - y * ww;
    const float* p = plane_c0 + (long)(g.wy0 + y) * W + (g.wx0 + x);
    float v[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) v[c] = p[(long)c * HW];
#pragma unroll
    for (int c = 0; c < NCS; ++c) win[c * warea + idx] = v[c];
  }
  __syncthreads();
  fwd_pool<NCS>(win, warea, ww, g.wy0, g.wx0, g, H, W, PH, PW, out_c0);
  __syncthreads();
}


// ---- 4-channel interleaved LDS window: win4[group][pos] = {c0,c1,c2,c3 at that cell} -------------------
// One ds_read_b128 per tap serves four channels (4 LDS cycles instead of 4 x 2 for ds_read_b32), the group
// stride is a compile-time constant so the channel group selects an instruction immediate instead of costing
// address VALU, and the per-component arithmetic keeps the reference's order ((w1*v1 + w2*v2) + w3*v3) + w4*v4.
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NCS, bool FAST>
__device__ __forceinline__ void fwd_pool4(const f4* win4, int rs, const RoiGeom& g, int H, int W, int PH,
                                          int PW, float* __restrict__ out_c0) {
  constexpr int NG = NCS / 4;
  constexpr int SG = kFwdLdsFloats / NCS;  // positions per channel group
  const int PHPW = PH * PW;
  for (int bin = threadIdx.x; bin < PHPW; bin += kThreads) {
    const int ph = bin / PW;
    const int pw = bin - ph * PW;
    f4 acc[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) acc[k] = (f4)(0.f);
    for (int iy = 0; iy < g.gh; ++iy) {
      const float y = sample_coord_t<FAST>(g.start_h, ph, g.bin_h, iy, g.gh, g.inv_gh);
      int yl, yh;
      float ly, hy;
      if (!axis_sample(y, H, yl, yh, ly, hy)) continue;
      const int ryl = (yl - g.wy0) * rs, ryh = (yh - g.wy0) * rs;
      for (int ix = 0; ix < g.gw; ++ix) {
        const float x = sample_coord_t<FAST>(g.start_w, pw, g.bin_w, ix, g.gw, g.inv_gw);
        int xl, xh;
        float lx, hx;
        if (!axis_sample(x, W, xl, xh, lx, hx)) continue;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const f4* p1 = win4 + ryl + (xl - g.wx0);
        const f4* p2 = win4 + ryl + (xh - g.wx0);
        const f4* p3 = win4 + ryh + (xl - g.wx0);
        const f4* p4 = win4 + ryh + (xh - g.wx0);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
          acc[k] += w1 * p1[k * SG] + w2 * p2[k * SG] + w3 * p3[k * SG] + w4 * p4[k * SG];
          // keep the scheduler from hoisting all 4*NG ds_read_b128 (64 VGPRs) ahead of the math
          if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const f4 o = FAST ? acc[k] * g.inv_count : acc[k] / g.count;
      (out_c0 + (long)(4 * k) * PHPW)[bin] = o.x;
      (out_c0 + (long)(4 * k + 1) * PHPW)[bin] = o.y;
      (out_c0 + (long)(4 * k + 2) * PHPW)[bin] = o.z;
      (out_c0 + (long)(4 * k + 3) * PHPW)[bin] = o.w;
    }
  }
}

template <int NCS>
__device__ __forceinline__ void fwd_batch_lds4(float* win, const float* __restrict__ plane_c0, int HW, int W,
                                               int H, const RoiGeom& g, int wh, int ww, int PH, int PW,
                                               float* __restrict__ out_c0) {
  constexpr int NG = NCS / 4;
  constexpr int SG = kFwdLdsFloats / NCS;
  f4* win4 = (f4*)win;
  const int warea = wh * ww;
  const float inv_ww = 1.f / (float)ww;
  for (int idx = threadIdx.x; idx < warea; idx += kThreads) {
    const int y = (int)(((float)idx + 0.5f) * inv_ww);
    const int x = idx - y * ww;
    const int off = (g.wy0 + y) * W + (g.wx0 + x);  // per-lane 32-bit offset; the channel base stays scalar
    float v[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) v[c] = (plane_c0 + (long)c * HW)[off];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      f4 t;
      t.x = v[4 * k]; t.y = v[4 * k + 1]; t.z = v[4 * k + 2]; t.w = v[4 * k + 3];
      win4[k * SG + idx] = t;
    }
  }
  __syncthreads();
  if (g.pow2)
    fwd_pool4<NCS, true>(win4, ww, g, H, W, PH, PW, out_c0);
  else
    fwd_pool4<NCS, false>(win4, ww, g, H, W, PH, PW, out_c0);
  __syncthreads();
}

// ---- per-workgroup sample tables (round 6) ---------------------------------------------------------------------------
// A RoI's sample geometry is the same for every channel, and per axis there are only PH * gh (PW * gw) distinct samples.
// They are evaluated ONCE per workgroup into two LDS tables -- window-relative tap offsets + the low-side weight, 8 bytes
// per entry -- by the very expressions of the per-lane code above (every bit of every weight is what it was), instead of
// once per lane, sample, channel batch and wave (~55 of the ~160 vector instructions a lane spent per sample and batch).
// The lane <-> bin mapping and the 16-channel batches (which amortise the tap addresses best) stay.
constexpr int kTabMax = 64;                  // entries per axis table: larger sampling grids take the per-lane path
struct AxisTap { short lo, hi; float l; };   // lo < 0: the sample lies outside the map (contributes nothing)

template <bool FAST>
__device__ __forceinline__ void fill_axis_table(AxisTap* tab, int P, int grid, float start, float bin, float inv_g, int size,
                                                int w0, int scale_lo, int t) {
  if (t >= 0 && t < P * grid) {
    const int p = t / grid, i = t - p * grid;
    const float v = sample_coord_t<FAST>(start, p, bin, i, grid, inv_g);
    int lo, hi;
    float l, h;
    AxisTap e;
    if (axis_sample(v, size, lo, hi, l, h)) {
      e.lo = (short)((lo - w0) * scale_lo);
      e.hi = (short)((hi - w0) * scale_lo);
      e.l = l;
    } else {
      e.lo = -1; e.hi = -1; e.l = 0.f;
    }
    tab[t] = e;
  }
}

// fwd_pool4 with the geometry read from the tables; `yrow` / `xrow` = the lane's first table entries (ph * gh, pw * gw)
template <int NCS, bool FAST>
__device__ __forceinline__ void fwd_pool4_tab(const f4* win4, const AxisTap* ytab, const AxisTap* xtab, int bin, int yrow,
                                              int xrow, const RoiGeom& g, int PHPW, float* __restrict__ out_c0) {
  constexpr int NG = NCS / 4;
  constexpr int SG = kFwdLdsFloats / NCS;  // positions per channel group
  if (bin >= PHPW) return;
  f4 acc[NG];
#pragma unroll
  for (int k = 0; k < NG; ++k) acc[k] = (f4)(0.f);
  for (int iy = 0; iy < g.gh; ++iy) {
    const AxisTap ye = ytab[yrow + iy];
    if (ye.lo < 0) continue;
    const float ly = ye.l, hy = 1.f - ly;
    for (int ix = 0; ix < g.gw; ++ix) {
      const AxisTap xe = xtab[xrow + ix];
      if (xe.lo < 0) continue;
      const float lx = xe.l, hx = 1.f - lx;
      const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
      const f4* p1 = win4 + (ye.lo + xe.lo);
      const f4* p2 = win4 + (ye.lo + xe.hi);
      const f4* p3 = win4 + (ye.hi + xe.lo);
      const f4* p4 = win4 + (ye.hi + xe.hi);
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        acc[k] += w1 * p1[k * SG] + w2 * p2[k * SG] + w3 * p3[k * SG] + w4 * p4[k * SG];
        if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    const f4 o = FAST ? acc[k] * g.inv_count : acc[k] / g.count;
    (out_c0 + (long)(4 * k) * PHPW)[bin] = o.x;
    (out_c0 + (long)(4 * k + 1) * PHPW)[bin] = o.y;
    (out_c0 + (long)(4 * k + 2) * PHPW)[bin] = o.z;
    (out_c0 + (long)(4 * k + 3) * PHPW)[bin] = o.w;
  }
}

template <int NCS>
__device__ __forceinline__ void fwd_batch_lds4_tab(float* win, const float* __restrict__ plane_c0, int HW, int W,
                                                   const RoiGeom& g, int wh, int ww, int PHPW, const AxisTap* ytab,
                                                   const AxisTap* xtab, int bin, int yrow, int xrow,
                                                   float* __restrict__ out_c0) {
  constexpr int NG = NCS / 4;
  constexpr int SG = kFwdLdsFloats / NCS;
  f4* win4 = (f4*)win;
  const int warea = wh * ww;
  const float inv_ww = 1.f / (float)ww;
  for (int idx = threadIdx.x; idx < warea; idx += kThreads) {
    const int y = (int)(((float)idx + 0.5f) * inv_ww);
    const int x = idx - y * ww;
    const int off = (g.wy0 + y) * W + (g.wx0 + x);
    float v[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) v[c] = (plane_c0 + (long)c * HW)[off];
#pragma unroll
    for (int k = 0; k < NG; ++k) win4[k * SG + idx] = (f4){v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
  }
  __syncthreads();  // (also publishes the tables before their first use)
  if (g.pow2) fwd_pool4_tab<NCS, true>(win4, ytab, xtab, bin, yrow, xrow, g, PHPW, out_c0);
  else fwd_pool4_tab<NCS, false>(win4, ytab, xtab, bin, yrow, xrow, g, PHPW, out_c0);
  __syncthreads();
}

__global__ __launch_bounds__(kThreads) void roi_align_fwd_kernel(
    const float* __restrict__ in, const float* __restrict__ rois, float* __restrict__ out,
    int R, int batch, int C, int H, int W, int PH, int PW, float scale, int sampling_ratio) {
  __shared__ __attribute__((aligned(16))) float win[kFwdLdsFloats];
  __shared__ AxisTap ytab[kTabMax], xtab[kTabMax];
  const int r = blockIdx.x % R;
  const int ct = blockIdx.x / R;
  const int c_begin = ct * kCPB;
  const int c_end = min(C, c_begin + kCPB);
  const int PHPW = PH * PW;
  const int HW = H * W;
  const RoiGeom g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
  float* out_r = out + (long)r * C * PHPW;

  if (g.empty) {
    // every sample is out of range (or the RoI is malformed): the reference emits zeros
    const int n = (c_end - c_begin) * PHPW;
    for (int i = threadIdx.x; i < n; i += kThreads) out_r[(long)c_begin * PHPW + i] = 0.f;
    return;
  }
  const int wh = g.wy1 - g.wy0 + 1, ww = g.wx1 - g.wx0 + 1;
  const int warea = wh * ww;
  const int cs_max = min(kMaxBatch, kFwdLdsFloats / warea);
  const float* img = in + (long)g.b * C * HW;

  int c = c_begin;
  // table-driven batches: both axis tables fit, one pass of the workgroup covers the bins, tap offsets fit 16 bits
  const bool tables = PH * g.gh <= kTabMax && PW * g.gw <= kTabMax && PHPW <= kThreads && warea <= 4352;
  int t_bin = threadIdx.x, t_yrow = 0, t_xrow = 0;
  if (tables) {
    if (g.pow2) {
      fill_axis_table<true>(ytab, PH, g.gh, g.start_h, g.bin_h, g.inv_gh, H, g.wy0, ww, threadIdx.x);
      fill_axis_table<true>(xtab, PW, g.gw, g.start_w, g.bin_w, g.inv_gw, W, g.wx0, 1, (int)threadIdx.x - 128);
    } else {
      fill_axis_table<false>(ytab, PH, g.gh, g.start_h, g.bin_h, g.inv_gh, H, g.wy0, ww, threadIdx.x);
      fill_axis_table<false>(xtab, PW, g.gw, g.start_w, g.bin_w, g.inv_gw, W, g.wx0, 1, (int)threadIdx.x - 128);
    }
    const int ph = (int)(((float)t_bin + 0.5f) * (1.f / (float)PW));  // exact for these sizes; once per lane, not per batch
    t_yrow = ph * g.gh;
    t_xrow = (t_bin - ph * PW) * g.gw;
  }
  while (c < c_end) {
    const int left = c_end - c;
    const float* plane = img + (long)c * HW;
    float* o = out_r + (long)c * PHPW;
    const int n = min(left, cs_max);
    if (tables && n >= 16) {
      fwd_batch_lds4_tab<16>(win, plane, HW, W, g, wh, ww, PHPW, ytab, xtab, t_bin, t_yrow, t_xrow, o);
      c += 16;
    } else if (tables && n >= 8) {
      fwd_batch_lds4_tab<8>(win, plane, HW, W, g, wh, ww, PHPW, ytab, xtab, t_bin, t_yrow, t_xrow, o);
      c += 8;
    } else if (tables && n >= 4) {
      fwd_batch_lds4_tab<4>(win, plane, HW, W, g, wh, ww, PHPW, ytab, xtab, t_bin, t_yrow, t_xrow, o);
      c += 4;
    } else if (n >= 16) {
      fwd_batch_lds4<16>(win, plane, HW, W, H, g, wh, ww, PH, PW, o);
      c += 16;
    } else if (n >= 8) {
      fwd_batch_lds4<8>(win, plane, HW, W, H, g, wh, ww, PH, PW, o);
      c += 8;
    } else if (n >= 4) {
      fwd_batch_lds4<4>(win, plane, HW, W, H, g, wh, ww, PH, PW, o);
      c += 4;
    } else if (n >= 2) {
      fwd_batch_lds<2>(win, plane, HW, W, H, g, wh, ww, PH, PW, o);
      c += 2;
    } else if (n == 1) {
      fwd_batch_lds<1>(win, plane, HW, W, H, g, wh, ww, PH, PW, o);
      c += 1;
    } else {
      // window larger than the LDS budget (very large feature maps): gather from global
      fwd_pool<1>(plane, HW, W, 0, 0, g, H, W, PH, PW, o);
      c += 1;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Forward fused with the consumer's stride: the res5 head's first 1x1 convolution (and its projection
// shortcut) has stride 2 (STRIDE_IN_1X1, resnet.py:155-204,258-275), so three quarters of the 14x14 pooled
// bins are never read.  This variant pools only bins (bs*i, bs*j) and writes them as [R, OH, OW, C] (NHWC),
// the layout the head's GEMMs want: a quarter of the output bytes, no strided slice / layout copy afterwards.
// Per-bin arithmetic is the bit-exact kernel's, so out[r, i, j, c] == roi_align_forward(...)[r, c, bs*i, bs*j].
// Work item = (output bin, 4-channel group): every lane pools one float4 and stores 16 contiguous bytes.
// ---------------------------------------------------------------------------------------
// Pair-layout stores (csrc/split_gemm.hip: per 32 channels 64 B bf16 hi | 64 B bf16 lo): the strided pooler can hand
// its bins to the res5 head's first GEMM directly, without an fp32 copy and a split pass in between.
typedef __bf16 pool_b2 __attribute__((ext_vector_type(2)));
typedef float pool_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pool_pack_bf16(float a, float b) {
  pool_f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pool_b2));
}
__device__ __forceinline__ void pool_store4_pair(char* pair_r, long obin, int c, int C, f4 o) {
  const unsigned h01 = pool_pack_bf16(o.x, o.y), h23 = pool_pack_bf16(o.z, o.w);
  const unsigned l01 = pool_pack_bf16(o.x - __uint_as_float(h01 << 16), o.y - __uint_as_float(h01 & 0xffff0000u));
  const unsigned l23 = pool_pack_bf16(o.z - __uint_as_float(h23 << 16), o.w - __uint_as_float(h23 & 0xffff0000u));
  char* d = pair_r + obin * 4L * C + (long)(c >> 5) * 128 + (c & 31) * 2;
  *(uint2*)d = make_uint2(h01, h23);
  *(uint2*)(d + 64) = make_uint2(l01, l23);
}
__device__ __forceinline__ void pool_store1_pair(char* pair_r, long obin, int c, int C, float o) {
  const __bf16 h = (__bf16)o;
  const __bf16 l = (__bf16)(o - (float)h);
  char* d = pair_r + obin * 4L * C + (long)(c >> 5) * 128 + (c & 31) * 2;
  *(__bf16*)d = h;
  *(__bf16*)(d + 64) = l;
}

template <int NCS, bool FAST, int LDSF = kFwdLdsFloats>
__device__ __forceinline__ void fwd_pool4_strided(const f4* win4, int rs, const RoiGeom& g, int H, int W, int bs,
                                                  int OH, int OW, float* __restrict__ out_rc, int C,
                                                  char* pair_r = nullptr, int c_abs = 0) {
  constexpr int NG = NCS / 4;
  constexpr int SG = LDSF / NCS;
  const int items = OH * OW * NG;
  for (int item = threadIdx.x; item < items; item += kThreads) {
    const int k = item % NG, obin = item / NG;
    const int oh = obin / OW;
    const int ph = oh * bs, pw = (obin - oh * OW) * bs;
    const f4* base = win4 + k * SG;
    f4 acc = (f4)(0.f);
    for (int iy = 0; iy < g.gh; ++iy) {
      const float y = sample_coord_t<FAST>(g.start_h, ph, g.bin_h, iy, g.gh, g.inv_gh);
      int yl, yh;
      float ly, hy;
      if (!axis_sample(y, H, yl, yh, ly, hy)) continue;
      const int ryl = (yl - g.wy0) * rs, ryh = (yh - g.wy0) * rs;
      for (int ix = 0; ix < g.gw; ++ix) {
        const float x = sample_coord_t<FAST>(g.start_w, pw, g.bin_w, ix, g.gw, g.inv_gw);
        int xl, xh;
        float lx, hx;
        if (!axis_sample(x, W, xl, xh, lx, hx)) continue;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        acc += w1 * base[ryl + (xl - g.wx0)] + w2 * base[ryl + (xh - g.wx0)] + w3 * base[ryh + (xl - g.wx0)] +
               w4 * base[ryh + (xh - g.wx0)];
      }
    }
    const f4 o = FAST ? acc * g.inv_count : acc / g.count;
    if (pair_r) pool_store4_pair(pair_r, obin, c_abs + 4 * k, C, o);
    else *(f4*)(out_rc + (long)obin * C + 4 * k) = o;
  }
}

// Scalar-layout fallback (windows too large for four interleaved channels; `src` may be the global plane).
template <int NCS>
__device__ __forceinline__ void fwd_pool_strided(const float* src, int cs, int rs, int oy, int ox, const RoiGeom& g,
                                                 int H, int W, int bs, int OH, int OW, float* __restrict__ out_rc,
                                                 int C, char* pair_r = nullptr, int c_abs = 0) {
  for (int obin = threadIdx.x; obin < OH * OW; obin += kThreads) {
    const int oh = obin / OW;
    const int ph = oh * bs, pw = (obin - oh * OW) * bs;
    float acc[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) acc[c] = 0.f;
    for (int iy = 0; iy < g.gh; ++iy) {
      int yl, yh;
      float ly, hy;
      if (!axis_sample(sample_coord(g.start_h, ph, g.bin_h, iy, g.gh), H, yl, yh, ly, hy)) continue;
      const int ryl = (yl - oy) * rs, ryh = (yh - oy) * rs;
      for (int ix = 0; ix < g.gw; ++ix) {
        int xl, xh;
        float lx, hx;
        if (!axis_sample(sample_coord(g.start_w, pw, g.bin_w, ix, g.gw), W, xl, xh, lx, hx)) continue;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const int o1 = ryl + (xl - ox), o2 = ryl + (xh - ox), o3 = ryh + (xl - ox), o4 = ryh + (xh - ox);
#pragma unroll
        for (int c = 0; c < NCS; ++c) {
          const float* p = src + c * cs;
          acc[c] += w1 * p[o1] + w2 * p[o2] + w3 * p[o3] + w4 * p[o4];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCS; ++c) {
      if (pair_r) pool_store1_pair(pair_r, obin, c_abs + c, C, acc[c] / g.count);
      else out_rc[(long)obin * C + c] = acc[c] / g.count;
    }
  }
}

template <int NCS, int LDSF = kFwdLdsFloats>
__device__ __forceinline__ void stage_window4(float* win, const float* __restrict__ plane_c0, int HW, int W,
                                              const RoiGeom& g, int wh, int ww) {
  constexpr int NG = NCS / 4;
  constexpr int SG = LDSF / NCS;
  f4* win4 = (f4*)win;
  const int warea = wh * ww;
  const float inv_ww = 1.f / (float)ww;
  for (int idx = threadIdx.x; idx < warea; idx += kThreads) {
    const int y = (int)(((float)idx + 0.5f) * inv_ww);
    const int x = idx - y * ww;
    const int off = (g.wy0 + y) * W + (g.wx0 + x);
    float v[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) v[c] = (plane_c0 + (long)c * HW)[off];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      f4 t;
      t.x = v[4 * k]; t.y = v[4 * k + 1]; t.z = v[4 * k + 2]; t.w = v[4 * k + 3];
      win4[k * SG + idx] = t;
    }
  }
}

template <int NCS, int LDSF = kFwdLdsFloats>
__device__ __forceinline__ void strided_batch4(float* win, const float* __restrict__ plane, int HW, int H, int W,
                                               const RoiGeom& g, int wh, int ww, int bs, int OH, int OW,
                                               float* __restrict__ out_rc, int C, char* pair_r, int c_abs) {
  stage_window4<NCS, LDSF>(win, plane, HW, W, g, wh, ww);
  __syncthreads();
  if (g.pow2)
    fwd_pool4_strided<NCS, true, LDSF>((const f4*)win, ww, g, H, W, bs, OH, OW, out_rc, C, pair_r, c_abs);
  else
    fwd_pool4_strided<NCS, false, LDSF>((const f4*)win, ww, g, H, W, bs, OH, OW, out_rc, C, pair_r, c_abs);
  __syncthreads();
}

// Two launches share the RoIs by window size: windows of up to kSmallWindow cells (most proposals) run with the 17 KB
// window budget and eight workgroups per CU; larger windows with a 64 KB budget, so that they still stage 8-16
// channels per batch instead of 1-4 (a 30x40-cell window took 32 single-channel batches per workgroup and those
// few RoIs set the kernel's duration).  A workgroup whose RoI belongs to the other launch exits after the geometry.
constexpr int kSmallWindow = 272;      // 4352 / 16: the 17 KB budget still holds 16 channels
constexpr int kBigLdsFloats = 16384;   // 64 KB

template <int LDSF, bool BIG>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_strided_nhwc_kernel(
    const float* __restrict__ in, const float* __restrict__ rois, float* __restrict__ out, int R, int batch, int C,
    int H, int W, int PH, int PW, int bs, int OH, int OW, float scale, int sampling_ratio, int pair_out) {
  __shared__ __attribute__((aligned(16))) float win[LDSF];
  const int r = blockIdx.x % R;
  const int ct = blockIdx.x / R;
  const int c_begin = ct * kCPB;
  const int c_end = min(C, c_begin + kCPB);
  const int HW = H * W;
  const RoiGeom g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
  float* out_r = out + (long)r * OH * OW * C;   // both layouts take 4*C bytes per bin
  char* pair_r = pair_out ? (char*)out_r : nullptr;
  if (g.empty) {
    if (BIG) return;
    for (int i = threadIdx.x; i < OH * OW * (c_end - c_begin); i += kThreads) {
      const long obin = i / (c_end - c_begin);
      const int c = c_begin + i % (c_end - c_begin);
      if (pair_r) pool_store1_pair(pair_r, obin, c, C, 0.f);
      else out_r[obin * C + c] = 0.f;
    }
    return;
  }
  const int wh = g.wy1 - g.wy0 + 1, ww = g.wx1 - g.wx0 + 1;
  const int warea = wh * ww;
  if ((warea > kSmallWindow) != BIG) return;  // the other launch owns this RoI (empty RoIs: handled above, by the small one)
  // small windows (most proposals: <= 136 cells): all 32 channels of the block in ONE batch -- half the barriers, and
  // every bin is written as a whole 128-byte line (fp32: 32 channels; pair layout: 64 B hi | 64 B lo)
  const int cs_max = min(2 * kMaxBatch, LDSF / warea);
  const float* img = in + (long)g.b * C * HW;
  const bool vec_ok = (C & 3) == 0;  // float4 stores need 16-byte aligned channel groups
  int c = c_begin;
  while (c < c_end) {
    const int left = c_end - c;
    const float* plane = img + (long)c * HW;
    float* o = out_r + c;
    const int n = min(left, cs_max);
    if (n >= 32 && vec_ok) {
      strided_batch4<32, LDSF>(win, plane, HW, H, W, g, wh, ww, bs, OH, OW, o, C, pair_r, c);
      c += 32;
    } else if (n >= 16 && vec_ok) {
      strided_batch4<16, LDSF>(win, plane, HW, H, W, g, wh, ww, bs, OH, OW, o, C, pair_r, c);
      c += 16;
    } else if (n >= 8 && vec_ok) {
      strided_batch4<8, LDSF>(win, plane, HW, H, W, g, wh, ww, bs, OH, OW, o, C, pair_r, c);
      c += 8;
    } else if (n >= 4 && vec_ok) {
      strided_batch4<4, LDSF>(win, plane, HW, H, W, g, wh, ww, bs, OH, OW, o, C, pair_r, c);
      c += 4;
    } else if (n >= 1) {  // one channel through LDS in the plain layout
      for (int idx = threadIdx.x; idx < warea; idx += kThreads) {
        const int y = idx / ww, x = idx - (idx / ww) * ww;
        win[idx] = plane[(g.wy0 + y) * W + (g.wx0 + x)];
      }
      __syncthreads();
      fwd_pool_strided<1>(win, warea, ww, g.wy0, g.wx0, g, H, W, bs, OH, OW, o, C, pair_r, c);
      __syncthreads();
      c += 1;
    } else {  // window larger than the LDS budget: gather from global
      fwd_pool_strided<1>(plane, HW, W, 0, 0, g, H, W, bs, OH, OW, o, C, pair_r, c);
      c += 1;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Strided pooler on an NHWC (channels-last) map.  The trunk computes in NHWC; read in that layout every bilinear tap
// of a sample is one contiguous C-vector, so the pooler needs no LDS window staging at all: lanes run over channel
// groups of 4 (one 16-byte load per tap, a whole 4 KB line per wave-pair at C = 1024), the 256 threads of a workgroup
// own one RoI and walk its OH x OW bins together.  Per (bin, channel) the samples are visited in the reference's order
// with the reference's expressions (fwd_pool4_strided above, the exact kernel's arithmetic), so the result is
// BIT-IDENTICAL to the NCHW kernels; neighbouring samples' taps are served by L1 / L2 (the map is 17 MB per image).
// ---------------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ void pool_nhwc_strided(const f4* __restrict__ img4, const RoiGeom& g, int H, int W, int C4,
                                                  int bs, int OH, int OW, float* __restrict__ out_r, char* pair_r,
                                                  int k0 = 0, int KT = -1) {
  // channel groups [k0, k0 + KT) of the C4 groups of a cell (KT < 0: all of them)
  if (KT < 0) KT = C4;
  const int items = OH * OW * KT;
  const int C = C4 * 4;
  if (g.gh == 2 && g.gw == 2) {
    // sampling_ratio 2 (every shipped config): the 2 x 2 samples of a bin are half a bin apart, so for the many RoIs
    // whose bins are smaller than two cells they fall into the same pair of rows and / or columns and their taps are
    // the SAME cells.  Each distinct (row pair, column pair) is fetched once -- 4 or 8 loads per bin instead of 16 on
    // those RoIs (the kernel is bound by the L2 -> CU path) -- and the samples are accumulated from registers in the
    // reference's order with the reference's expressions: the bits do not change.
    for (int item = threadIdx.x; item < items; item += kThreads) {
      const int k = k0 + item % KT, obin = item / KT;
      const int oh = obin / OW;
      const int ph = oh * bs, pw = (obin - oh * OW) * bs;
      const f4* base = img4 + k;
      int yl[2], yh[2], xl[2], xh[2];
      float ly[2], hy[2], lx[2], hx[2];
      bool oky[2], okx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        oky[i] = axis_sample(sample_coord_t<FAST>(g.start_h, ph, g.bin_h, i, 2, g.inv_gh), H, yl[i], yh[i], ly[i], hy[i]);
        okx[i] = axis_sample(sample_coord_t<FAST>(g.start_w, pw, g.bin_w, i, 2, g.inv_gw), W, xl[i], xh[i], lx[i], hx[i]);
      }
      const bool same_rows = oky[0] && oky[1] && yl[0] == yl[1] && yh[0] == yh[1];
      const bool same_cols = okx[0] && okx[1] && xl[0] == xl[1] && xh[0] == xh[1];
      f4 t[2][2][4];
      auto load4 = [&](int iy, int ix) {
        const long ryl = (long)yl[iy] * W * C4, ryh = (long)yh[iy] * W * C4;
        const long cl = (long)xl[ix] * C4, ch = (long)xh[ix] * C4;
        t[iy][ix][0] = base[ryl + cl];
        t[iy][ix][1] = base[ryl + ch];
        t[iy][ix][2] = base[ryh + cl];
        t[iy][ix][3] = base[ryh + ch];
      };
      auto copy4 = [&](int iy, int ix, int sy, int sx) {
#pragma unroll
        for (int q = 0; q < 4; ++q) t[iy][ix][q] = t[sy][sx][q];
      };
      if (oky[0] && okx[0]) load4(0, 0);
      if (oky[0] && okx[1]) { if (same_cols) copy4(0, 1, 0, 0); else load4(0, 1); }
      if (oky[1] && okx[0]) { if (same_rows) copy4(1, 0, 0, 0); else load4(1, 0); }
      if (oky[1] && okx[1]) {
        if (same_rows) copy4(1, 1, 0, 1);
        else if (same_cols) copy4(1, 1, 1, 0);
        else load4(1, 1);
      }
      f4 acc = (f4)(0.f);
#pragma unroll
      for (int iy = 0; iy < 2; ++iy) {
        if (!oky[iy]) continue;
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
          if (!okx[ix]) continue;
          const float w1 = hy[iy] * hx[ix], w2 = hy[iy] * lx[ix], w3 = ly[iy] * hx[ix], w4 = ly[iy] * lx[ix];
          acc += w1 * t[iy][ix][0] + w2 * t[iy][ix][1] + w3 * t[iy][ix][2] + w4 * t[iy][ix][3];
        }
      }
      const f4 o = FAST ? acc * g.inv_count : acc / g.count;
      if (pair_r) pool_store4_pair(pair_r, obin, 4 * k, C, o);
      else *(f4*)(out_r + (long)obin * C + 4 * k) = o;
    }
    return;
  }
  for (int item = threadIdx.x; item < items; item += kThreads) {
    const int k = k0 + item % KT, obin = item / KT;
    const int oh = obin / OW;
    const int ph = oh * bs, pw = (obin - oh * OW) * bs;
    const f4* base = img4 + k;
    f4 acc = (f4)(0.f);
    for (int iy = 0; iy < g.gh; ++iy) {
      const float y = sample_coord_t<FAST>(g.start_h, ph, g.bin_h, iy, g.gh, g.inv_gh);
      int yl, yh;
      float ly, hy;
      if (!axis_sample(y, H, yl, yh, ly, hy)) continue;
      const long ryl = (long)yl * W * C4, ryh = (long)yh * W * C4;
      for (int ix = 0; ix < g.gw; ++ix) {
        const float x = sample_coord_t<FAST>(g.start_w, pw, g.bin_w, ix, g.gw, g.inv_gw);
        int xl, xh;
        float lx, hx;
        if (!axis_sample(x, W, xl, xh, lx, hx)) continue;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const long cl = (long)xl * C4, ch = (long)xh * C4;
        acc += w1 * base[ryl + cl] + w2 * base[ryl + ch] + w3 * base[ryh + cl] + w4 * base[ryh + ch];
      }
    }
    const f4 o = FAST ? acc * g.inv_count : acc / g.count;
    if (pair_r) pool_store4_pair(pair_r, obin, 4 * k, C, o);
    else *(f4*)(out_r + (long)obin * C + 4 * k) = o;
  }
}

__global__ __launch_bounds__(kThreads) void roi_align_fwd_nhwc_in_strided_kernel(
    const float* __restrict__ in, const float* __restrict__ rois, float* __restrict__ out, int R, int batch, int C,
    int H, int W, int PH, int PW, int bs, int OH, int OW, float scale, int sampling_ratio, int pair_out) {
  const int r = blockIdx.x;
  const RoiGeom g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
  float* out_r = out + (long)r * OH * OW * C;   // both layouts take 4*C bytes per bin
  char* pair_r = pair_out ? (char*)out_r : nullptr;
  const int C4 = C >> 2;
  if (g.empty) {
    for (int item = threadIdx.x; item < OH * OW * C4; item += kThreads) {
      const int k = item % C4, obin = item / C4;
      if (pair_r) pool_store4_pair(pair_r, obin, 4 * k, C, (f4)(0.f));
      else *(f4*)(out_r + (long)obin * C + 4 * k) = (f4)(0.f);
    }
    return;
  }
  const f4* img4 = (const f4*)(in + (long)g.b * H * W * C);
  if (g.pow2) pool_nhwc_strided<true>(img4, g, H, W, C4, bs, OH, OW, out_r, pair_r);
  else pool_nhwc_strided<false>(img4, g, H, W, C4, bs, OH, OW, out_r, pair_r);
}

// ---------------------------------------------------------------------------------------
// NHWC map, window staged in LDS.  The direct form above fetches every tap from L1 / L2: 4-16 loads of 16 bytes per (bin,
// 4 channels), ~1.6 MB per RoI at C = 1024, and is bound by the L2 -> CU path (3.2 GB per 2000-RoI launch: 0.25 ms).  A
// RoI's WINDOW is ~45 cells (median; p90 ~430): staged once per (RoI, 32-channel tile) -- every cell's 32 channels are one
// contiguous 128-byte line in NHWC -- the same bins read their taps from LDS (fwd_pool4_strided, the arithmetic of the
// bit-exact kernel: identical bits) and the L2 -> CU traffic drops to the window bytes.  Windows of more than
// kSmallWindow cells keep the direct form, on the workgroup's own 32 channels (handing them to a second launch with one
// workgroup per RoI was tried: the few large RoIs then run on a few workgroups and set the duration -- 289 us in the step
// against 176 us for this form and 248 us for the direct form alone).
// ---------------------------------------------------------------------------------------
template <int NCS, int LDSF = kFwdLdsFloats>
__device__ __forceinline__ void stage_window4_nhwc(float* win, const float* __restrict__ img_c0, int C, int W,
                                                   const RoiGeom& g, int wh, int ww) {
  constexpr int NG = NCS / 4;
  constexpr int SG = LDSF / NCS;
  f4* win4 = (f4*)win;
  const int n = wh * ww * NG;
  const float inv_ww = 1.f / (float)ww;
  for (int i = threadIdx.x; i < n; i += kThreads) {
    const int k = i % NG, idx = i / NG;   // NG consecutive lanes read one contiguous line of a cell
    const int y = (int)(((float)idx + 0.5f) * inv_ww);
    const int x = idx - y * ww;
    win4[k * SG + idx] = *(const f4*)(img_c0 + ((long)(g.wy0 + y) * W + (g.wx0 + x)) * C + 4 * k);
  }
}

template <int NCS>
__device__ __forceinline__ void strided_batch4_nhwc(float* win, const float* __restrict__ img_c0, int C, int H, int W,
                                                    const RoiGeom& g, int wh, int ww, int bs, int OH, int OW,
                                                    float* __restrict__ out_rc, char* pair_r, int c_abs) {
  stage_window4_nhwc<NCS>(win, img_c0, C, W, g, wh, ww);
  __syncthreads();
  if (g.pow2)
    fwd_pool4_strided<NCS, true>((const f4*)win, ww, g, H, W, bs, OH, OW, out_rc, C, pair_r, c_abs);
  else
    fwd_pool4_strided<NCS, false>((const f4*)win, ww, g, H, W, bs, OH, OW, out_rc, C, pair_r, c_abs);
  __syncthreads();
}

__global__ __launch_bounds__(kThreads) void roi_align_fwd_nhwc_in_strided_lds_kernel(
    const float* __restrict__ in, const float* __restrict__ rois, float* __restrict__ out, int R, int batch, int C,
    int H, int W, int PH, int PW, int bs, int OH, int OW, float scale, int sampling_ratio, int pair_out) {
  __shared__ __attribute__((aligned(16))) float win[kFwdLdsFloats];
  const int r = blockIdx.x % R;
  const int ct = blockIdx.x / R;            // C % 32 == 0 (checked by the launcher): whole 32-channel tiles
  const int c_begin = ct * kCPB;
  const RoiGeom g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
  float* out_r = out + (long)r * OH * OW * C;
  char* pair_r = pair_out ? (char*)out_r : nullptr;
  if (g.empty) {
    for (int item = threadIdx.x; item < OH * OW * (kCPB / 4); item += kThreads) {
      const int k = item % (kCPB / 4), obin = item / (kCPB / 4);
      if (pair_r) pool_store4_pair(pair_r, obin, c_begin + 4 * k, C, (f4)(0.f));
      else *(f4*)(out_r + (long)obin * C + c_begin + 4 * k) = (f4)(0.f);
    }
    return;
  }
  const int wh = g.wy1 - g.wy0 + 1, ww = g.wx1 - g.wx0 + 1;
  const int warea = wh * ww;
  const float* img = in + (long)g.b * H * W * C;
  if (warea > kSmallWindow) {  // direct form on this workgroup's channels
    if (g.pow2) pool_nhwc_strided<true>((const f4*)img, g, H, W, C >> 2, bs, OH, OW, out_r, pair_r, c_begin >> 2, kCPB / 4);
    else pool_nhwc_strided<false>((const f4*)img, g, H, W, C >> 2, bs, OH, OW, out_r, pair_r, c_begin >> 2, kCPB / 4);
    return;
  }
  if (warea * 32 <= kFwdLdsFloats) {   // <= 136 cells: the whole tile in one batch
    strided_batch4_nhwc<32>(win, img + c_begin, C, H, W, g, wh, ww, bs, OH, OW, out_r + c_begin, pair_r, c_begin);
  } else {                             // <= 272 cells: two batches of 16 channels
    strided_batch4_nhwc<16>(win, img + c_begin, C, H, W, g, wh, ww, bs, OH, OW, out_r + c_begin, pair_r, c_begin);
    strided_batch4_nhwc<16>(win, img + c_begin + 16, C, H, W, g, wh, ww, bs, OH, OW, out_r + c_begin + 16, pair_r,
                            c_begin + 16);
  }
}

// ---------------------------------------------------------------------------------------
// Backward
//
// Bilinear average pooling is separable: out = Ay * win * Ax^T with Ay[ph][y] the summed row
// weights of bin-row ph's samples and Ax[pw][x] likewise.  The gradient of the window is
// Ay^T * G * Ax / count, which each lane evaluates for ONE window cell as a short gather
// over the (ph, pw) bins that touch it -- no atomics inside the tile.  One fp32 atomic per
// (window cell, channel) then folds the RoI's window into grad_input, instead of the
// reference's four atomics per (bin, sample, channel).
// ---------------------------------------------------------------------------------------

template <int NCS>
__device__ __forceinline__ void bwd_batch(const float* Ay, const float* Ax, const int* phlo,
                                          const int* phhi, const int* pwlo, const int* pwhi,
                                          float* G, const float* __restrict__ gout_c0,
                                          float* __restrict__ gin_c0, int HW, int W,
                                          const RoiGeom& g, int wh, int ww, int PH, int PW) {
  const int PHPW = PH * PW;
  const int n = NCS * PHPW;
  for (int i = threadIdx.x; i < n; i += kThreads) G[i] = gout_c0[i];
  __syncthreads();
  const int warea = wh * ww;
  const float inv_ww = 1.f / (float)ww;
  for (int idx = threadIdx.x; idx < warea; idx += kThreads) {
    const int y = (int)(((float)idx + 0.5f) * inv_ww);
    const int x = idx - y * ww;
    const int p0 = phlo[y], p1 = phhi[y], q0 = pwlo[x], q1 = pwhi[x];
    if (p1 < p0 || q1 < q0) continue;
    float s[NCS];
#pragma unroll
    for (int c = 0; c < NCS; ++c) s[c] = 0.f;
    for (int ph = p0; ph <= p1; ++ph) {
      const float ay = Ay[ph * wh + y];
      for (int pw = q0; pw <= q1; ++pw) {
        const float w = ay * Ax[pw * ww + x];
        const float* gp = G + ph * PW + pw;
#pragma unroll
        for (int c = 0; c < NCS; ++c) s[c] += w * gp[c * PHPW];
      }
    }
    float* dst = gin_c0 + (long)(g.wy0 + y) * W + (g.wx0 + x);
#pragma unroll
    for (int c = 0; c < NCS; ++c) atomicAdd(dst + (long)c * HW, s[c] / g.count);
  }
  __syncthreads();
}

__global__ __launch_bounds__(kThreads) void roi_align_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gin,
    int R, int batch, int C, int H, int W, int PH, int PW, float scale, int sampling_ratio) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // layout: Ay[PH*H] | Ax[PW*W] | phlo[H] phhi[H] pwlo[W] pwhi[W] | G[kMaxBatch*PH*PW]
  float* Ay = smem;
  float* Ax = Ay + PH * H;
  int* phlo = (int*)(Ax + PW * W);
  int* phhi = phlo + H;
  int* pwlo = phhi + H;
  int* pwhi = pwlo + W;
  float* G = (float*)(pwhi + W);

  const int r = blockIdx.x % R;
  const int ct = blockIdx.x / R;
  const int c_begin = ct * kCPB;
  const int c_end = min(C, c_begin + kCPB);
  const int PHPW = PH * PW;
  const int HW = H * W;
  const RoiGeom g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
  if (g.empty) return;
  const int wh = g.wy1 - g.wy0 + 1, ww = g.wx1 - g.wx0 + 1;

  for (int i = threadIdx.x; i < PH * wh; i += kThreads) Ay[i] = 0.f;
  for (int i = threadIdx.x; i < PW * ww; i += kThreads) Ax[i] = 0.f;
  for (int i = threadIdx.x; i < wh; i += kThreads) { phlo[i] = 0x7fffffff; phhi[i] = -1; }
  for (int i = threadIdx.x; i < ww; i += kThreads) { pwlo[i] = 0x7fffffff; pwhi[i] = -1; }
  __syncthreads();
  for (int t = threadIdx.x; t < PH * g.gh; t += kThreads) {
    const int ph = t / g.gh, iy = t - ph * g.gh;
    int yl, yh;
    float ly, hy;
    if (axis_sample(sample_coord(g.start_h, ph, g.bin_h, iy, g.gh), H, yl, yh, ly, hy)) {
      yl -= g.wy0; yh -= g.wy0;
      atomicAdd(&Ay[ph * wh + yl], hy);
      atomicAdd(&Ay[ph * wh + yh], ly);
      atomicMin(&phlo[yl], ph); atomicMax(&phhi[yl], ph);
      atomicMin(&phlo[yh], ph); atomicMax(&phhi[yh], ph);
    }
  }
  for (int t = threadIdx.x; t < PW * g.gw; t += kThreads) {
    const int pw = t / g.gw, ix = t - pw * g.gw;
    int xl, xh;
    float lx, hx;
    if (axis_sample(sample_coord(g.start_w, pw, g.bin_w, ix, g.gw), W, xl, xh, lx, hx)) {
      xl -= g.wx0; xh -= g.wx0;
      atomicAdd(&Ax[pw * ww + xl], hx);
      atomicAdd(&Ax[pw * ww + xh], lx);
      atomicMin(&pwlo[xl], pw); atomicMax(&pwhi[xl], pw);
      atomicMin(&pwlo[xh], pw); atomicMax(&pwhi[xh], pw);
    }
  }
  __syncthreads();

  const float* go_r = gout + (long)r * C * PHPW;
  float* gin_b = gin + (long)g.b * C * HW;
  int c = c_begin;
  while (c < c_end) {
    const int left = c_end - c;
    const float* go = go_r + (long)c * PHPW;
    float* gi = gin_b + (long)c * HW;
    if (left >= 16) {
      bwd_batch<16>(Ay, Ax, phlo, phhi, pwlo, pwhi, G, go, gi, HW, W, g, wh, ww, PH, PW);
      c += 16;
    } else if (left >= 8) {
      bwd_batch<8>(Ay, Ax, phlo, phhi, pwlo, pwhi, G, go, gi, HW, W, g, wh, ww, PH, PW);
      c += 8;
    } else if (left >= 4) {
      bwd_batch<4>(Ay, Ax, phlo, phhi, pwlo, pwhi, G, go, gi, HW, W, g, wh, ww, PH, PW);
      c += 4;
    } else if (left >= 2) {
      bwd_batch<2>(Ay, Ax, phlo, phhi, pwlo, pwhi, G, go, gi, HW, W, g, wh, ww, PH, PW);
      c += 2;
    } else {
      bwd_batch<1>(Ay, Ax, phlo, phhi, pwlo, pwhi, G, go, gi, HW, W, g, wh, ww, PH, PW);
      c += 1;
    }
  }
}


// Fallback for shapes whose separable tables do not fit LDS (huge maps / pooled sizes):
// one lane per grad_output element scattering its samples with global atomics.
__global__ __launch_bounds__(kThreads) void roi_align_bwd_scatter_kernel(
    const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gin,
    long total, int batch, int C, int H, int W, int PH, int PW, float scale,
    int sampling_ratio) {
  const int PHPW = PH * PW;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long)gridDim.x * kThreads) {
    const int bin = (int)(i % PHPW);
    const long rc = i / PHPW;
    const int c = (int)(rc % C);
    const int r = (int)(rc / C);
    const int ph = bin / PW, pw = bin - ph * PW;
    const RoiGeom g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
    if (g.empty) continue;
    float* plane = gin + ((long)g.b * C + c) * H * W;
    const float top = gout[i];
    for (int iy = 0; iy < g.gh; ++iy) {
      int yl, yh;
      float ly, hy;
      if (!axis_sample(sample_coord(g.start_h, ph, g.bin_h, iy, g.gh), H, yl, yh, ly, hy)) continue;
      for (int ix = 0; ix < g.gw; ++ix) {
        int xl, xh;
        float lx, hx;
        if (!axis_sample(sample_coord(g.start_w, pw, g.bin_w, ix, g.gw), W, xl, xh, lx, hx)) continue;
        atomicAdd(plane + yl * W + xl, top * (hy * hx) / g.count);
        atomicAdd(plane + yl * W + xh, top * (hy * lx) / g.count);
        atomicAdd(plane + yh * W + xl, top * (ly * hx) / g.count);
        atomicAdd(plane + yh * W + xh, top * (ly * lx) / g.count);
      }
    }
  }
}

}  // namespace

extern "C" int ovis_roi_align_forward_f32(const float* input, const float* rois, float* output,
                                          int num_rois, int batch, int channels, int height,
                                          int width, int pooled_h, int pooled_w,
                                          float spatial_scale, int sampling_ratio,
                                          void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 ||
      pooled_h <= 0 || pooled_w <= 0)
    return OVIS_EINVAL;
  if (num_rois == 0 || channels == 0) return OVIS_OK;  // empty output (ROIAlign_cuda.cu:278-281)
  if (!input || !rois || !output) return OVIS_EINVAL;
  const long n_ct = ovis_ceil_div(channels, kCPB);
  const long blocks = n_ct * num_rois;
  if (blocks > 0x7fffffffL) return OVIS_ERANGE;
  hipLaunchKernelGGL(roi_align_fwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0,
                     (hipStream_t)stream, input, rois, output, num_rois, batch, channels,
                     height, width, pooled_h, pooled_w, spatial_scale, sampling_ratio);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

// roi_align_fwd_mfma.hip
int ovis_roi_align_forward_mfma_launch(const float* input, const float* rois, float* output, int num_rois, int batch,
                                       int channels, int height, int width, int pooled_h, int pooled_w,
                                       float spatial_scale, int sampling_ratio, void* workspace,
                                       size_t workspace_bytes, hipStream_t s);
extern "C" int ovis_roi_align_forward_mfma_supported(int height, int width, int pooled_h, int pooled_w);

// roi_align_bwd_plane.hip
int ovis_roi_align_backward_plane_launch(const float* grad_output, const float* rois, float* grad_input,
                                         int num_rois, int batch, int channels, int height, int width,
                                         int pooled_h, int pooled_w, int bin_stride, float spatial_scale,
                                         int sampling_ratio, void* workspace, size_t workspace_bytes, hipStream_t s,
                                         int nhwc_small);
extern "C" int ovis_roi_align_backward_plane_supported(int height, int width, int pooled_h, int pooled_w);

extern "C" int ovis_roi_align_backward_f32(const float* grad_output, const float* rois,
                                           float* grad_input, int num_rois, int batch,
                                           int channels, int height, int width, int pooled_h,
                                           int pooled_w, float spatial_scale,
                                           int sampling_ratio, void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 ||
      pooled_h <= 0 || pooled_w <= 0)
    return OVIS_EINVAL;
  const size_t in_bytes = (size_t)batch * channels * height * width * sizeof(float);
  if (in_bytes == 0) return OVIS_OK;
  if (!grad_input) return OVIS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  OVIS_HIP_TRY(hipMemsetAsync(grad_input, 0, in_bytes, s));  // at::zeros, ROIAlign_cuda.cu:316
  if (num_rois == 0) return OVIS_OK;
  if (!grad_output || !rois) return OVIS_EINVAL;
  const size_t lds = sizeof(float) * ((size_t)pooled_h * height + (size_t)pooled_w * width +
                                      2 * (size_t)(height + width) +
                                      (size_t)kMaxBatch * pooled_h * pooled_w);
  if (lds <= 64 * 1024) {
    const long n_ct = ovis_ceil_div(channels, kCPB);
    const long blocks = n_ct * num_rois;
    if (blocks > 0x7fffffffL) return OVIS_ERANGE;
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, s,
                       grad_output, rois, grad_input, num_rois, batch, channels, height, width,
                       pooled_h, pooled_w, spatial_scale, sampling_ratio);
  } else {
    const long total = (long)num_rois * channels * pooled_h * pooled_w;
    const long blocks = ovis_ceil_div(total, kThreads);
    hipLaunchKernelGGL(roi_align_bwd_scatter_kernel,
                       dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(kThreads), 0, s,
                       grad_output, rois, grad_input, total, batch, channels, height, width,
                       pooled_h, pooled_w, spatial_scale, sampling_ratio);
  }
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_roi_align_backward_ws_f32(const float* grad_output, const float* rois, float* grad_input,
                                              int num_rois, int batch, int channels, int height, int width,
                                              int pooled_h, int pooled_w, float spatial_scale,
                                              int sampling_ratio, void* workspace, size_t workspace_bytes,
                                              void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return OVIS_EINVAL;
  if ((size_t)batch * channels == 0) return OVIS_OK;
  if (num_rois > 0 && ovis_roi_align_backward_plane_supported(height, width, pooled_h, pooled_w)) {
    if (!grad_output || !rois || !grad_input) return OVIS_EINVAL;
    const int rc = ovis_roi_align_backward_plane_launch(grad_output, rois, grad_input, num_rois, batch, channels,
                                                        height, width, pooled_h, pooled_w, 1, spatial_scale,
                                                        sampling_ratio, workspace, workspace_bytes,
                                                        (hipStream_t)stream, 0);
    if (rc != -100) return rc;  // -100: offsets would not fit 32 bits -> atomic path below
  }
  return ovis_roi_align_backward_f32(grad_output, rois, grad_input, num_rois, batch, channels, height, width,
                                     pooled_h, pooled_w, spatial_scale, sampling_ratio, stream);
}

// Backward of the strided pooler (ovis_roi_align_forward_strided_nhwc_f32): grad_output holds only the bins
// (bin_stride * i, bin_stride * j) as [num_rois, channels, ceil(pooled_h / bin_stride), ceil(pooled_w / bin_stride)]
// tiles -- a quarter of the bytes of the zero-scattered full tile at bin_stride 2.  Plane-owner kernel only: shapes it
// does not cover return OVIS_ERANGE (the caller scatters into a full tile and uses ovis_roi_align_backward_ws_f32).
extern "C" int ovis_roi_align_backward_strided_ws_f32(const float* grad_output, const float* rois, float* grad_input,
                                                      int num_rois, int batch, int channels, int height, int width,
                                                      int pooled_h, int pooled_w, int bin_stride, float spatial_scale,
                                                      int sampling_ratio, void* workspace, size_t workspace_bytes,
                                                      void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0 ||
      bin_stride <= 0)
    return OVIS_EINVAL;
  if ((size_t)batch * channels == 0) return OVIS_OK;
  if (!grad_input) return OVIS_EINVAL;
  if (num_rois == 0)
    return (int)hipMemsetAsync(grad_input, 0, (size_t)batch * channels * height * width * sizeof(float), (hipStream_t)stream);
  const int th = (pooled_h + bin_stride - 1) / bin_stride, tw = (pooled_w + bin_stride - 1) / bin_stride;
  if (!ovis_roi_align_backward_plane_supported(height, width, th, tw)) return OVIS_ERANGE;
  if (!grad_output || !rois) return OVIS_EINVAL;
  const int rc = ovis_roi_align_backward_plane_launch(grad_output, rois, grad_input, num_rois, batch, channels, height,
                                                      width, pooled_h, pooled_w, bin_stride, spatial_scale,
                                                      sampling_ratio, workspace, workspace_bytes, (hipStream_t)stream, 0);
  return rc == -100 ? OVIS_ERANGE : rc;
}

// The same backward from the gradient AS THE PRODUCING GEMM LEAVES IT: NHWC [num_rois, th, tw, channels] fp32 (th, tw =
// ceil(pooled / bin_stride) <= 8, e.g. the 7 x 7 tiles of the res5 head).  The layout change the plane-owner kernel needs
// ([R, C, th, tw] tiles; a separate copy kernel before this entry point existed) hands every value over already split into
// bf16 hi | lo, so the main kernel spends no vector arithmetic on the gradient operand and both stages are two matrix
// instructions.  workspace: ovis_roi_align_backward_strided_nhwc_workspace_bytes.  Shapes it does not cover: OVIS_ERANGE.
extern "C" int ovis_roi_align_backward_strided_nhwc_ws_f32(const float* grad_output_nhwc, const float* rois,
                                                           float* grad_input, int num_rois, int batch, int channels,
                                                           int height, int width, int pooled_h, int pooled_w,
                                                           int bin_stride, float spatial_scale, int sampling_ratio,
                                                           void* workspace, size_t workspace_bytes, void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0 ||
      bin_stride <= 0)
    return OVIS_EINVAL;
  if ((size_t)batch * channels == 0) return OVIS_OK;
  if (!grad_input) return OVIS_EINVAL;
  if (num_rois == 0)
    return (int)hipMemsetAsync(grad_input, 0, (size_t)batch * channels * height * width * sizeof(float), (hipStream_t)stream);
  const int th = (pooled_h + bin_stride - 1) / bin_stride, tw = (pooled_w + bin_stride - 1) / bin_stride;
  if (th > 8 || tw > 8 || !ovis_roi_align_backward_plane_supported(height, width, th, tw)) return OVIS_ERANGE;
  if (!grad_output_nhwc || !rois) return OVIS_EINVAL;
  const int rc = ovis_roi_align_backward_plane_launch(grad_output_nhwc, rois, grad_input, num_rois, batch, channels, height,
                                                      width, pooled_h, pooled_w, bin_stride, spatial_scale,
                                                      sampling_ratio, workspace, workspace_bytes, (hipStream_t)stream, 1);
  return rc == -100 ? OVIS_ERANGE : rc;
}

extern "C" int ovis_roi_align_forward_ws_f32(const float* input, const float* rois, float* output, int num_rois,
                                             int batch, int channels, int height, int width, int pooled_h,
                                             int pooled_w, float spatial_scale, int sampling_ratio, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return OVIS_EINVAL;
  if (num_rois == 0 || channels == 0) return OVIS_OK;
  if (!input || !rois || !output) return OVIS_EINVAL;
  if (ovis_roi_align_forward_mfma_supported(height, width, pooled_h, pooled_w))
    return ovis_roi_align_forward_mfma_launch(input, rois, output, num_rois, batch, channels, height, width, pooled_h,
                                              pooled_w, spatial_scale, sampling_ratio, workspace, workspace_bytes,
                                              (hipStream_t)stream);
  return ovis_roi_align_forward_f32(input, rois, output, num_rois, batch, channels, height, width, pooled_h, pooled_w,
                                    spatial_scale, sampling_ratio, stream);
}

static int strided_nhwc_launch(const float* input, const float* rois, float* output, int num_rois, int batch,
                               int channels, int height, int width, int pooled_h, int pooled_w, int bin_stride,
                               float spatial_scale, int sampling_ratio, int pair_out, void* stream);

extern "C" int ovis_roi_align_forward_strided_nhwc_f32(const float* input, const float* rois, float* output,
                                                       int num_rois, int batch, int channels, int height, int width,
                                                       int pooled_h, int pooled_w, int bin_stride,
                                                       float spatial_scale, int sampling_ratio, void* stream) {
  return strided_nhwc_launch(input, rois, output, num_rois, batch, channels, height, width, pooled_h, pooled_w, bin_stride,
                             spatial_scale, sampling_ratio, 0, stream);
}

extern "C" int ovis_roi_align_forward_strided_pair_f32(const float* input, const float* rois, void* output_pair,
                                                       int num_rois, int batch, int channels, int height, int width,
                                                       int pooled_h, int pooled_w, int bin_stride,
                                                       float spatial_scale, int sampling_ratio, void* stream) {
  if (channels % 32 != 0 || ((uintptr_t)output_pair & 15)) return OVIS_ERANGE;
  return strided_nhwc_launch(input, rois, (float*)output_pair, num_rois, batch, channels, height, width, pooled_h,
                             pooled_w, bin_stride, spatial_scale, sampling_ratio, 1, stream);
}

// input in NHWC ([batch, height, width, channels] contiguous); output NHWC fp32 (pair_out == 0) or pair rows
extern "C" int ovis_roi_align_forward_strided_from_nhwc_f32(const float* input_nhwc, const float* rois, void* output,
                                                            int num_rois, int batch, int channels, int height,
                                                            int width, int pooled_h, int pooled_w, int bin_stride,
                                                            float spatial_scale, int sampling_ratio, int pair_out,
                                                            void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0 ||
      bin_stride <= 0)
    return OVIS_EINVAL;
  if (num_rois == 0 || channels == 0) return OVIS_OK;
  if (!input_nhwc || !rois || !output) return OVIS_EINVAL;
  if (channels % 4 != 0 || (pair_out && channels % 32 != 0) || ((uintptr_t)input_nhwc & 15) || ((uintptr_t)output & 15))
    return OVIS_ERANGE;
  const int oh = (pooled_h + bin_stride - 1) / bin_stride, ow = (pooled_w + bin_stride - 1) / bin_stride;
  const long tiles = (long)(channels / kCPB) * num_rois;
  // The LDS form is launched with 40 KB of UNUSED dynamic LDS on top of its 17 KB window tile: two of its workgroups per CU
  // instead of nine.  Alone the kernel is latency-bound and likes the nine; in the two-stream training step its workgroups
  // take over the slots of the other stream's GEMM workgroups as those retire and then mostly wait on memory -- same-box
  // A/B of the pipelined student step: 34.0 ms unpadded (4 and 3 per CU: the same), 33.3 ms at two per CU, 34.0 ms at one,
  // 33.4 ms with the LDS-free direct form; the single-stream teacher step does not notice (24.2-24.5 ms all forms).
  // The pad applies to the PAIR-output form only: that is the pooler of frozen features, i.e. of the student-teacher step's
  // side-stream half; the fp32-output form serves trainable features (the teacher configuration's single-stream step), where
  // nothing competes for the CUs and nine workgroups per CU are simply faster (289 -> 180 us per launch at 1024 RoIs).
  const int kCoResidencyPadBytes = pair_out ? 40 * 1024 : 0;
  if (channels % kCPB == 0 && tiles <= 0x7fffffffL)
    hipLaunchKernelGGL(roi_align_fwd_nhwc_in_strided_lds_kernel, dim3((unsigned)tiles), dim3(kThreads), kCoResidencyPadBytes,
                       (hipStream_t)stream, input_nhwc, rois, (float*)output, num_rois, batch, channels, height, width,
                       pooled_h, pooled_w, bin_stride, oh, ow, spatial_scale, sampling_ratio, pair_out);
  else
    hipLaunchKernelGGL(roi_align_fwd_nhwc_in_strided_kernel, dim3((unsigned)num_rois), dim3(kThreads), 0, (hipStream_t)stream,
                       input_nhwc, rois, (float*)output, num_rois, batch, channels, height, width, pooled_h, pooled_w,
                       bin_stride, oh, ow, spatial_scale, sampling_ratio, pair_out);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

static int strided_nhwc_launch(const float* input, const float* rois, float* output, int num_rois, int batch,
                               int channels, int height, int width, int pooled_h, int pooled_w, int bin_stride,
                               float spatial_scale, int sampling_ratio, int pair_out, void* stream) {
  if (num_rois < 0 || batch < 0 || channels < 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0 ||
      bin_stride <= 0)
    return OVIS_EINVAL;
  if (num_rois == 0 || channels == 0) return OVIS_OK;
  if (!input || !rois || !output) return OVIS_EINVAL;
  const int oh = (pooled_h + bin_stride - 1) / bin_stride, ow = (pooled_w + bin_stride - 1) / bin_stride;
  const long blocks = (long)ovis_ceil_div(channels, kCPB) * num_rois;
  if (blocks > 0x7fffffffL) return OVIS_ERANGE;
  hipLaunchKernelGGL((roi_align_fwd_strided_nhwc_kernel<kFwdLdsFloats, false>), dim3((unsigned)blocks), dim3(kThreads), 0,
                     (hipStream_t)stream, input, rois, output, num_rois, batch, channels, height, width, pooled_h,
                     pooled_w, bin_stride, oh, ow, spatial_scale, sampling_ratio, pair_out);
  OVIS_LAUNCH_CHECK();
  if ((long)height * width > kSmallWindow) {  // larger windows can exist at all
    hipLaunchKernelGGL((roi_align_fwd_strided_nhwc_kernel<kBigLdsFloats, true>), dim3((unsigned)blocks), dim3(kThreads), 0,
                       (hipStream_t)stream, input, rois, output, num_rois, batch, channels, height, width, pooled_h,
                       pooled_w, bin_stride, oh, ow, spatial_scale, sampling_ratio, pair_out);
    OVIS_LAUNCH_CHECK();
  }
  return OVIS_OK;
}
