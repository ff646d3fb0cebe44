// Pieces shared by the matrix-core RoIAlign kernels (roi_align_bwd_plane.hip, roi_align_fwd_mfma.hip), gfx950.
//
// Bilinear average pooling is separable: out_c = Ay . win_c . Ax^T, with Ax[j][x] the summed column weights of
// bin-column j's samples on feature column x and Ay[i][y] likewise for rows (x 1/count).  Both factors depend on
// the RoI only, so a plan kernel evaluates them once per RoI, cut into 16-cell blocks per axis and stored in the
// lane layout of a v_mfma_f32_16x16x16_bf16 operand, split into bf16 hi + lo parts: hi + lo carries 16 mantissa
// bits and the three products hi.hi + hi.lo + lo.hi are accumulated in fp32 by the matrix core (error ~1e-5
// relative per term; the matrix pipe runs them 16x faster than an f32-input MFMA).
#pragma once
#include "roi_geom.h"

namespace ovis_roi {

typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int kT = 16;        // window block edge == MFMA tile edge

static __device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));  // v_cvt_pk_bf16_f32 (RNE)
}

// v -> bf16 hi (returned .x,.y) and bf16 lo of the exact remainder (.z,.w); element e of the 4-vector
// sits in half (e & 1) of word (e >> 1), which is the k-order of an MFMA 16x16x16 operand.
static __device__ __forceinline__ u4 split_bf16(f4 v) {
  const unsigned h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
  const float r0 = v.x - __uint_as_float(h01 << 16), r1 = v.y - __uint_as_float(h01 & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(h23 << 16), r3 = v.w - __uint_as_float(h23 & 0xffff0000u);
  return (u4){h01, h23, pack_bf16(r0, r1), pack_bf16(r2, r3)};
}

static __device__ __forceinline__ s4 as_s4(unsigned a, unsigned b) {
  u2 u = {a, b};
  return __builtin_bit_cast(s4, u);
}

// D += (Ahi + Alo) . (Bhi + Blo) without the lo.lo term
static __device__ __forceinline__ f4 mfma3(u4 a, u4 b, f4 acc) {
  const s4 ah = as_s4(a.x, a.y), al = as_s4(a.z, a.w), bh = as_s4(b.x, b.y), bl = as_s4(b.z, b.w);
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);
  return acc;
}

// a0 . b0 + a1 . b1 as two K = 16 products into one accumulator (the pre-split small-tile form: the operands carry hi halves
// in k-groups 0, 1 and lo halves in groups 2, 3 on one side, hi / lo halves on the other: all four hi/lo terms)
static __device__ __forceinline__ f4 mfma2x2(u2 a0, u2 a1, u2 b0, u2 b1) {
  f4 acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as_s4(a0.x, a0.y), as_s4(b0.x, b0.y), (f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as_s4(a1.x, a1.y), as_s4(b1.x, b1.y), acc, 0, 0, 0);
}

// Summed weight the samples of bin `p` put on feature cell `cell` along one axis (the reference's
// bilinear_interpolate_gradient set-up, ROIAlign_cuda.cu:125-175, reduced to one axis).
static __device__ __forceinline__ float axis_weight(float start, float bin, int grid, int p, int size, int cell) {
  float acc = 0.f;
  const float step = bin / (float)grid;  // one division per call instead of one per sample (<= 1 ulp on the coordinate)
  const float base = start + (float)p * bin;
  for (int i = 0; i < grid; ++i) {
    int lo, hi;
    float l, h;
    if (!axis_sample(base + ((float)i + .5f) * step, size, lo, hi, l, h)) continue;
    acc += (lo == cell ? h : 0.f) + (hi == cell ? l : 0.f);
  }
  return acc;
}


static __device__ __forceinline__ int block_origin(int w0, int blk, int size) {
  // a block's 16-cell footprint starts at its first owned cell, pulled back so that it lies inside the map
  return max(min(w0 + blk * kT, size - kT), 0);
}

}  // namespace ovis_roi
