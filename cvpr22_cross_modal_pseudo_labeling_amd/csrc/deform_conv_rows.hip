// Deformable convolution BACKWARD on NHWC rows for gfx950 (MI355X): no column buffer in the reference's layout, both
// contractions on the pair-layout split GEMM (csrc/split_gemm.hip).
//
// Reference: maskrcnn_benchmark/csrc/cuda/deform_conv_cuda.cu:271-383 (input / offset gradients), :385-497 (weight gradient),
// :580-694 (modulated); kernels deform_conv_kernel_cuda.cu:287-342 (col2im), :381-443 (col2im_coord), :643-774 (modulated).
// There the column buffer [C*KH*KW, step*Ho*Wo] sits between three fp32 GEMMs and three scatter / gather kernels.  Here
// rows m = (image, h_out, w_out), k = (tap, channel) -- the layout of the forward's implicit GEMM -- and
//
//   dcol[m, (t, c)] = sum_n dY[m, n] W[n, c, t]            one NT split GEMM (host: ovis_split_gemm_pair), fp32 rows
//   dX, dOffset, dMask  <-  dcn_col2im_rows_kernel          ONE pass over dcol: per (m, t) the sample geometry once, lanes
//                                                           along the channels: 4 coalesced atomic adds per channel into
//                                                           dX (NHWC), the four cells' values gathered for the coordinate /
//                                                           mask gradients, their channel sums reduced in the wave
//   col[m, (t, c)] in PAIR layout  <-  dcn_im2col_pair_rows_kernel   (sampled rows, bf16 hi | lo, written once)
//   dW[n, (t, c)] = sum_m dY[m, n] col[m, (t, c)]           one TN split GEMM (host: ovis_split_gemm_pair_tn)
//
// The per-element arithmetic is the reference's: the bilinear sample with zero padding (four terms in its order), its
// derivative w.r.t. the sampling position (get_coordinate_weight, :125-195) and the scatter weights (get_gradient_weight).
// Sums over channels and over samples that hit one input cell are fp32 atomics / wave reductions: their order is not fixed
// (the reference's col2im is atomic as well), results differ from run to run at rounding level.
#include "ovis_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct RowsGeom {
  int B, C, H, W, KH, KW, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, Ho, Wo;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

// Sample geometry of (row m, tap t, deformable group grp): position, validity, the four cells (element offsets into the
// NHWC image, -1 = outside) and the mask value.
struct Sample {
  float h, w, m;
  bool inside;
  int hl, wl;
  int o[4];
};

__device__ __forceinline__ Sample make_sample(const RowsGeom& g, const float* __restrict__ offset,
                                              const float* __restrict__ mask, int row, int t, int grp) {
  Sample s;
  const int wo = row % g.Wo, t1 = row / g.Wo;
  const int ho = t1 % g.Ho, b = t1 / g.Ho;
  const long plane = (long)g.Ho * g.Wo, pix = (long)ho * g.Wo + wo;
  const int T = g.KH * g.KW;
  const int ky = t / g.KW, kx = t - ky * g.KW;
  const float* off = offset + (((long)b * g.dg + grp) * 2 * T + 2 * t) * plane + pix;
  s.h = (float)(ho * g.stride_h - g.pad_h + ky * g.dil_h) + off[0];
  s.w = (float)(wo * g.stride_w - g.pad_w + kx * g.dil_w) + off[plane];
  s.m = mask ? mask[(((long)b * g.dg + grp) * T + t) * plane + pix] : 1.f;
  s.inside = s.h > -1.f && s.w > -1.f && s.h < (float)g.H && s.w < (float)g.W;
  s.hl = (int)floorf(s.h);
  s.wl = (int)floorf(s.w);
  const int hh = s.hl + 1, wh = s.wl + 1;
  const int img = b * g.H * g.W;
  s.o[0] = (s.inside && s.hl >= 0 && s.wl >= 0) ? (img + s.hl * g.W + s.wl) * g.C : -1;
  s.o[1] = (s.inside && s.hl >= 0 && wh <= g.W - 1) ? (img + s.hl * g.W + wh) * g.C : -1;
  s.o[2] = (s.inside && hh <= g.H - 1 && s.wl >= 0) ? (img + hh * g.W + s.wl) * g.C : -1;
  s.o[3] = (s.inside && hh <= g.H - 1 && wh <= g.W - 1) ? (img + hh * g.W + wh) * g.C : -1;
  return s;
}

// ---- sampled rows in pair layout: col[m, (t, c)] = mask * bilinear(x, position(m, t)) --------------------------------------
// One workgroup per row m; a thread handles 4 consecutive channels of one tap per pass (a wave = 256 channels of a tap:
// whole 128-byte lines of every cell).  Row m of the result: T * C / 32 blocks of [hi(32 x bf16) | lo(32 x bf16)].
__global__ __launch_bounds__(256) void dcn_im2col_pair_rows_kernel(const float* __restrict__ x_nhwc,
                                                                  const float* __restrict__ offset,
                                                                  const float* __restrict__ mask, char* __restrict__ colp,
                                                                  long colp_row_bytes, RowsGeom g) {
  const int row = blockIdx.x;
  const int T = g.KH * g.KW, c4n = g.C >> 2, cpg = g.C / g.dg;
  char* dst_row = colp + (long)row * colp_row_bytes;
  for (int i = threadIdx.x; i < T * c4n; i += 256) {
    const int t = i / c4n, c = (i - t * c4n) * 4;
    const Sample s = make_sample(g, offset, mask, row, t, c / cpg);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s.inside) {
      const float lh = s.h - s.hl, lw = s.w - s.wl, uh = 1.f - lh, uw = 1.f - lw;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 v1 = s.o[0] >= 0 ? *(const f32x4*)(x_nhwc + s.o[0] + c) : z;
      const f32x4 v2 = s.o[1] >= 0 ? *(const f32x4*)(x_nhwc + s.o[1] + c) : z;
      const f32x4 v3 = s.o[2] >= 0 ? *(const f32x4*)(x_nhwc + s.o[2] + c) : z;
      const f32x4 v4 = s.o[3] >= 0 ? *(const f32x4*)(x_nhwc + s.o[3] + c) : z;
      v = (uh * uw) * v1 + (uh * lw) * v2 + (lh * uw) * v3 + (lh * lw) * v4;
    }
    if (mask) v *= s.m;
    const unsigned h0 = pack_bf16(v.x, v.y), h1 = pack_bf16(v.z, v.w);
    const unsigned l0 = pack_bf16(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
    const unsigned l1 = pack_bf16(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
    const int k = t * g.C + c;
    char* d = dst_row + (long)(k >> 5) * 128 + (k & 31) * 2;
    *(uint2*)d = make_uint2(h0, h1);
    *(uint2*)(d + 64) = make_uint2(l0, l1);
  }
}

// ---- dcol -> dX (NHWC, atomics), dOffset, dMask ------------------------------------------------------------------------------
// Same thread mapping.  Eight neighbouring lanes (32 channels) always belong to one (tap, deformable group): their
// coordinate / mask sums are combined with three xor-shuffles before one lane adds them to the (zero-filled) outputs.
__global__ __launch_bounds__(256) void dcn_col2im_rows_kernel(const float* __restrict__ dcol, long dcol_ld,
                                                             const float* __restrict__ x_nhwc,
                                                             const float* __restrict__ offset,
                                                             const float* __restrict__ mask, float* __restrict__ dx_nhwc,
                                                             float* __restrict__ grad_offset,
                                                             float* __restrict__ grad_mask, RowsGeom g) {
  const int row = blockIdx.x;
  const int T = g.KH * g.KW, c4n = g.C >> 2, cpg = g.C / g.dg;
  const int wo = row % g.Wo, t1 = row / g.Wo;
  const int ho = t1 % g.Ho, b = t1 / g.Ho;
  const long plane = (long)g.Ho * g.Wo, pix = (long)ho * g.Wo + wo;
  const float* drow = dcol + (long)row * dcol_ld;
  const int total = T * c4n;
  // every lane of a wave runs the same number of passes (the shuffles need all 64 lanes): out-of-range lanes add zeros
  for (int i0 = 0; i0 < total; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool live = i < total;
    const int ii = live ? i : total - 1;
    const int t = ii / c4n, c = (ii - t * c4n) * 4;
    const int grp = c / cpg;
    const Sample s = make_sample(g, offset, mask, row, t, grp);
    float gh = 0.f, gw = 0.f, gm = 0.f;
    if (live && s.inside) {
      const f32x4 cv = *(const f32x4*)(drow + (long)t * g.C + c);
      const f32x4 top = mask ? cv * s.m : cv;
      const float lh = s.h - s.hl, lw = s.w - s.wl, uh = 1.f - lh, uw = 1.f - lw;   // uh = hl + 1 - h, uw = wl + 1 - w
      const float wgt[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
      // d(sample)/dh and d/dw per cell (get_coordinate_weight): -uw, -lw, +uw, +lw   and   -uh, +uh, -lh, +lh
      const float dh[4] = {-uw, -lw, uw, lw}, dw[4] = {-uh, uh, -lh, lh};
      f32x4 ch = {0.f, 0.f, 0.f, 0.f}, cw = ch, val = ch;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (s.o[q] < 0) continue;
        if (dx_nhwc) {
          float* p = dx_nhwc + s.o[q] + c;
          atomicAdd(p + 0, wgt[q] * top.x);
          atomicAdd(p + 1, wgt[q] * top.y);
          atomicAdd(p + 2, wgt[q] * top.z);
          atomicAdd(p + 3, wgt[q] * top.w);
        }
        if (grad_offset) {
          const f32x4 xv = *(const f32x4*)(x_nhwc + s.o[q] + c);
          ch += dh[q] * xv;
          cw += dw[q] * xv;
          val += wgt[q] * xv;
        }
      }
      if (grad_offset) {
        const f32x4 a = ch * cv, bq = cw * cv, mv = val * cv;
        gh = (a.x + a.y + a.z + a.w) * s.m;
        gw = (bq.x + bq.y + bq.z + bq.w) * s.m;
        gm = mv.x + mv.y + mv.z + mv.w;
      }
    }
    if (grad_offset) {
#pragma unroll
      for (int d = 1; d < 8; d <<= 1) {
        gh += __shfl_xor(gh, d, 64);
        gw += __shfl_xor(gw, d, 64);
        gm += __shfl_xor(gm, d, 64);
      }
      if (live && (threadIdx.x & 7) == 0) {
        float* go = grad_offset + (((long)b * g.dg + grp) * 2 * T + 2 * t) * plane + pix;
        atomicAdd(go, gh);
        atomicAdd(go + plane, gw);
        if (grad_mask) atomicAdd(grad_mask + (((long)b * g.dg + grp) * T + t) * plane + pix, gm);
      }
    }
  }
}

// WIDE form of the pass above for C % 256 == 0 and (C / dg) % 256 == 0: a wave's 256 channels then belong to ONE (tap, group),
// so its sample geometry is wave-uniform and the scatter can be issued lane-contiguously -- the lane's four consecutive
// channels go through a wave-private LDS transpose (one 16-byte write, four 4-byte reads) and atomic j of a cell covers
// channels 64 j .. 64 j + 63 (256 contiguous bytes = 2 lines per instruction; with four consecutive channels per lane every
// instruction touched 8 lines, and the pass ran at 74 G lane-atomics/s against the 330 G/s a contiguous stream reaches,
// tools/microbench/global_atomic_bench.hip).  The coordinate / mask sums are reduced over the whole wave.
__global__ __launch_bounds__(256) void dcn_col2im_rows_wide_kernel(const float* __restrict__ dcol, long dcol_ld,
                                                                  const float* __restrict__ x_nhwc,
                                                                  const float* __restrict__ offset,
                                                                  const float* __restrict__ mask, float* __restrict__ dx_nhwc,
                                                                  float* __restrict__ grad_offset,
                                                                  float* __restrict__ grad_mask, RowsGeom g) {
  __shared__ f32x4 xpose[256];
  const int row = blockIdx.x;
  const int T = g.KH * g.KW, c4n = g.C >> 2, cpg = g.C / g.dg;
  const int wo = row % g.Wo, t1 = row / g.Wo;
  const int ho = t1 % g.Ho, b = t1 / g.Ho;
  const long plane = (long)g.Ho * g.Wo, pix = (long)ho * g.Wo + wo;
  const float* drow = dcol + (long)row * dcol_ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int total = T * c4n;   // a multiple of 64: every wave of a pass is wholly inside or wholly outside
  for (int i0 = wave * 64; i0 < total; i0 += 256) {
    const int t = i0 / c4n, c0 = (i0 - t * c4n) * 4;   // the wave's tap and its first channel
    const int grp = c0 / cpg;
    const Sample s = make_sample(g, offset, mask, row, t, grp);
    if (!s.inside) continue;   // wave-uniform
    const int c = c0 + lane * 4;
    const f32x4 cv = *(const f32x4*)(drow + (long)t * g.C + c);
    const f32x4 top = mask ? cv * s.m : cv;
    const float lh = s.h - s.hl, lw = s.w - s.wl, uh = 1.f - lh, uw = 1.f - lw;
    const float wgt[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
    if (dx_nhwc) {
      xpose[wave * 64 + lane] = top;                       // float index 4 * lane + e = channel c0 + 4 * lane + e
      const float* tp = (const float*)(xpose + wave * 64);  // (wave-private: no barrier, the wave runs in lockstep)
      float tt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) tt[j] = tp[64 * j + lane];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (s.o[q] < 0) continue;
        float* p = dx_nhwc + s.o[q] + c0 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(p + 64 * j, wgt[q] * tt[j]);
      }
    }
    if (grad_offset) {
      const float dh[4] = {-uw, -lw, uw, lw}, dw[4] = {-uh, uh, -lh, lh};
      f32x4 ch = {0.f, 0.f, 0.f, 0.f}, cw = ch, val = ch;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (s.o[q] < 0) continue;
        const f32x4 xv = *(const f32x4*)(x_nhwc + s.o[q] + c);
        ch += dh[q] * xv;
        cw += dw[q] * xv;
        val += wgt[q] * xv;
      }
      const f32x4 a = ch * cv, bq = cw * cv, mv = val * cv;
      float gh = (a.x + a.y + a.z + a.w) * s.m, gw = (bq.x + bq.y + bq.z + bq.w) * s.m, gm = mv.x + mv.y + mv.z + mv.w;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        gh += __shfl_xor(gh, d, 64);
        gw += __shfl_xor(gw, d, 64);
        gm += __shfl_xor(gm, d, 64);
      }
      if (lane == 0) {
        float* go = grad_offset + (((long)b * g.dg + grp) * 2 * T + 2 * t) * plane + pix;
        atomicAdd(go, gh);
        atomicAdd(go + plane, gw);
        if (grad_mask) atomicAdd(grad_mask + (((long)b * g.dg + grp) * T + t) * plane + pix, gm);
      }
    }
  }
}

int rows_check(const RowsGeom& g) {
  if (g.B <= 0 || g.C <= 0 || g.H <= 0 || g.W <= 0 || g.KH <= 0 || g.KW <= 0 || g.stride_h <= 0 || g.stride_w <= 0 ||
      g.dil_h <= 0 || g.dil_w <= 0 || g.dg <= 0 || g.Ho <= 0 || g.Wo <= 0)
    return OVIS_EINVAL;
  if (g.C % g.dg != 0 || (g.C / g.dg) % 32 != 0) return OVIS_ERANGE;
  if ((double)g.B * g.H * g.W * g.C >= 2147483648.0 || (double)g.B * g.Ho * g.Wo >= 2147483648.0) return OVIS_ERANGE;
  return OVIS_OK;
}

}  // namespace

extern "C" int ovis_deform_im2col_pair_rows_f32(const float* input_nhwc, const float* offset, const float* mask,
                                                void* columns_pair, long columns_row_bytes, int batch, int channels,
                                                int height, int width, int out_h, int out_w, int kernel_h, int kernel_w,
                                                int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                                int deformable_group, void* stream) {
  const RowsGeom g{batch, channels, height, width, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                   deformable_group, out_h, out_w};
  if (batch == 0) return OVIS_OK;
  if (const int rc = rows_check(g)) return rc;
  if (!input_nhwc || !offset || !columns_pair) return OVIS_EINVAL;
  if (((uintptr_t)input_nhwc & 15) || ((uintptr_t)columns_pair & 15) || columns_row_bytes % 16 ||
      columns_row_bytes < 4L * kernel_h * kernel_w * channels)
    return OVIS_EINVAL;
  const long rows = (long)batch * out_h * out_w;
  hipLaunchKernelGGL(dcn_im2col_pair_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, input_nhwc, offset,
                     mask, (char*)columns_pair, columns_row_bytes, g);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_deform_col2im_rows_f32(const float* dcol_rows, long dcol_ld, const float* input_nhwc,
                                           const float* offset, const float* mask, float* grad_input_nhwc,
                                           float* grad_offset, float* grad_mask, int batch, int channels, int height,
                                           int width, int out_h, int out_w, int kernel_h, int kernel_w, int stride_h,
                                           int stride_w, int pad_h, int pad_w, int dil_h, int dil_w, int deformable_group,
                                           void* stream) {
  const RowsGeom g{batch, channels, height, width, kernel_h, kernel_w, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                   deformable_group, out_h, out_w};
  if (batch == 0) return OVIS_OK;
  if (const int rc = rows_check(g)) return rc;
  if (!dcol_rows || !offset || (!grad_input_nhwc && !grad_offset) || (grad_offset && !input_nhwc) ||
      (grad_mask && (!mask || !grad_offset)))
    return OVIS_EINVAL;
  if (((uintptr_t)dcol_rows & 15) || ((uintptr_t)input_nhwc & 15) || dcol_ld % 4 ||
      dcol_ld < (long)kernel_h * kernel_w * channels)
    return OVIS_EINVAL;
  const long rows = (long)batch * out_h * out_w;
  if (channels % 256 == 0 && (channels / deformable_group) % 256 == 0)
    hipLaunchKernelGGL(dcn_col2im_rows_wide_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dcol_rows,
                       dcol_ld, input_nhwc, offset, mask, grad_input_nhwc, grad_offset, grad_mask, g);
  else
    hipLaunchKernelGGL(dcn_col2im_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dcol_rows, dcol_ld,
                       input_nhwc, offset, mask, grad_input_nhwc, grad_offset, grad_mask, g);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
