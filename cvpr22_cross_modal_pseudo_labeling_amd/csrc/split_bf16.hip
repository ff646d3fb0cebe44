// fp32 -> bf16 hi/lo operand split for fp32-accurate GEMMs on the bf16 matrix pipe (gfx950).
//
// x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi).  A product of two split operands keeps
// hi.hi + hi.lo + lo.hi (relative error ~4e-6 in practice, cf. 1.7e-6 for a plain fp32 GEMM) and runs three
// bf16 MFMA passes instead of one fp32 pass that is 16x slower.  To feed ONE plain bf16 GEMM, the three
// products are concatenated along K:
//     left  operand rows:  [ hi | hi | lo ]      (mode 0)
//     right operand rows:  [ hi | lo | hi ]      (mode 1)
// so that  sum_k' L[m,k'] R[n,k']  over the 3K columns is exactly the three-term product.  Used by the res5
// 1x1 convolutions (modeling/backbone.py), which are [R*49, Cin] x [Cout, Cin]^T GEMMs in NHWC.
//
// HBM-bound byte kernel: 4 B read + 6 B written per element, 16-byte loads / 8-byte stores, grid-stride.
#include "ovis_common.h"

namespace {
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

template <int MODE>
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ src, long src_rs,
                                                          unsigned short* __restrict__ dst, long rows, int cols) {
  const int qcols = cols >> 2;  // float4 groups per row
  const long total = rows * qcols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / qcols;
    const int c = (int)(i - r * qcols) * 4;
    const float4 v = *(const float4*)(src + r * src_rs + c);
    const unsigned h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
    const unsigned l01 = pack_bf16(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
    const unsigned l23 = pack_bf16(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
    unsigned short* d = dst + r * 3L * cols + c;
    const uint2 hi = make_uint2(h01, h23), lo = make_uint2(l01, l23);
    *(uint2*)d = hi;
    *(uint2*)(d + cols) = MODE == 0 ? hi : lo;
    *(uint2*)(d + 2 * cols) = MODE == 0 ? lo : hi;
  }
}
}  // namespace

extern "C" int ovis_split_bf16x3_f32(const float* src, long src_row_stride, void* dst_bf16, long rows, int cols,
                                     int mode, void* stream) {
  if (rows < 0 || cols < 0 || (mode != 0 && mode != 1)) return OVIS_EINVAL;
  if (rows == 0 || cols == 0) return OVIS_OK;
  if (!src || !dst_bf16) return OVIS_EINVAL;
  if (cols % 4 != 0 || src_row_stride % 4 != 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst_bf16 & 7))
    return OVIS_ERANGE;  // the caller falls back to the fp32 GEMM
  const long total = rows * (cols / 4);
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 8L * OVIS_NUM_CU ? blocks : 8L * OVIS_NUM_CU);
  if (mode == 0)
    hipLaunchKernelGGL(split_bf16x3_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_row_stride,
                       (unsigned short*)dst_bf16, rows, cols);
  else
    hipLaunchKernelGGL(split_bf16x3_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_row_stride,
                       (unsigned short*)dst_bf16, rows, cols);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fused GEMM epilogue for the NHWC res5 head: y = act(y + bias[col] (+ residual)) in place, one pass
// (the separate bias add, residual add and ReLU of the reference's Bottleneck.forward, resnet.py:323-344,
// are 3 read-modify-write passes over a [R*49, C] tensor).  16-byte accesses, grid-stride.
// ---------------------------------------------------------------------------------------------------
namespace {
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                      const float* __restrict__ res, long rows, int cols) {
  const int qcols = cols >> 2;
  const long total = rows * qcols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % qcols) * 4;
    float4 v = ((float4*)y)[i];
    if (bias) {
      const float4 b = *(const float4*)(bias + c);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (RES) {
      const float4 r = ((const float4*)res)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    ((float4*)y)[i] = v;
  }
}
}  // namespace

extern "C" int ovis_bias_act_f32(float* y, const float* bias, const float* residual, long rows, int cols, int relu,
                                 void* stream) {
  if (rows < 0 || cols < 0) return OVIS_EINVAL;
  if (rows == 0 || cols == 0) return OVIS_OK;
  if (!y) return OVIS_EINVAL;
  if (cols % 4 != 0 || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15)) return OVIS_ERANGE;
  const long total = rows * (cols / 4);
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 8L * OVIS_NUM_CU ? blocks : 8L * OVIS_NUM_CU);
  hipStream_t s = (hipStream_t)stream;
#define OVIS_BA(RES_, RELU_) \
  hipLaunchKernelGGL((bias_act_kernel<RES_, RELU_>), dim3(grid), dim3(256), 0, s, y, bias, residual, rows, cols)
  if (residual) { if (relu) OVIS_BA(true, true); else OVIS_BA(true, false); }
  else { if (relu) OVIS_BA(false, true); else OVIS_BA(false, false); }
#undef OVIS_BA
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

// ---------------------------------------------------------------------------------------------------
// Split + im2col for 3x3 (any odd KH x KW, stride 1, "same" zero padding) convolutions on NHWC tensors, so
// that the convolution and both of its gradients become plain bf16 GEMMs with fp32 accumulation:
//   src [R, H, W, C] f32  ->  dst [R*H*W, 3 * T*C] bf16,  T = KH*KW taps, row m = (r, y, x):
//       [ hi(tap 0) .. hi(tap T-1) | hi(tap 0) .. hi(tap T-1) | lo(tap 0) .. lo(tap T-1) ]
//   tap t = (ky, kx) reads pixel (y + ky - KH/2, x + kx - KW/2), zeros outside the map; with `flip` the taps
//   are taken in reverse order (the data-gradient convolution uses the 180-degree rotated kernel).
// Against a mode-1 split of the weight matrix [Cout, T*C] this is the three-term product of split_bf16x3.
// HBM-bound: every source element is read once and written 3 x T times as bf16 (27 x 2 B for a 3x3).
// ---------------------------------------------------------------------------------------------------
namespace {
// Scatter form: a thread owns (source pixel, V channels), reads them ONCE, splits them once and writes the result
// into the T rows whose neighbourhood contains that pixel (row of pixel p - offset(t), tap t); the taps of its own
// row that fall outside the map are zero-filled by the same thread.  HBM traffic = source once + rows once (the
// gather form re-read every source element T times and fetched 3.3x the source from beyond L2).
template <int V>  // channels per thread: 8 (16-byte stores) when channels % 8 == 0, else 4
__global__ __launch_bounds__(256) void im2col_split_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                          long pixels, int H, int W, int C, int KH, int KW, int flip) {
  const int qc = C / V;
  const int T = KH * KW;
  const long total = pixels * qc;
  const long TC = (long)T * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % qc) * V;
    const long m = i / qc;
    const int x = (int)(m % W);
    const int y = (int)((m / W) % H);
    unsigned hi[V / 2], lo[V / 2];
    const float* p = src + m * C + c;
#pragma unroll
    for (int k = 0; k < V / 4; ++k) {
      const float4 v = *(const float4*)(p + 4 * k);
      const unsigned h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
      hi[2 * k] = h01;
      hi[2 * k + 1] = h23;
      lo[2 * k] = pack_bf16(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
      lo[2 * k + 1] = pack_bf16(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
    }
    for (int t = 0; t < T; ++t) {
      const int ts = flip ? T - 1 - t : t;
      const int dy = ts / KW - KH / 2, dx = ts % KW - KW / 2;  // row (y', x') reads pixel (y' + dy, x' + dx) at tap t
      unsigned short* own = dst + m * 3 * TC + (long)t * C + c;
      if (y + dy < 0 || y + dy >= H || x + dx < 0 || x + dx >= W) {  // this row's tap t is outside the map: zeros
        if (V == 8) {
          const uint4 z = make_uint4(0u, 0u, 0u, 0u);
          *(uint4*)own = z; *(uint4*)(own + TC) = z; *(uint4*)(own + 2 * TC) = z;
        } else {
          const uint2 z = make_uint2(0u, 0u);
          *(uint2*)own = z; *(uint2*)(own + TC) = z; *(uint2*)(own + 2 * TC) = z;
        }
      }
      const int yr = y - dy, xr = x - dx;  // the row whose tap t is this pixel
      if (yr >= 0 && yr < H && xr >= 0 && xr < W) {
        unsigned short* d = dst + (m - (long)dy * W - dx) * 3 * TC + (long)t * C + c;
        if (V == 8) {
          const uint4 h = make_uint4(hi[0], hi[1], hi[2], hi[3]), l = make_uint4(lo[0], lo[1], lo[2], lo[3]);
          *(uint4*)d = h; *(uint4*)(d + TC) = h; *(uint4*)(d + 2 * TC) = l;
        } else {
          const uint2 h = make_uint2(hi[0], hi[1]), l = make_uint2(lo[0], lo[1]);
          *(uint2*)d = h; *(uint2*)(d + TC) = h; *(uint2*)(d + 2 * TC) = l;
        }
      }
    }
  }
}
}  // namespace

extern "C" int ovis_im2col_split_bf16x3_f32(const float* src, void* dst_bf16, long num, int height, int width,
                                            int channels, int kh, int kw, int flip, void* stream) {
  if (num < 0 || height <= 0 || width <= 0 || channels < 0 || kh <= 0 || kw <= 0 || !(kh & 1) || !(kw & 1))
    return OVIS_EINVAL;
  if (num == 0 || channels == 0) return OVIS_OK;
  if (!src || !dst_bf16) return OVIS_EINVAL;
  if (channels % 4 != 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst_bf16 & 7)) return OVIS_ERANGE;
  const long pixels = num * height * width;
  const bool wide = channels % 8 == 0 && ((uintptr_t)dst_bf16 & 15) == 0;
  const long total = pixels * (channels / (wide ? 8 : 4));
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 16L * OVIS_NUM_CU ? blocks : 16L * OVIS_NUM_CU);
  if (wide)
    hipLaunchKernelGGL(im2col_split_kernel<8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst_bf16, pixels, height, width, channels, kh, kw, flip);
  else
    hipLaunchKernelGGL(im2col_split_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst_bf16, pixels, height, width, channels, kh, kw, flip);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
