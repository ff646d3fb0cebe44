// Weight preparation of EVERY trainable convolution of a model in one launch, for gfx950 (MI355X).
// csrc/split_gemm.hip::weight_prep_pair_kernel turns one raw weight [N, C, T] into the two pair-layout operands the split
// GEMMs read -- the tap-major forward matrix and the transposed matrix of the data gradient, both times the folded FrozenBN
// scale -- once per convolution and training step: 42 launches of a few microseconds at the head of the teacher step's
// blocks, each in front of the GEMM that waits for it.  The weights only change in the optimizer step
// (maskrcnn_benchmark/engine/trainer.py:139 ``optimizer.step()``), so all of them are prepared HERE, in one launch behind the
// fused SGD launch (csrc/optim.hip), into buffers that live as long as the model: a table of items like the SGD kernel's,
// a table of (item, tile) per workgroup.  Per value the arithmetic and the byte layout are those of weight_prep_pair_kernel
// (the test compares the buffers bit for bit).
#include "ovis_common.h"

namespace {

struct PrepItem {
  const float* w;      // [N, C, T] raw weight (T = KH * KW taps, contiguous)
  const float* scale;  // [N] folded FrozenBN scale or null
  char* fwd;           // forward matrix: row n at fwd + n * fwd_row_bytes, k = t * C + c
  char* bwd;           // transposed matrix (or null): row c at bwd + c * bwd_row_bytes, k' = t * N + n
  long fwd_row_bytes;  // >= 4 * T * C: rows of a wider matrix ([w3 | wd] of a projection block) are written in place
  long bwd_row_bytes;
  int N, C, T, pad_;
};
static_assert(sizeof(PrepItem) == 64, "the host builds this table byte by byte");

typedef __bf16 prep_b2 __attribute__((ext_vector_type(2)));
typedef float prep_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned prep_pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE), as split_gemm.hip's pack
  prep_f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, prep_b2));
}

constexpr int kPrepThreads = 256, kPrepTile = 32, kPrepMaxTaps = 15;  // 32 x (32 T + 1) floats of LDS per workgroup

// One workgroup = one 32 (output channels) x 32 (input channels) x T tile of one weight: read ONCE, row by row (32 T
// contiguous floats per output channel: coalesced), scaled, parked in LDS; then written out twice -- per (n, t) the 32 input
// channels of the tile are exactly one 128-byte pair block of the forward matrix, per (c, t) the 32 output channels exactly one
// pair block of the transposed matrix -- 16-byte stores, four lanes per block.  (The per-convolution kernel reads the weight
// a second time for the transposed matrix, 4 bytes per lane at a stride of C T floats: 237 us for the teacher's 23 M
// weights in this form; the tile form moves 92 MB in and 184 MB out once.)
__global__ __launch_bounds__(kPrepThreads) void weight_prep_pair_multi_kernel(const PrepItem* __restrict__ items,
                                                                            const int2* __restrict__ blocks) {
  extern __shared__ float tile[];  // [32][32 T + 1]
  const int2 bc = blocks[blockIdx.x];
  const PrepItem it = items[bc.x];
  const int N = it.N, C = it.C, T = it.T;
  const int tiles_c = C / kPrepTile;
  const int n0 = (bc.y / tiles_c) * kPrepTile, c0 = (bc.y % tiles_c) * kPrepTile;
  const int row = kPrepTile * T, S = row + 1;
  const int row4 = row >> 2;  // 32 T floats per output channel = 8 T 16-byte loads (c0 * T * 4 bytes is a multiple of 128)
  for (int idx = threadIdx.x; idx < kPrepTile * row4; idx += kPrepThreads) {
    const int r = idx / row4, j = (idx - r * row4) * 4;
    const float sc = it.scale ? it.scale[n0 + r] : 1.f;
    const float4 v = *(const float4*)(it.w + ((long)(n0 + r) * C + c0) * T + j);
    float* d = tile + r * S + j;
    d[0] = v.x * sc; d[1] = v.y * sc; d[2] = v.z * sc; d[3] = v.w * sc;
  }
  __syncthreads();
  const int tasks = kPrepTile * T * 4;
  for (int task = threadIdx.x; task < 2 * tasks; task += kPrepThreads) {
    const bool fwd = task < tasks;
    if (!fwd && !it.bwd) break;
    const int tk = fwd ? task : task - tasks;
    const int q = tk & 3, rt = tk >> 2;
    const int t = rt % T, r = rt / T;   // r: the tile's output channel (forward) / input channel (transposed)
    float v[8];
    char* d;
    if (fwd) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[r * S + (q * 8 + j) * T + t];
      const int k = t * C + c0 + q * 8;
      d = it.fwd + (long)(n0 + r) * it.fwd_row_bytes + (long)(k >> 5) * 128 + (k & 31) * 2;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[(q * 8 + j) * S + r * T + t];
      const int k = t * N + n0 + q * 8;
      d = it.bwd + (long)(c0 + r) * it.bwd_row_bytes + (long)(k >> 5) * 128 + (k & 31) * 2;
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = prep_pack_bf16(v[2 * j], v[2 * j + 1]);
      l[j] = prep_pack_bf16(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u));
    }
    *(uint4*)d = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(d + 64) = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

}  // namespace

extern "C" int ovis_weight_prep_pair_multi_f32(const void* items, const void* blocks, int num_blocks, int max_taps,
                                               void* stream) {
  if (num_blocks < 0 || max_taps <= 0) return OVIS_EINVAL;
  if (num_blocks == 0) return OVIS_OK;
  if (!items || !blocks) return OVIS_EINVAL;
  if (max_taps > kPrepMaxTaps) return OVIS_ERANGE;
  const size_t lds = sizeof(float) * kPrepTile * (kPrepTile * (size_t)max_taps + 1);
  hipLaunchKernelGGL(weight_prep_pair_multi_kernel, dim3((unsigned)num_blocks), dim3(kPrepThreads), lds, (hipStream_t)stream,
                     (const PrepItem*)items, (const int2*)blocks);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_weight_prep_tile(void) { return kPrepTile; }
