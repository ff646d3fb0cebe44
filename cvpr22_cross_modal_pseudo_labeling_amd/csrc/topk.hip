// Sorted top-k of every row of a score matrix for gfx950 (MI355X): the `objectness.topk(pre_nms_top_n, dim=1, sorted=True)`
// of maskrcnn_benchmark/modeling/rpn/inference.py:95 (RPNPostProcessor.forward_for_single_feature_map).  The tensor
// library's route for k = 12000 of 63000 is a multi-block radix select, a gather and a merge sort of the selection:
// ~80 launches of a few microseconds each, 0.4 ms of a chain that holds the stream in front of the NMS.  Here: ONE device
// radix sort of the whole batch (rocPRIM, ROCm's primitive library) on 64-bit keys
//     (num_rows - 1 - row) << 32 | monotone(score)
// in descending order -- rows come out in order, each sorted by descending score -- between a key-building launch and a
// launch that writes the first k entries of every row.  The sort is stable and the payload is the flat index, so equal
// scores keep ASCENDING index order (torch.topk leaves the order of ties unspecified); NaNs sort as the largest values, as
// torch's do.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "ovis_common.h"

namespace {

// float -> unsigned with the order torch sorts by: -0 == +0 (the add turns -0 into +0), every NaN above +inf
__device__ __forceinline__ unsigned monotone_bits(float v) {
  if (v != v) return 0xffffffffu;
  const unsigned b = __float_as_uint(v + 0.f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void topk_keys_kernel(const float* __restrict__ scores, long row_stride, int row_len,
                                                        int num_rows, unsigned long long* __restrict__ keys,
                                                        unsigned* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (i >= row_len) return;
  const size_t flat = (size_t)row * row_len + i;
  keys[flat] = ((unsigned long long)(unsigned)(num_rows - 1 - row) << 32) | monotone_bits(scores[(size_t)row * row_stride + i]);
  vals[flat] = (unsigned)flat;
}

__global__ __launch_bounds__(256) void topk_emit_kernel(const float* __restrict__ scores, long row_stride, int row_len, int k,
                                                        const unsigned* __restrict__ sorted_vals,
                                                        float* __restrict__ out_scores, long long* __restrict__ out_idx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (j >= k) return;
  const unsigned flat = sorted_vals[(size_t)row * row_len + j];
  const int i = (int)(flat - (unsigned)row * (unsigned)row_len);
  out_scores[(size_t)row * k + j] = scores[(size_t)row * row_stride + i];
  out_idx[(size_t)row * k + j] = i;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct TopkLayout {
  size_t keys_in, keys_out, vals_in, vals_out, temp, temp_bytes, total;
};

int key_bits(int num_rows) {
  int b = 0;
  while ((1 << b) < num_rows) ++b;
  return 32 + b;
}

int topk_layout(int num_rows, int row_len, TopkLayout* L) {
  const size_t n = (size_t)num_rows * row_len;
  size_t temp = 0;
  hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, temp, (const unsigned long long*)nullptr,
                                                (unsigned long long*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr, n,
                                                0u, (unsigned)key_bits(num_rows));
  if (e != hipSuccess) return (int)e;
  size_t off = 0;
  L->keys_in = off;  off = align256(off + 8 * n);
  L->keys_out = off; off = align256(off + 8 * n);
  L->vals_in = off;  off = align256(off + 4 * n);
  L->vals_out = off; off = align256(off + 4 * n);
  L->temp = off;     off = align256(off + temp);
  L->temp_bytes = temp;
  L->total = off;
  return OVIS_OK;
}

}  // namespace

extern "C" size_t ovis_topk_sorted_workspace_bytes(int num_rows, int row_len) {
  if (num_rows <= 0 || row_len <= 0 || (size_t)num_rows * row_len > 0x7fffffffull) return 0;
  TopkLayout L;
  if (topk_layout(num_rows, row_len, &L) != OVIS_OK) return 0;
  return L.total;
}

extern "C" int ovis_topk_sorted_f32(const float* scores, long row_stride, int num_rows, int row_len, int k,
                                    float* out_scores, int64_t* out_idx, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  if (num_rows < 0 || row_len < 0 || k < 0 || k > row_len) return OVIS_EINVAL;
  if (num_rows == 0 || k == 0) return OVIS_OK;
  if (!scores || !out_scores || !out_idx || !workspace || row_stride < row_len) return OVIS_EINVAL;
  if (num_rows > 65535 || (size_t)num_rows * row_len > 0x7fffffffull) return OVIS_ERANGE;
  TopkLayout L;
  int rc = topk_layout(num_rows, row_len, &L);
  if (rc != OVIS_OK) return rc;
  if (workspace_bytes < L.total) return OVIS_ENOSPC;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  unsigned long long* keys_in = (unsigned long long*)(ws + L.keys_in);
  unsigned long long* keys_out = (unsigned long long*)(ws + L.keys_out);
  unsigned* vals_in = (unsigned*)(ws + L.vals_in);
  unsigned* vals_out = (unsigned*)(ws + L.vals_out);
  hipLaunchKernelGGL(topk_keys_kernel, dim3(ovis_ceil_div(row_len, 256), num_rows), dim3(256), 0, s, scores, row_stride,
                     row_len, num_rows, keys_in, vals_in);
  OVIS_LAUNCH_CHECK();
  size_t temp = L.temp_bytes;
  OVIS_HIP_TRY(rocprim::radix_sort_pairs_desc((void*)(ws + L.temp), temp, (const unsigned long long*)keys_in, keys_out,
                                              (const unsigned*)vals_in, vals_out, (size_t)num_rows * row_len, 0u,
                                              (unsigned)key_bits(num_rows), s));
  hipLaunchKernelGGL(topk_emit_kernel, dim3(ovis_ceil_div(k, 256), num_rows), dim3(256), 0, s, scores, row_stride, row_len, k,
                     (const unsigned*)vals_out, out_scores, (long long*)out_idx);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
