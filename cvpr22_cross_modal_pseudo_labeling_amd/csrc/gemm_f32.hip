// fp32 GEMM on the CDNA4 matrix cores for the cross-modal head (region embedding x text embedding).
//
// Reference semantics: maskrcnn_benchmark/modeling/roi_heads/box_head/roi_box_predictors.py:66-71 --
//   cls_emb = emb_pred(x)                (Linear 2048 -> 768, fp32)
//   cls_logit = einsum('pe,ce->pc', cls_emb, cls_score)
//   bbox_pred = Linear 2048 -> 8
// and their autograd transposes.  All are C[m,n] = sum_k A(m,k) * B(n,k) (+ bias[n]) with different operand
// strides, so ONE kernel with strided operands serves forward (NT) and both backward products (NN, TN).
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (one rounding per product, an fmaf chain in k order), so the result
// differs from a cuBLAS/ATen fp32 GEMM only by summation order.  Tile: 64x64 per 256-lane workgroup, each of
// the 4 waves owns one 32x32 accumulator (16 VGPRs); K is staged through a double-buffered LDS tile 16 at a time
// (registers -> LDS, the next tile's global loads in flight during the MFMAs, one barrier per step), k-major so
// the per-lane MFMA operand reads (A[i=lane&31][k=lane>>5]) are bank-conflict free.
#include "ovis_common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int BM = 64, BN = 64, BK = 16, LDP = BM + 4;

// A [64 x BK] operand tile travels global -> registers -> LDS (as T[k][row]) in two steps so that the loads of
// tile k+1 are in flight while the matrix cores work on tile k.  `rs`/`cs` are the element strides of (row, k).
// mode 0: k contiguous (cs == 1), 16-byte aligned rows: 4 lanes x float4 cover one row's 16 k
// mode 1: rows contiguous (rs == 1): 16 lanes x float4 cover the 64 rows of one k
// mode 2: generic strides / ragged edges, element-wise with bounds checks
__device__ __forceinline__ float4 load_tile(const float* __restrict__ P, long rs, long cs, int row0, int k0, int nrows,
                                            int K, int mode) {
  const int t = threadIdx.x;
  if (mode == 0) {
    const int r = t >> 2, kq = (t & 3) * 4;
    return *(const float4*)(P + (long)(row0 + r) * rs + (k0 + kq));
  }
  if (mode == 1) {
    const int k = t >> 4, rq = (t & 15) * 4;
    return *(const float4*)(P + (long)(k0 + k) * cs + (row0 + rq));
  }
  float4 v;
  float* pv = (float*)&v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = t + e * 256;
    const int k = idx >> 6, r = idx & 63;
    pv[e] = (row0 + r < nrows && k0 + k < K) ? P[(long)(row0 + r) * rs + (long)(k0 + k) * cs] : 0.f;
  }
  return v;
}

__device__ __forceinline__ void store_tile(float* T, float4 v, int mode) {
  const int t = threadIdx.x;
  if (mode == 0) {
    const int r = t >> 2, kq = (t & 3) * 4;
    T[(kq + 0) * LDP + r] = v.x;
    T[(kq + 1) * LDP + r] = v.y;
    T[(kq + 2) * LDP + r] = v.z;
    T[(kq + 3) * LDP + r] = v.w;
  } else if (mode == 1) {
    const int k = t >> 4, rq = (t & 15) * 4;
    *(float4*)(T + k * LDP + rq) = v;
  } else {
    const float* pv = (const float*)&v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = t + e * 256;
      T[(idx >> 6) * LDP + (idx & 63)] = pv[e];
    }
  }
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long a_rs, long a_cs,
                                                      const float* __restrict__ B, long b_rs, long b_cs,
                                                      const float* __restrict__ bias, float* __restrict__ C,
                                                      long c_rs, int M, int N, int K, int flags, float alpha,
                                                      int k_chunk) {
  // split-K: slice z of the grid contracts k in [z * k_chunk, min(K, (z + 1) * k_chunk)) and writes its partial tile
  // (bias from slice 0) into its own fp32 slab [z][M][N] of the caller's workspace (`C` then points at the slabs and c_rs
  // == N); gemm_f32_slab_sum_kernel adds the slabs in slice order -- no atomics, bit-reproducible.  Small GEMMs of the
  // head fill fewer than one workgroup per CU otherwise, and with one wave per SIMD nothing hides the LDS / barrier stalls.
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDP];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDP];
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  // full-tile vector paths need the whole 64-row tile in range (and every k-step full: checked per step)
  const bool a_full = m0 + BM <= M, b_full = n0 + BN <= N;
  const int a_vec = !a_full ? 2 : (flags & 1) ? 0 : (flags & 2) ? 1 : 2;
  const int b_vec = !b_full ? 2 : (flags & 4) ? 0 : (flags & 8) ? 1 : 2;
  const int k_begin = blockIdx.z * k_chunk;
  const int k_end = min(K, k_begin + k_chunk);
  auto mode_at = [&](int vec, int k0) { return k0 + BK <= K ? vec : 2; };
  f16v acc = {0.f};
  float4 ra = load_tile(A, a_rs, a_cs, m0, k_begin, M, K, mode_at(a_vec, k_begin));
  float4 rb = load_tile(B, b_rs, b_cs, n0, k_begin, N, K, mode_at(b_vec, k_begin));
  int buf = 0;
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
    store_tile(As[buf], ra, mode_at(a_vec, k0));
    store_tile(Bs[buf], rb, mode_at(b_vec, k0));
    __syncthreads();  // one barrier per k-step: the other buffer was last read before the previous barrier
    if (k0 + BK < k_end) {  // next tile's loads fly while this tile is in the matrix pipe
      ra = load_tile(A, a_rs, a_cs, m0, k0 + BK, M, K, mode_at(a_vec, k0 + BK));
      rb = load_tile(B, b_rs, b_cs, n0, k0 + BK, N, K, mode_at(b_vec, k0 + BK));
    }
    const float* ap = As[buf] + (lane >> 5) * LDP + wm + (lane & 31);
    const float* bp = Bs[buf] + (lane >> 5) * LDP + wn + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * 2 * LDP], bp[kk * 2 * LDP], acc, 0, 0, 0);
    buf ^= 1;
  }
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  // epilogue: C = alpha * acc (+ C if flags&16) + bias[col] (or bias[row] if flags&32)
  const int col = n0 + wn + (lane & 31);
  const bool accumulate = flags & 16, row_bias = flags & 32, split = flags & 64;
  if (split && blockIdx.z != 0) bias = nullptr;
  if (col < N) {
    const float bv = (bias && !row_bias) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < M) {
        float v = alpha * acc[r] + bv;
        if (bias && row_bias) v += bias[row];
        float* dst = C + ((split ? (long)blockIdx.z * M : 0) + row) * c_rs + col;
        *dst = (accumulate && !split) ? *dst + v : v;
      }
    }
  }
}

// C = (C +) slab[0] + slab[1] + ... in slice order (4 columns per thread when N % 4 == 0, else one)
__global__ __launch_bounds__(256) void gemm_f32_slab_sum_kernel(const float* __restrict__ slabs, float* __restrict__ C,
                                                               long c_rs, int M, int N, int slices, int accumulate) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / N;
    const int n = (int)(i - m * N);
    float v = slabs[i];
    for (int z = 1; z < slices; ++z) v += slabs[(long)z * total + i];
    float* dst = C + m * c_rs + n;
    *dst = accumulate ? *dst + v : v;
  }
}

// Per-noun best region: scores[w] = sigmoid(max_p <emb[p], noun[w]>), index[w] = argmax_p (lowest p on ties).
// maskrcnn_benchmark/modeling/detector/st_generalized_rcnn.py:243-262.  Grid = (noun, chunk of 64 regions): each
// wave walks 16 regions with the embedding dimension split over its lanes (16-byte loads), the workgroup's best
// (score, region) goes into the noun's 64-bit key with one atomicMax -- key = order-preserving score bits << 32 |
// ~region, so the maximum key is the maximum score and, among equal scores, the lowest region -- and a second
// tiny kernel unpacks the keys.  (The first version ran ONE workgroup per noun: 5 workgroups on 256 CUs, 0.9 ms.)
constexpr int kAlignChunk = 64;

__device__ __forceinline__ unsigned ordered_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void region_noun_partial_kernel(const float* __restrict__ emb,
                                                                 const float* __restrict__ nouns,
                                                                 unsigned long long* __restrict__ keys, int P, int D) {
  __shared__ unsigned long long s_key[4];
  const int w = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p0 = blockIdx.y * kAlignChunk;
  const float* nv = nouns + (long)w * D;
  unsigned long long best = 0ull;
  for (int p = p0 + wave; p < min(p0 + kAlignChunk, P); p += 4) {
    const float* e = emb + (long)p * D;
    float s = 0.f;
    if ((D & 3) == 0) {
      for (int d = lane * 4; d < D; d += 256) {
        const float4 a = *(const float4*)(e + d), b = *(const float4*)(nv + d);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      }
    } else {
      for (int d = lane; d < D; d += 64) s += e[d] * nv[d];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const unsigned long long key = ((unsigned long long)ordered_bits(s) << 32) | (unsigned)(~p);
    best = key > best ? key : best;
  }
  if (lane == 0) s_key[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) best = s_key[k] > best ? s_key[k] : best;
    if (best) atomicMax(keys + w, best);
  }
}

__global__ void region_noun_finalize_kernel(const unsigned long long* __restrict__ keys, float* __restrict__ raw,
                                            float* __restrict__ prob, long long* __restrict__ index, int num_nouns) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= num_nouns) return;
  const unsigned long long key = keys[w];
  const unsigned ob = (unsigned)(key >> 32);
  const float best = __uint_as_float((ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob);
  raw[w] = best;
  prob[w] = 1.f / (1.f + expf(-best));
  index[w] = (long long)(unsigned)(~(unsigned)key);  // written last: `keys` aliases `index`
}

}  // namespace

// split K until ~3 workgroups per CU are resident (12 waves / CU), keeping >= 8 k-steps per slice
static int gemm_f32_slices(int M, int N, int K, int* k_chunk_out) {
  const int tiles = ovis_ceil_div(N, BN) * ovis_ceil_div(M, BM);
  int slices = 1;
  while (slices < 8 && tiles * slices < 3 * OVIS_NUM_CU && K / (slices * 2) >= 8 * BK) slices *= 2;
  int k_chunk = K;
  if (slices > 1) {
    k_chunk = ovis_ceil_div(ovis_ceil_div(K, slices), BK) * BK;
    slices = ovis_ceil_div(K, k_chunk);
  }
  if (k_chunk_out) *k_chunk_out = k_chunk;
  return slices;
}

extern "C" size_t ovis_gemm_f32_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int slices = gemm_f32_slices(M, N, K, nullptr);
  return slices > 1 ? (size_t)slices * M * N * sizeof(float) : 0;
}

extern "C" int ovis_gemm_ex_ws_f32(const float* A, long a_row_stride, long a_k_stride, const float* B,
                                   long b_row_stride, long b_k_stride, const float* bias, int bias_per_row,
                                   float alpha, int accumulate, float* C, long c_row_stride, int M, int N, int K,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (M < 0 || N < 0 || K < 0) return OVIS_EINVAL;
  if (M == 0 || N == 0) return OVIS_OK;
  if (!A || !B || !C) return OVIS_EINVAL;
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  int flags = 0;
  if (a_k_stride == 1 && a_row_stride % 4 == 0 && al16(A)) flags |= 1;
  else if (a_row_stride == 1 && a_k_stride % 4 == 0 && al16(A)) flags |= 2;
  if (b_k_stride == 1 && b_row_stride % 4 == 0 && al16(B)) flags |= 4;
  else if (b_row_stride == 1 && b_k_stride % 4 == 0 && al16(B)) flags |= 8;
  if (accumulate) flags |= 16;
  if (bias_per_row) flags |= 32;
  // K is cut only when the caller brought the slabs' workspace (ovis_gemm_f32_workspace_bytes); without it: one slice
  int k_chunk = K;
  int slices = workspace ? gemm_f32_slices(M, N, K, &k_chunk) : 1;
  if (slices > 1 && workspace_bytes < (size_t)slices * M * N * sizeof(float)) return OVIS_ENOSPC;
  if (slices == 1) k_chunk = K;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ovis_ceil_div(N, BN), ovis_ceil_div(M, BM), slices);
  if (slices > 1) {
    flags |= 64;
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, A, a_row_stride, a_k_stride, B, b_row_stride, b_k_stride, bias,
                       (float*)workspace, (long)N, M, N, K, flags, alpha, k_chunk);
    OVIS_LAUNCH_CHECK();
    const long total = (long)M * N;
    hipLaunchKernelGGL(gemm_f32_slab_sum_kernel, dim3((unsigned)(total / 256 + 1 < 4L * OVIS_NUM_CU ? total / 256 + 1 : 4L * OVIS_NUM_CU)),
                       dim3(256), 0, s, (const float*)workspace, C, c_row_stride, M, N, slices, accumulate);
    OVIS_LAUNCH_CHECK();
    return OVIS_OK;
  }
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, A, a_row_stride, a_k_stride, B, b_row_stride, b_k_stride, bias, C,
                     c_row_stride, M, N, K, flags, alpha, k_chunk);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_gemm_ex_f32(const float* A, long a_row_stride, long a_k_stride, const float* B,
                                long b_row_stride, long b_k_stride, const float* bias, int bias_per_row,
                                float alpha, int accumulate, float* C, long c_row_stride, int M, int N, int K,
                                void* stream) {
  return ovis_gemm_ex_ws_f32(A, a_row_stride, a_k_stride, B, b_row_stride, b_k_stride, bias, bias_per_row, alpha, accumulate,
                             C, c_row_stride, M, N, K, nullptr, 0, stream);
}

extern "C" int ovis_gemm_f32(const float* A, long a_row_stride, long a_k_stride, const float* B,
                             long b_row_stride, long b_k_stride, const float* bias, float* C,
                             long c_row_stride, int M, int N, int K, void* stream) {
  return ovis_gemm_ex_f32(A, a_row_stride, a_k_stride, B, b_row_stride, b_k_stride, bias, 0, 1.f, 0, C,
                          c_row_stride, M, N, K, stream);
}

extern "C" int ovis_region_noun_align_f32(const float* region_emb, const float* noun_emb, float* raw_scores,
                                          float* sigmoid_scores, int64_t* best_region, int num_regions,
                                          int num_nouns, int dim, void* stream) {
  if (num_regions <= 0 || num_nouns < 0 || dim <= 0) return OVIS_EINVAL;
  if (num_nouns == 0) return OVIS_OK;
  if (!region_emb || !noun_emb || !raw_scores || !sigmoid_scores || !best_region) return OVIS_EINVAL;
  // the int64 index output doubles as the 64-bit key array of the partial kernel
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)best_region;
  OVIS_HIP_TRY(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * num_nouns, s));
  hipLaunchKernelGGL(region_noun_partial_kernel, dim3(num_nouns, ovis_ceil_div(num_regions, kAlignChunk)), dim3(256),
                     0, s, region_emb, noun_emb, keys, num_regions, dim);
  OVIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(region_noun_finalize_kernel, dim3(ovis_ceil_div(num_nouns, 64)), dim3(64), 0, s, keys,
                     raw_scores, sigmoid_scores, (long long*)best_region, num_nouns);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
