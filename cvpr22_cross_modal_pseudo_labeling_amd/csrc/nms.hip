// Greedy IoU non-maximum suppression for gfx950 (MI355X), fully device-resident.
//
// Reference behaviour: maskrcnn_benchmark/csrc/cuda/nms.cu:13-131 (64x64 IoU bitmask tiles,
// then a *host* loop over an 18 MB D2H copy of the mask) and csrc/nms.h:10-28.
// Here the same bitmask formulation is kept (it makes the result bit-exact by construction:
// the IoU expression below is devIoU's, evaluated with FP contraction off), but
//   * the score sort is a device radix sort (stable: ties keep the lower index first),
//   * the sorted boxes are gathered on the fly, only the upper triangle of tiles is computed,
//   * the sequential reduce runs on the device in ONE 1024-lane workgroup: wave 0 resolves a
//     64-box chunk from its diagonal tile with scalar bit tricks (wave64 == one mask word),
//     then all 16 waves OR the surviving rows into the LDS-resident `removed` words,
//   * survivors are compacted in ascending ORIGINAL index order through an LDS bitmap, so
//     the reference's final index sort (nms.cu:127-130) is not needed.
// Nothing is copied to the host; the caller reads back one int32 (the count) if it needs
// a dense result tensor.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "ovis_common.h"

namespace {

constexpr int kTile = 64;          // == wavefront size == bits per mask word
constexpr int kReduceThreads = 1024;
constexpr int kMaxBlocks = 2048;   // K <= 131072 (keeps the reduce kernel under 64 KB of LDS)

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ void iota_kernel(int* idx, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = i;
}

// devIoU (nms.cu:13-21) with the operand roles of the reference: a = higher-scored box.
__device__ __forceinline__ float iou_xyxy(const float4 a, const float4 b) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
  const float inter = width * height;
  const float sa = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
  const float sb = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  return inter / (sa + sb - inter);
}

// grid (nb, nb); block (row_blk = y, col_blk = x), tiles below the diagonal exit.
// GROUPED: a box only suppresses boxes of its own group (the per-class NMS of the detection post-processing,
// roi_heads/box_head/inference.py:121-163, as ONE launch over the candidates of every class).
template <bool GE, bool GROUPED>
__global__ __launch_bounds__(kTile) void nms_mask_kernel(const float4* __restrict__ boxes,
                                                         const int* __restrict__ groups,
                                                         const int* __restrict__ order, int K,
                                                         int nb, float thr,
                                                         unsigned long long* __restrict__ mask,
                                                         unsigned long long* __restrict__ diag_t) {
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (row_blk > col_blk) return;
  {  // batched form: image blockIdx.z of [N, K, ...] inputs; order == nullptr: the boxes come score-sorted already
    const size_t img = blockIdx.z;
    boxes += img * K;
    if (GROUPED) groups += img * K;
    mask += img * K * nb;
    diag_t += img * K;
  }
  __shared__ float4 col_boxes[kTile];
  __shared__ int col_group[kTile];
  const int lane = threadIdx.x;
  const int col = col_blk * kTile + lane;
  if (col < K) {
    const int o = order ? order[col] : col;
    col_boxes[lane] = boxes[o];
    if (GROUPED) col_group[lane] = groups[o];
  }
  __syncthreads();
  const int row = row_blk * kTile + lane;
  unsigned long long bits = 0;
  if (row < K) {
    const int orow = order ? order[row] : row;
    const float4 a = boxes[orow];
    const int ga = GROUPED ? groups[orow] : 0;
    const int ncol = min(K - col_blk * kTile, kTile);
    const int start = (row_blk == col_blk) ? lane + 1 : 0;
    for (int j = start; j < ncol; ++j) {
      if (GROUPED && col_group[j] != ga) continue;
      const float v = iou_xyxy(a, col_boxes[j]);
      const bool hit = GE ? (v >= thr) : (v > thr);
      if (hit) bits |= 1ull << j;
    }
    mask[(size_t)row * nb + col_blk] = bits;
  }
  if (row_blk == col_blk) {
    // transposed diagonal tile: word c = the higher-scored boxes of this chunk that suppress box c.  The
    // device-side reduce resolves a chunk from it with a few ballot iterations instead of a 64-step loop.
    unsigned long long mine = 0;
#pragma unroll 8
    for (int c = 0; c < kTile; ++c) {
      const unsigned long long col = __ballot((bits >> c) & 1ull);
      if (lane == c) mine = col;
    }
    if (row < K) diag_t[row] = mine;
  }
}

// Tail shared by both reduce kernels: survivors (keepw, by sorted position) -> bitmap by ORIGINAL index -> ascending
// compaction into keep_out.  groups != nullptr: boxes with a negative group never survive (the RPN's small-box filter,
// rpn/inference.py:114: they are flagged instead of being removed in front of the NMS).  nk_stride == 2 (batched form):
// num_keep[1] = survivors with index < `below` (they form a prefix of keep_out) and keep_out is zero-filled past the
// survivors, so that a padded gather is safe.
__device__ __forceinline__ void nms_compact(const unsigned long long* keepw, unsigned long long* obits, int* part,
                                            const int* __restrict__ order, const int* __restrict__ groups, int K, int nb,
                                            int below, int nk_stride, long long* __restrict__ keep_out,
                                            int* __restrict__ num_keep) {
  __shared__ int below_cnt;
  const int tid = threadIdx.x;
  if (tid == 0) below_cnt = 0;
  for (int p = tid; p < K; p += kReduceThreads) {
    if ((keepw[p >> 6] >> (p & 63)) & 1ull) {
      const int o = order ? order[p] : p;
      if (groups == nullptr || groups[o] >= 0) atomicOr(&obits[o >> 6], 1ull << (o & 63));
    }
  }
  __syncthreads();
  // exclusive scan of per-thread popcounts over contiguous word segments
  const int seg = (nb + kReduceThreads - 1) / kReduceThreads;
  const int w0 = tid * seg, w1 = min(nb, w0 + seg);
  int cnt = 0, cb = 0;
  for (int w = w0; w < w1; ++w) {
    const unsigned long long bits = obits[w];
    cnt += __popcll(bits);
    const int lim = below - w * 64;  // bits [0, lim) of this word lie below the limit
    if (lim > 0) cb += __popcll(lim >= 64 ? bits : (bits & ((1ull << lim) - 1ull)));
  }
  part[tid] = cnt;
  if (cb) atomicAdd(&below_cnt, cb);
  __syncthreads();
  for (int off = 1; off < kReduceThreads; off <<= 1) {
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int pos = part[tid] - cnt;
  for (int w = w0; w < w1; ++w) {
    unsigned long long bits = obits[w];
    while (bits) {
      const int i = __builtin_ctzll(bits);
      bits &= bits - 1;
      keep_out[pos++] = (long long)w * 64 + i;
    }
  }
  const int total = part[kReduceThreads - 1];
  if (nk_stride == 2)
    for (int i = total + tid; i < K; i += kReduceThreads) keep_out[i] = 0;
  if (tid == kReduceThreads - 1) {
    num_keep[0] = total;
    if (nk_stride == 2) num_keep[1] = below_cnt;
  }
}

__global__ __launch_bounds__(kReduceThreads) void nms_reduce_kernel(
    const unsigned long long* __restrict__ mask, const int* __restrict__ order, const int* __restrict__ groups, int K,
    int nb, int below, int nk_stride, long long* __restrict__ keep_out, int* __restrict__ num_keep) {
  {
    const size_t img = blockIdx.x;
    mask += img * K * nb;
    if (groups) groups += img * K;
    keep_out += img * K;
    num_keep += img * nk_stride;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
  unsigned long long* removed = sm;        // [nb] suppression bits per sorted position
  unsigned long long* keepw = sm + nb;     // [nb] survivors per sorted position
  unsigned long long* obits = sm + 2 * nb; // [nb] survivors per ORIGINAL index
  int* part = (int*)(sm + 3 * nb);         // [kReduceThreads] scan scratch
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < nb; i += kReduceThreads) { removed[i] = 0; obits[i] = 0; }
  __syncthreads();

  for (int b = 0; b < nb; ++b) {
    if (wave == 0) {
      const int row = b * kTile + lane;
      const unsigned long long diag = row < K ? mask[(size_t)row * nb + b] : 0ull;
      const int nvalid = min(K - b * kTile, kTile);
      const unsigned long long valid = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
      unsigned long long alive = uniform64(~removed[b] & valid);
      unsigned long long keep = 0;
      while (alive) {
        const int i = __builtin_ctzll(alive);
        keep |= 1ull << i;
        alive &= ~(readlane64(diag, i) | (1ull << i));
      }
      if (lane == 0) keepw[b] = keep;
    }
    __syncthreads();
    const unsigned long long keep = uniform64(keepw[b]);
    if (keep != 0) {
      for (int j0 = b + 1; j0 < nb; j0 += kTile) {
        const int j = j0 + lane;
        unsigned long long acc = 0, kk = keep;
        int rank = 0;
        while (kk) {
          const int i = __builtin_ctzll(kk);
          kk &= kk - 1;
          if (((rank++) & 15) == wave && j < nb) acc |= mask[(size_t)(b * kTile + i) * nb + j];
        }
        if (acc) atomicOr(&removed[j], acc);
      }
    }
    __syncthreads();
  }

  // survivors -> bitmap indexed by original box index (ascending order comes for free)
  nms_compact(keepw, obits, part, order, groups, K, nb, below, nk_stride, keep_out, num_keep);
}


// Pipelined reduce for nb <= kFastBlocks (K <= 12288: every RPN setting of the reference).
// The chunk-b rows of the mask are needed only after chunk b is resolved, but WHICH bytes are needed does not
// depend on the result -- so every lane streams its slice of the 64 rows of chunk b+1 into registers while
// chunk b is being resolved, and applies the survivor mask to data that is already on chip.  Thread layout:
// row = tid/16 (64 rows of the chunk), 16 lanes per row cover words b + l16 + 16*m.  Per chunk the critical
// path is: barrier -> wave 0 resolves 64 boxes with scalar bit tricks -> barrier -> masked OR of registers
// (4 rows pre-reduced in-wave with shuffles, then one 64-bit LDS atomic-or per word).
constexpr int kFastBlocks = 192;
constexpr int kMaxM = kFastBlocks / 16;

__device__ __forceinline__ unsigned long long shfl_xor64(unsigned long long v, int m) {
  unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(kReduceThreads) void nms_reduce_pipelined_kernel(
    const unsigned long long* __restrict__ mask, const unsigned long long* __restrict__ diag_t,
    const int* __restrict__ order, const int* __restrict__ groups, int K, int nb, int below, int nk_stride,
    long long* __restrict__ keep_out, int* __restrict__ num_keep) {
  {
    const size_t img = blockIdx.x;
    mask += img * K * nb;
    diag_t += img * K;
    if (groups) groups += img * K;
    keep_out += img * K;
    num_keep += img * nk_stride;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned long long sm[];
  unsigned long long* removed = sm;
  unsigned long long* keepw = sm + nb;
  unsigned long long* obits = sm + 2 * nb;
  int* part = (int*)(sm + 3 * nb);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = tid >> 4, l16 = tid & 15;
  for (int i = tid; i < nb; i += kReduceThreads) { removed[i] = 0; obits[i] = 0; }

  // three register buffers: chunk b is consumed while chunks b+1 and b+2 are in flight from L2 / MALL
  unsigned long long buf0[kMaxM], buf1[kMaxM], buf2[kMaxM], dg0 = 0, dg1 = 0, dg2 = 0;
  auto load_chunk = [&](int b, unsigned long long* dst, unsigned long long& diag) {
    const long r = (long)b * kTile + row;
    const unsigned long long* src = mask + r * nb;
#pragma unroll
    for (int m = 0; m < kMaxM; ++m) {
      const int j = b + 1 + l16 + 16 * m;
      dst[m] = (b < nb && r < K && j < nb) ? src[j] : 0ull;
    }
    if (wave == 0) {  // wave 0 additionally streams the transposed diagonal tile it resolves from
      const long rd = (long)b * kTile + lane;
      diag = (b < nb && rd < K) ? diag_t[rd] : 0ull;
    }
  };
  auto step = [&](int b, const unsigned long long* cur, unsigned long long diag_cur) {
    if (wave == 0) {
      // Greedy survivors of the chunk as a fixed point: box i survives iff it is alive and no SURVIVING
      // higher-scored box of the chunk suppresses it.  Starting from "all alive", after t rounds the first
      // t boxes are final, so the iteration reaches the unique greedy answer in <= 64 rounds (typically 2-4);
      // one round is a lane-local AND plus a ballot.
      const int nvalid = min(K - b * kTile, kTile);
      const unsigned long long valid = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
      const unsigned long long alive = uniform64(~removed[b] & valid);
      const bool me_alive = (alive >> lane) & 1ull;
      unsigned long long keep = alive;
      for (int it = 0; it <= kTile; ++it) {
        const unsigned long long next = __ballot(me_alive && (diag_cur & keep) == 0ull);
        if (next == keep) break;
        keep = next;
      }
      if (lane == 0) keepw[b] = keep;
    }
    __syncthreads();
    const unsigned long long keep = uniform64(keepw[b]);
    const bool mine = (keep >> row) & 1ull;
#pragma unroll
    for (int m = 0; m < kMaxM; ++m) {
      if (b + 1 + 16 * m >= nb) break;  // uniform
      // 64-bit integer LDS atomics are cheap on gfx950 (~4-15 cycles per wave instruction even with the
      // 4 rows of a wave colliding), cheaper than pre-reducing the rows with cross-lane shuffles
      const unsigned long long v = mine ? cur[m] : 0ull;
      const int j = b + 1 + l16 + 16 * m;
      if (v != 0ull && j < nb) atomicOr(&removed[j], v);
    }
    __syncthreads();
  };
  load_chunk(0, buf0, dg0);
  load_chunk(1, buf1, dg1);
  __syncthreads();
  for (int b = 0; b < nb; b += 3) {
    load_chunk(b + 2, buf2, dg2);
    step(b, buf0, dg0);
    if (b + 1 >= nb) break;
    load_chunk(b + 3, buf0, dg0);
    step(b + 1, buf1, dg1);
    if (b + 2 >= nb) break;
    load_chunk(b + 4, buf1, dg1);
    step(b + 2, buf2, dg2);
  }

  nms_compact(keepw, obits, part, order, groups, K, nb, below, nk_stride, keep_out, num_keep);
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct NmsLayout {
  size_t keys_out, idx_in, order, sort_temp, sort_bytes, mask, diag_t, total;
};

int nms_layout(int K, NmsLayout* L) {
  const size_t nb = (size_t)ovis_ceil_div(K, kTile);
  size_t sort_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, sort_bytes, (const float*)nullptr, (float*)nullptr,
                                                (const int*)nullptr, (int*)nullptr, (size_t)K);
  if (e != hipSuccess) return (int)e;
  size_t off = 0;
  L->keys_out = off; off = align256(off + sizeof(float) * (size_t)K);
  L->idx_in = off;   off = align256(off + sizeof(int) * (size_t)K);
  L->order = off;    off = align256(off + sizeof(int) * (size_t)K);
  L->sort_temp = off; off = align256(off + sort_bytes);
  L->sort_bytes = sort_bytes;
  L->mask = off;     off = align256(off + sizeof(unsigned long long) * (size_t)K * nb);
  L->diag_t = off;   off = align256(off + sizeof(unsigned long long) * (size_t)K);
  L->total = off;
  return OVIS_OK;
}

}  // namespace

extern "C" size_t ovis_nms_workspace_bytes(int num_boxes) {
  if (num_boxes <= 0) return 0;
  NmsLayout L;
  if (nms_layout(num_boxes, &L) != OVIS_OK) return 0;
  return L.total;
}

static int nms_impl(const float* boxes, const float* scores, const int* groups, int num_boxes, float threshold,
                    int ge_mode, void* workspace, size_t workspace_bytes, int64_t* keep_out, int32_t* num_keep,
                    void* stream);

extern "C" int ovis_nms_f32(const float* boxes, const float* scores, int num_boxes,
                            float threshold, int ge_mode, void* workspace,
                            size_t workspace_bytes, int64_t* keep_out, int32_t* num_keep,
                            void* stream) {
  return nms_impl(boxes, scores, nullptr, num_boxes, threshold, ge_mode, workspace, workspace_bytes, keep_out, num_keep,
                  stream);
}

extern "C" int ovis_nms_grouped_f32(const float* boxes, const float* scores, const int32_t* groups, int num_boxes,
                                    float threshold, int ge_mode, void* workspace, size_t workspace_bytes,
                                    int64_t* keep_out, int32_t* num_keep, void* stream) {
  if (num_boxes > 0 && !groups) return OVIS_EINVAL;
  return nms_impl(boxes, scores, groups, num_boxes, threshold, ge_mode, workspace, workspace_bytes, keep_out, num_keep,
                  stream);
}

static int nms_impl(const float* boxes, const float* scores, const int* groups, int num_boxes, float threshold,
                    int ge_mode, void* workspace, size_t workspace_bytes, int64_t* keep_out, int32_t* num_keep,
                    void* stream) {
  if (num_boxes < 0 || !num_keep) return OVIS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (num_boxes == 0) {
    OVIS_HIP_TRY(hipMemsetAsync(num_keep, 0, sizeof(int32_t), s));
    return OVIS_OK;
  }
  if (!boxes || !scores || !keep_out || !workspace) return OVIS_EINVAL;
  const int K = num_boxes;
  const int nb = ovis_ceil_div(K, kTile);
  if (nb > kMaxBlocks) return OVIS_ERANGE;
  NmsLayout L;
  int rc = nms_layout(K, &L);
  if (rc != OVIS_OK) return rc;
  if (workspace_bytes < L.total) return OVIS_ENOSPC;
  char* ws = (char*)workspace;
  float* keys_out = (float*)(ws + L.keys_out);
  int* idx_in = (int*)(ws + L.idx_in);
  int* order = (int*)(ws + L.order);
  unsigned long long* mask = (unsigned long long*)(ws + L.mask);
  unsigned long long* diag_t = (unsigned long long*)(ws + L.diag_t);

  hipLaunchKernelGGL(iota_kernel, dim3(ovis_ceil_div(K, 256)), dim3(256), 0, s, idx_in, K);
  OVIS_LAUNCH_CHECK();
  size_t sort_bytes = L.sort_bytes;
  // rocPRIM's device radix sort (ROCm's native primitive library): stable, so equal scores keep the lower index first
  OVIS_HIP_TRY(rocprim::radix_sort_pairs_desc((void*)(ws + L.sort_temp), sort_bytes, scores, keys_out, (const int*)idx_in,
                                              order, (size_t)K, 0u, 32u, s));
  dim3 grid(nb, nb);
#define OVIS_NMS_MASK(GE_, GR_)                                                                            \
  hipLaunchKernelGGL((nms_mask_kernel<GE_, GR_>), grid, dim3(kTile), 0, s, (const float4*)boxes, groups, order, K, nb, \
                     threshold, mask, diag_t)
  if (groups) { if (ge_mode) OVIS_NMS_MASK(true, true); else OVIS_NMS_MASK(false, true); }
  else { if (ge_mode) OVIS_NMS_MASK(true, false); else OVIS_NMS_MASK(false, false); }
#undef OVIS_NMS_MASK
  OVIS_LAUNCH_CHECK();
  const size_t lds = sizeof(unsigned long long) * 3 * (size_t)nb + sizeof(int) * kReduceThreads;
  if (nb <= kFastBlocks)
    hipLaunchKernelGGL(nms_reduce_pipelined_kernel, dim3(1), dim3(kReduceThreads), lds, s, mask, diag_t, order,
                       (const int*)nullptr, K, nb, 0, 1, (long long*)keep_out, num_keep);
  else
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(kReduceThreads), lds, s, mask, order, (const int*)nullptr, K, nb,
                       0, 1, (long long*)keep_out, num_keep);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

// Batched, score-sorted form for the RPN proposal pipeline (rpn/inference.py:95-116): the candidates of every image
// arrive as the top-k prefix in descending score order, so there is no sort; `drop` flags (negative = the box was
// removed by the small-box filter in front of the NMS) ride on the group mechanism.
extern "C" size_t ovis_nms_presorted_workspace_bytes(int num_images, int num_boxes) {
  if (num_images <= 0 || num_boxes <= 0) return 0;
  const size_t nb = (size_t)ovis_ceil_div(num_boxes, kTile);
  return (size_t)num_images * (align256(sizeof(unsigned long long) * (size_t)num_boxes * nb) +
                               align256(sizeof(unsigned long long) * (size_t)num_boxes));
}

extern "C" int ovis_nms_presorted_batched_f32(const float* boxes, const int32_t* drop, int num_images, int num_boxes,
                                              float threshold, int ge_mode, int below, void* workspace,
                                              size_t workspace_bytes, int64_t* keep_out, int32_t* num_keep,
                                              void* stream) {
  if (num_images < 0 || num_boxes < 0 || !num_keep) return OVIS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (num_images == 0) return OVIS_OK;
  if (num_boxes == 0) {
    OVIS_HIP_TRY(hipMemsetAsync(num_keep, 0, sizeof(int32_t) * 2 * (size_t)num_images, s));
    return OVIS_OK;
  }
  if (!boxes || !keep_out || !workspace) return OVIS_EINVAL;
  const int K = num_boxes, nb = ovis_ceil_div(K, kTile);
  if (nb > kMaxBlocks || num_images > 65535) return OVIS_ERANGE;
  if (workspace_bytes < ovis_nms_presorted_workspace_bytes(num_images, K)) return OVIS_ENOSPC;
  // contiguous [N][K * nb] masks followed by [N][K] transposed diagonal tiles (the kernels index them by image)
  unsigned long long* mask = (unsigned long long*)workspace;
  unsigned long long* diag_t = mask + (size_t)num_images * K * nb;
  dim3 grid(nb, nb, num_images);
#define OVIS_NMS_MASK(GE_, GR_)                                                                                          \
  hipLaunchKernelGGL((nms_mask_kernel<GE_, GR_>), grid, dim3(kTile), 0, s, (const float4*)boxes, (const int*)drop,       \
                     (const int*)nullptr, K, nb, threshold, mask, diag_t)
  if (drop) { if (ge_mode) OVIS_NMS_MASK(true, true); else OVIS_NMS_MASK(false, true); }
  else { if (ge_mode) OVIS_NMS_MASK(true, false); else OVIS_NMS_MASK(false, false); }
#undef OVIS_NMS_MASK
  OVIS_LAUNCH_CHECK();
  const size_t lds = sizeof(unsigned long long) * 3 * (size_t)nb + sizeof(int) * kReduceThreads;
  if (nb <= kFastBlocks)
    hipLaunchKernelGGL(nms_reduce_pipelined_kernel, dim3(num_images), dim3(kReduceThreads), lds, s, mask, diag_t,
                       (const int*)nullptr, (const int*)drop, K, nb, below, 2, (long long*)keep_out, num_keep);
  else
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(num_images), dim3(kReduceThreads), lds, s, mask, (const int*)nullptr,
                       (const int*)drop, K, nb, below, 2, (long long*)keep_out, num_keep);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
