// RoIAlign backward for gfx950 (MI355X), plane-owner form on the matrix cores.
//
// Reference semantics: maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:125-254 (four atomicAdd per
// bin x sample x channel into a zero-filled grad_input).  Here no atomic is issued at all:
//
//  * Bilinear average pooling is separable, out_c = Ay . win_c . Ax^T, so the gradient of a RoI's
//    window is  dwin_c = Ay^T . G_c . Ax  with G_c the 14x14 grad_output tile of channel c,
//    Ax[j][x] the summed column weights of bin-column j's samples and Ay[i][y] likewise (x 1/count).
//    Both factors depend on the RoI only, NOT on the channel.
//  * A plan kernel evaluates Ax / Ay once per RoI, cuts the window into 16-cell blocks per axis and
//    stores every block already in the lane layout of an MFMA operand, split into bf16 hi + lo parts
//    (hi + lo carries 16 mantissa bits; the three products hi.hi + hi.lo + lo.hi are accumulated in
//    fp32 by the matrix core, error ~1e-5 relative per term, far inside the 1e-3 contract).
//  * The main kernel gives every (image n, channel c) gradient plane to ONE wave, which keeps the
//    whole H x W plane in LDS (16.8 KB for the C4 map; 8 waves = 8 planes per CU, 2048 planes = one
//    full round of the 256 CUs) and walks the image's RoIs in list order.  Per RoI the wave loads its
//    784-byte G tile straight from HBM into the A-operand layout (prefetched 6 RoIs deep in
//    registers), and runs the chain
//        T  = G_c  . Ax_blk      (v_mfma_f32_16x16x16_bf16 x3)      [bins x 16 window columns]
//        dW = Ay_blk^T . T       (x3; T's accumulator layout IS the B-operand layout, no shuffle)
//    then adds dW into its plane with plain LDS read-modify-write -- exclusive ownership, so no
//    atomics, no barriers (a workgroup is a single wave), bit-reproducible run to run.
//  * The plane is written to HBM once with coalesced stores: no zero-fill pass, HBM traffic equals the
//    algorithmic bytes (grad_output once + grad_input once).
//
// LDS carries only the plane read-modify-write (8 ds ops per RoI x channel); the sparse gather the
// previous kernels did through LDS / global atomics is two dense 16x16x16 products on the otherwise
// idle matrix pipe.
#include "ovis_common.h"
#include "roi_geom.h"

namespace {
using namespace ovis_roi;

typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int kT = 16;        // window block edge == MFMA tile edge
constexpr int kDepth = 10;    // items in flight per wave (G tile + table blocks): ~63 KB of G per CU covers ~2 us of HBM latency
constexpr int kPlanThreads = 1024;
constexpr int kMinLds = 20 * 1024;  // 160 KB / 8: pins residency at 8 single-wave workgroups per CU

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));  // v_cvt_pk_bf16_f32 (RNE)
}

// v -> bf16 hi (returned .x,.y) and bf16 lo of the exact remainder (.z,.w); element e of the 4-vector
// sits in half (e & 1) of word (e >> 1), which is the k-order of an MFMA 16x16x16 operand.
__device__ __forceinline__ u4 split_bf16(f4 v) {
  const unsigned h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
  const float r0 = v.x - __uint_as_float(h01 << 16), r1 = v.y - __uint_as_float(h01 & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(h23 << 16), r3 = v.w - __uint_as_float(h23 & 0xffff0000u);
  return (u4){h01, h23, pack_bf16(r0, r1), pack_bf16(r2, r3)};
}

__device__ __forceinline__ s4 as_s4(unsigned a, unsigned b) {
  u2 u = {a, b};
  return __builtin_bit_cast(s4, u);
}

// D += (Ahi + Alo) . (Bhi + Blo) without the lo.lo term
__device__ __forceinline__ f4 mfma3(u4 a, u4 b, f4 acc) {
  const s4 ah = as_s4(a.x, a.y), al = as_s4(a.z, a.w), bh = as_s4(b.x, b.y), bl = as_s4(b.z, b.w);
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);
  return acc;
}

// Summed weight the samples of bin `p` put on feature cell `cell` along one axis (the reference's
// bilinear_interpolate_gradient set-up, ROIAlign_cuda.cu:125-175, reduced to one axis).
__device__ __forceinline__ float axis_weight(float start, float bin, int grid, int p, int size, int cell) {
  float acc = 0.f;
  for (int i = 0; i < grid; ++i) {
    int lo, hi;
    float l, h;
    if (!axis_sample(sample_coord(start, p, bin, i, grid), size, lo, hi, l, h)) continue;
    acc += (lo == cell ? h : 0.f) + (hi == cell ? l : 0.f);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------------
// Plan kernel.
// A RoI's window is cut into 16 x 16 blocks; block (xb, yb) OWNS columns [wx0+16xb, wx0+16xb+15] & rows
// likewise, clipped to the window.  Its 16 x 16 footprint starts at origin (oy, ox) = the owned corner pulled
// back so the footprint lies inside the map (when the map is at least 16 wide / high): every cell of the
// footprint is a valid plane cell, and the table entries of cells the block does not own are exactly zero, so
// the main kernel can add the whole footprint unconditionally.
//
// Blocks [0, batch): the 16 waves of block n build image n's item list -- one uint4 of BYTE offsets
//   {G tile of (r, channel 0) in grad_output, footprint origin in the plane, tx block, ty block}
// per (RoI, xb, yb), RoIs in input order, empty / foreign RoIs dropped, padded with zero-contribution items to
// a whole number of ring rounds plus the prefetch run-ahead.  Remaining blocks: one wave per RoI writes its
// table blocks (16 B per lane):
//   tx[r * NXB + xb][lane] = B operand of stage 1 (bf16 hi x4 | bf16 lo x4): lane (col q = lane & 15, k-group
//                            s = lane >> 4) holds Ax[j = min(4s, PW-4)+e][ox+q], zero where j < 4s (the pulled-back
//                            last group repeats bins of the previous one);
//   ty[r * NYB + yb][lane] = A operand of stage 2 (same packing): lane (row q, s) holds Ay[i = 4s+e][oy+q] / count.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ inline long list_stride(int R, int NXB, int NYB) { return (long)R * NXB * NYB + 2 * kDepth + 2; }

__device__ __forceinline__ int block_origin(int w0, int blk, int size) {
  return max(min(w0 + blk * kT, size - kT), 0);
}

__global__ __launch_bounds__(kPlanThreads) void roi_bwd_plan_kernel(
    const float* __restrict__ rois, int R, int batch, int C, int H, int W, int PH, int PW, float scale,
    int sampling_ratio, u4* __restrict__ list, int* __restrict__ counts, u4* __restrict__ tx,
    u4* __restrict__ ty, int NXB, int NYB) {
  __shared__ int wave_total[kPlanThreads / 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < batch) {
    const int n = blockIdx.x;
    u4* my = list + (long)n * list_stride(R, NXB, NYB);
    if (n == 0 && wave == 0) {  // the all-zero table blocks the padding items point at
      tx[((long)R * NXB) * 64 + lane] = (u4){0u, 0u, 0u, 0u};
      ty[((long)R * NYB) * 64 + lane] = (u4){0u, 0u, 0u, 0u};
    }
    int base = 0;
    for (int rb = 0; rb < R; rb += kPlanThreads) {
      const int r = rb + threadIdx.x;
      int nbx = 0, nby = 0;
      RoiGeom g;
      if (r < R) {
        g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
        if (!g.empty && g.b == n) {
          nbx = (g.wx1 - g.wx0 + kT) / kT;
          nby = (g.wy1 - g.wy0 + kT) / kT;
        }
      }
      const int nb = nbx * nby;
      int incl = nb;  // inclusive wave prefix sum
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
      }
      if (lane == 63) wave_total[wave] = incl;
      __syncthreads();
      int before = 0, all = 0;
#pragma unroll
      for (int w = 0; w < kPlanThreads / 64; ++w) {
        const int t = wave_total[w];
        before += w < wave ? t : 0;
        all += t;
      }
      int o = base + before + incl - nb;
      for (int xb = 0; xb < nbx; ++xb)
        for (int yb = 0; yb < nby; ++yb)
          my[o++] = (u4){(unsigned)r * (unsigned)(C * PH * PW) * 4u,
                         (unsigned)(block_origin(g.wy0, yb, H) * W + block_origin(g.wx0, xb, W)) * 4u,
                         (unsigned)(r * NXB + xb) * 1024u, (unsigned)(r * NYB + yb) * 1024u};
      base += all;
      __syncthreads();
    }
    // pad to a whole number of ring rounds plus the prefetch run-ahead with items that add exact zeros
    const int padded = (base + kDepth - 1) / kDepth * kDepth;
    for (int i = base + threadIdx.x; i < padded + kDepth + 1; i += kPlanThreads)
      my[i] = (u4){0u, 0u, (unsigned)(R * NXB) * 1024u, (unsigned)(R * NYB) * 1024u};
    if (threadIdx.x == 0) counts[n] = padded;
    return;
  }
  const int r = ((int)blockIdx.x - batch) * (kPlanThreads / 64) + wave;
  if (r >= R) return;
  const RoiGeom g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
  if (g.empty) return;
  const int q = lane & 15, s = lane >> 4;
  for (int xb = 0; g.wx0 + xb * kT <= g.wx1; ++xb) {
    const int col = block_origin(g.wx0, xb, W) + q;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (col >= g.wx0 + xb * kT && col <= min(g.wx0 + xb * kT + kT - 1, g.wx1)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = min(4 * s, PW - 4) + e;  // the main kernel's k-slot -> bin-column map (pulled-back last group)
        if (j >= 4 * s) v[e] = axis_weight(g.start_w, g.bin_w, g.gw, j, W, col);
      }
    }
    tx[((long)r * NXB + xb) * 64 + lane] = split_bf16(v);
  }
  for (int yb = 0; g.wy0 + yb * kT <= g.wy1; ++yb) {
    const int row = block_origin(g.wy0, yb, H) + q;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (row >= g.wy0 + yb * kT && row <= min(g.wy0 + yb * kT + kT - 1, g.wy1)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * s + e;
        if (i < PH) v[e] = axis_weight(g.start_h, g.bin_h, g.gh, i, H, row) / g.count;
      }
    }
    ty[((long)r * NYB + yb) * 64 + lane] = split_bf16(v);
  }
}

// ---------------------------------------------------------------------------------------------------
// Main kernel: one wave == one workgroup == one (image, channel) plane.
// The per-item code is straight-line: exactly three 16-byte loads per item (G slice, tx block, ty block), issued
// kDepth items ahead into a register ring, item scalars by s_load one step further ahead, footprint added
// unconditionally (FIT: map >= 16 x 16).
//
// The ring loads are inline asm so that hipcc does not count them: left to itself it waits vmcnt(0) before every
// item (its loop-carried bookkeeping gives up on a ring this deep) and the prefetch is drained each time.  Rules
// followed (guide 5.7 form ii): destinations are "+v" (refilled in place, never copied), every consumer sits
// below a wait statement naming the registers it reads, loads are issued in item order so the counted wait
// vmcnt(3 * kDepth - 3) retires exactly the oldest item, sched_barrier pins the statement order.  Addresses are
// SGPR base (kernel pointer + the item's byte offset, SALU) + a per-lane 32-bit VGPR offset that never changes.
// ---------------------------------------------------------------------------------------------------
#define OVIS_GLOAD4(dst, voff, sbase) \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define OVIS_WAIT3(N, a, b, c) \
  asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "i"(N) : "memory")

template <bool FIT>
__global__ __launch_bounds__(64) void roi_bwd_mfma_kernel(
    const float* __restrict__ gout, const u4* __restrict__ list, const int* __restrict__ counts,
    const u4* __restrict__ tx, const u4* __restrict__ ty, float* __restrict__ gin, int R, int batch,
    int C, int H, int W, int PH, int PW, int NXB, int NYB) {
  extern __shared__ __attribute__((aligned(16))) float plane[];
  const int lane = threadIdx.x;
  const int n = blockIdx.x % batch;  // neighbouring workgroups alternate images: a CU's 8 planes mix them
  const int c = blockIdx.x / batch;
  const int HW = H * W;
  const int PHPW = PH * PW;
  for (int i = lane; i < HW; i += 64) plane[i] = 0.f;

  const int cnt = counts[n];  // padded to a multiple of kDepth by the plan kernel
  float* dst = gin + ((long)n * C + c) * HW;
  {
    const u4* my = list + (long)n * list_stride(R, NXB, NYB);
    const int q = lane & 15, s = lane >> 4;
    // Lane's slice of a G tile: row q, four columns starting at min(4s, PW - 4) -- the last k-group is pulled back
    // so that no lane reads past its row (the plan kernel zeroes the duplicated k-slots in tx).  Rows q >= PH re-read
    // row PH - 1: finite values that meet zero ty entries (i >= PH).
    const unsigned g_lane = (unsigned)(min(q, PH - 1) * PW + min(4 * s, PW - 4)) * 4u;
    const unsigned t_lane = (unsigned)lane * 16u;
    const char* gbase = (const char*)(gout + (long)c * PHPW);
    const char* txb = (const char*)tx;
    const char* tyb = (const char*)ty;
    const unsigned lane_cell = (unsigned)(4 * s * W + q) * 4u;  // byte offset of the lane's first footprint cell

    f4 rg[kDepth];     // lane's 4 grad_output values G[i = min(q, PH-1)][j = min(4s, PW-4) + e] of the item in slot d
    u4 rbx[kDepth];    // its column block of tx
    u4 ray[kDepth];    // its row block of ty
    unsigned org[kDepth];  // its footprint origin (LDS byte offset)
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      rg[d] = (f4){0.f, 0.f, 0.f, 0.f};
      rbx[d] = ray[d] = (u4){0u, 0u, 0u, 0u};
    }
#define OVIS_FETCH(d, e)                         \
  do {                                           \
    const char* pg_ = gbase + (e).x;             \
    const char* pb_ = txb + (e).z;               \
    const char* pa_ = tyb + (e).w;               \
    OVIS_GLOAD4(rg[d], g_lane, pg_);             \
    OVIS_GLOAD4(rbx[d], t_lane, pb_);            \
    OVIS_GLOAD4(ray[d], t_lane, pa_);            \
    org[d] = (e).y;                              \
  } while (0)

#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      const u4 e = my[d];
      OVIS_FETCH(d, e);
    }
    u4 epre = my[kDepth];  // entry of the item to prefetch next

    // Software pipeline: the footprint of item k-1 is folded into the plane while item k runs through the matrix
    // pipe -- its four plane reads are issued before the wait on item k's loads, the adds + writes sit between
    // the two MFMA stages of item k.  Starts with a zero footprint at cell 0 (adds 0.0f, FIT) / nothing (masked).
    f4 w_prev = {0.f, 0.f, 0.f, 0.f};
    unsigned cell_prev = lane_cell;
    bool on_prev = false;  // masked form only: does the lane own a column of the previous footprint

    // cnt is a multiple of kDepth and the list runs kDepth + 1 entries past it (zero-contribution padding items)
    for (int k0 = 0; k0 < cnt; k0 += kDepth) {
#pragma unroll
      for (int d = 0; d < kDepth; ++d) {
        const int k = k0 + d;
        const unsigned cell = org[d] + lane_cell;
        float* pp = (float*)((char*)plane + cell_prev);
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (FIT) {
          v0 = pp[0]; v1 = pp[W]; v2 = pp[2 * W]; v3 = pp[3 * W];
        }
        __builtin_amdgcn_sched_barrier(0);
        OVIS_WAIT3(3 * kDepth - 3, rg[d], rbx[d], ray[d]);  // the oldest item's three loads have landed
        __builtin_amdgcn_sched_barrier(0);
        // stage 1: T[i][x] = sum_j G[i][j] Ax[j][x]
        const u4 a1 = split_bf16(rg[d]);
        const f4 t = mfma3(a1, rbx[d], (f4){0.f, 0.f, 0.f, 0.f});
        // previous item's footprint (its reads were issued above, their latency is behind stage 1 by now)
        if (FIT) {
          pp[0] = v0 + w_prev.x;
          pp[W] = v1 + w_prev.y;
          pp[2 * W] = v2 + w_prev.z;
          pp[3 * W] = v3 + w_prev.w;
        } else if (on_prev) {  // maps narrower / lower than one block: mask the footprint
          const int y = (int)(cell_prev >> 2) / W;
          if (y + 0 < H) pp[0] += w_prev.x;
          if (y + 1 < H) pp[W] += w_prev.y;
          if (y + 2 < H) pp[2 * W] += w_prev.z;
          if (y + 3 < H) pp[3 * W] += w_prev.w;
        }
        // stage 2: dW[y][x] = sum_i Ay[i][y] T[i][x]; T's accumulator layout (col = lane & 15, row = 4s + e) is
        // the B-operand layout, so it only needs the hi/lo split
        const u4 b2 = split_bf16(t);
        w_prev = mfma3(ray[d], b2, (f4){0.f, 0.f, 0.f, 0.f});
        cell_prev = cell;
        if (!FIT) on_prev = (int)(org[d] >> 2) % W + q < W;
        // slot d is dead from here: refill it in place, kDepth items ahead (always three loads)
        __builtin_amdgcn_sched_barrier(0);
        OVIS_FETCH(d, epre);
        __builtin_amdgcn_sched_barrier(0);
        epre = my[k + kDepth + 1];
      }
    }
    {  // drain the pipeline: the last item's footprint
      float* pp = (float*)((char*)plane + cell_prev);
      if (FIT) {
        pp[0] += w_prev.x;
        pp[W] += w_prev.y;
        pp[2 * W] += w_prev.z;
        pp[3 * W] += w_prev.w;
      } else if (on_prev) {
        const int y = (int)(cell_prev >> 2) / W;
        if (y + 0 < H) pp[0] += w_prev.x;
        if (y + 1 < H) pp[W] += w_prev.y;
        if (y + 2 < H) pp[2 * W] += w_prev.z;
        if (y + 3 < H) pp[3 * W] += w_prev.w;
      }
    }
    // the prefetches past the end still target the ring registers: retire them before anything reuses them
#pragma unroll
    for (int d = 0; d < kDepth; ++d) OVIS_WAIT3(0, rg[d], rbx[d], ray[d]);
#undef OVIS_FETCH
  }
  for (int i = lane; i < HW; i += 64) dst[i] = plane[i];
}

}  // namespace

#define OVIS_PLANE_TOO_BIG (-100)  // internal: not an ABI code, the caller falls back
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// workspace layout: list int4[batch * list_stride] | counts int[batch] (padded) | tx u4[(R * NXB + 1) * 64] | ty u4[(R * NYB + 1) * 64]
extern "C" size_t ovis_roi_align_backward_workspace_bytes(int num_rois, int batch, int height, int width) {
  if (num_rois <= 0 || batch <= 0 || height <= 0 || width <= 0) return 0;
  const size_t nxb = (size_t)ovis_ceil_div(width, kT), nyb = (size_t)ovis_ceil_div(height, kT);
  return align_up((size_t)batch * list_stride(num_rois, (int)nxb, (int)nyb) * sizeof(u4), 256) +
         align_up((size_t)batch * sizeof(int), 256) + ((size_t)num_rois * (nxb + nyb) + 2) * 64 * sizeof(u4);
}

// Whether the plane-owner kernel covers this shape (otherwise the caller falls back to the atomic kernels).
extern "C" int ovis_roi_align_backward_plane_supported(int height, int width, int pooled_h, int pooled_w) {
  return pooled_h >= 1 && pooled_w >= 4 && pooled_h <= kT && pooled_w <= kT && height < 65536 && width < 32768 &&
         (size_t)height * width * sizeof(float) <= 40 * 1024;
}

int ovis_roi_align_backward_plane_launch(const float* grad_output, const float* rois, float* grad_input,
                                         int num_rois, int batch, int channels, int height, int width,
                                         int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                         void* workspace, size_t workspace_bytes, hipStream_t s) {
  // item byte offsets are 32-bit: grad_output and the tables must each stay below 4 GiB (else: atomic path)
  if ((double)num_rois * channels * pooled_h * pooled_w * 4.0 >= 4294967296.0 ||
      ((double)num_rois * ovis_ceil_div(width, kT) + 1) * 1024.0 >= 4294967296.0 ||
      ((double)num_rois * ovis_ceil_div(height, kT) + 1) * 1024.0 >= 4294967296.0)
    return OVIS_PLANE_TOO_BIG;
  const size_t need = ovis_roi_align_backward_workspace_bytes(num_rois, batch, height, width);
  if (!workspace || workspace_bytes < need) return OVIS_ENOSPC;
  if (((uintptr_t)workspace & 255) != 0) return OVIS_EINVAL;
  const int NXB = ovis_ceil_div(width, kT), NYB = ovis_ceil_div(height, kT);
  char* w = (char*)workspace;
  u4* list = (u4*)w;
  w += align_up((size_t)batch * list_stride(num_rois, NXB, NYB) * sizeof(u4), 256);
  int* counts = (int*)w;
  w += align_up((size_t)batch * sizeof(int), 256);
  u4* tx = (u4*)w;
  u4* ty = tx + ((size_t)num_rois * NXB + 1) * 64;

  const long plan_blocks = (long)batch + ovis_ceil_div(num_rois, kPlanThreads / 64);
  const long blocks = (long)batch * channels;
  if (plan_blocks > 0x7fffffffL || blocks > 0x7fffffffL) return OVIS_ERANGE;
  hipLaunchKernelGGL(roi_bwd_plan_kernel, dim3((unsigned)plan_blocks), dim3(kPlanThreads), 0, s, rois, num_rois,
                     batch, channels, height, width, pooled_h, pooled_w, spatial_scale, sampling_ratio, list, counts, tx, ty,
                     NXB, NYB);
  OVIS_LAUNCH_CHECK();
  size_t lds = align_up((size_t)height * width * sizeof(float), 16);
  if (lds < (size_t)kMinLds) lds = kMinLds;
  if (height >= kT && width >= kT)
    hipLaunchKernelGGL(roi_bwd_mfma_kernel<true>, dim3((unsigned)blocks), dim3(64), lds, s, grad_output, list, counts,
                       tx, ty, grad_input, num_rois, batch, channels, height, width, pooled_h, pooled_w, NXB, NYB);
  else
    hipLaunchKernelGGL(roi_bwd_mfma_kernel<false>, dim3((unsigned)blocks), dim3(64), lds, s, grad_output, list,
                       counts, tx, ty, grad_input, num_rois, batch, channels, height, width, pooled_h, pooled_w, NXB,
                       NYB);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
