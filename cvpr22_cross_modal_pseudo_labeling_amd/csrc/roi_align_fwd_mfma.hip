// RoIAlign forward for gfx950 (MI355X) on the matrix cores.
//
// Reference semantics: maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:16-122 (one thread per output bin, four
// gathered taps per sample).  Here the pooled tile of one (RoI, channel) is two small matrix products,
//     V   = Ay . win_c (/ count)  [bin rows x window columns]      (v_mfma_f32_16x16x16_bf16 x3)
//     out = V . Ax^T              [bin rows x bin columns]         (x3; V's accumulator layout IS the B layout)
// with Ax / Ay the separable bilinear-average weights of the RoI (roi_mfma.h), window cut into 16 x 16 blocks.
// Values are the reference's up to the bf16 hi/lo split of the operands (~1e-5 relative to sum |w . x|; the
// bit-exact kernel in roi_align.hip stays available as ovis_roi_align_forward_f32).
//
//  * plan kernel: one wave per RoI writes the table blocks once, in MFMA operand lane order;
//  * main kernel: a 4-wave workgroup owns (RoI, 64 channels); every wave loads the RoI's table blocks into
//    registers ONCE and walks its 16 channels: the feature block goes straight from L2 into the operand layout
//    (lane = (window column, 4 consecutive rows): four dword loads whose quarter-waves read 64 contiguous bytes,
//    next channel's loads in flight while the current one is in the matrix pipe), split to bf16 hi/lo in
//    registers, 6 MFMAs, one 16-byte streaming store per lane (4 consecutive bin columns of a bin row).
//    No LDS, no gathers; workgroups are ordered channel-slab major so the co-resident ones read the same 64 planes
//    from the XCD L2s, and HBM sees the feature map once and the 0.8 MB / RoI output once.
#include "ovis_common.h"
#include "roi_mfma.h"

namespace {
using namespace ovis_roi;

constexpr int kFwdWaves = 4;
constexpr int kSlab = 64;  // channels per workgroup
typedef f4 f4u __attribute__((aligned(4)));

struct FwdHdr {
  int wy0, wx0;    // window origin (feature cells)
  int nyb, nxb;    // 16-cell blocks per axis; 0 = every sample falls outside the map (output is zeros)
  int wy1, wx1;    // last window row / column
  int pad0, pad1;
};

// fx[r * NXB + xb][lane] = A operand of stage 2: lane (bin column j = lane & 15, k-group s) holds Ax[j][ox + 4s + e]
// fy[r * NYB + yb][lane] = B operand of stage 1: lane (bin row    i = lane & 15, k-group s) holds Ay[i][oy + 4s + e] / count
// (zero outside the cells the block owns and for j >= PW / i >= PH), each as bf16 hi x4 | bf16 lo x4.
__global__ __launch_bounds__(256) void roi_fwd_plan_kernel(const float* __restrict__ rois, int R, int batch, int H,
                                                           int W, int PH, int PW, float scale, int sampling_ratio,
                                                           FwdHdr* __restrict__ hdr, u4* __restrict__ fx,
                                                           u4* __restrict__ fy, int NXB, int NYB) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const RoiGeom g = make_geom(rois + (long)r * 5, scale, H, W, PH, PW, sampling_ratio, batch);
  const int nxb = g.empty ? 0 : (g.wx1 - g.wx0 + kT) / kT, nyb = g.empty ? 0 : (g.wy1 - g.wy0 + kT) / kT;
  if (lane == 0) hdr[r] = FwdHdr{g.wy0, g.wx0, nyb, nxb, g.wy1, g.wx1, 0, 0};
  const int q = lane & 15, s = lane >> 4;
  for (int xb = 0; xb < nxb; ++xb) {
    const int ox = block_origin(g.wx0, xb, W);
    const int lo = g.wx0 + xb * kT, hi = min(lo + kT - 1, g.wx1);
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < PW) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = ox + 4 * s + e;
        if (col >= lo && col <= hi) v[e] = axis_weight(g.start_w, g.bin_w, g.gw, q, W, col);
      }
    }
    fx[((long)r * NXB + xb) * 64 + lane] = split_bf16(v);
  }
  for (int yb = 0; yb < nyb; ++yb) {
    const int oy = block_origin(g.wy0, yb, H);
    const int lo = g.wy0 + yb * kT, hi = min(lo + kT - 1, g.wy1);
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < PH) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = oy + 4 * s + e;
        if (row >= lo && row <= hi) v[e] = axis_weight(g.start_h, g.bin_h, g.gh, q, H, row) / g.count;
      }
    }
    fy[((long)r * NYB + yb) * 64 + lane] = split_bf16(v);
  }
}

// Vector loads and stores share ONE in-order counter (vmcnt) on gfx950, and hipcc cannot count stores that sit in
// exec-masked regions: left to itself it waited vmcnt(0) for the prefetched feature block of the next channel, i.e.
// for the ~1 us write acknowledgement of the current channel's tile, every iteration (measured: loads alone 0.19 ms,
// stores alone 0.21 ms, together 0.41 ms).  So both are inline asm here and counted by hand (guide 5.7 form ii): per
// channel exactly 4 * NY * NX dword loads into `nxt` and NST stores are issued, in that order, and the wait for
// `nxt` at the end of the iteration is vmcnt(NST) -- the stores stay in flight.
#define OVIS_FLOAD(dst, voff, sbase) \
  asm volatile("global_load_dword %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory")

// Number of store instructions per pooled tile: one 16-byte store for the k-groups with 4 columns left, one
// narrower store for the k-group that holds the row's tail (PW % 4 columns).
__host__ __device__ constexpr int tile_stores(int PW) { return (PW >= 4 ? 1 : 0) + (PW % 4 != 0 ? 1 : 0); }

// The chain is arranged (fwd_channels) so that a lane ends up with FOUR CONSECUTIVE bin columns of one bin row:
// accumulator (col = lane & 15 = bin row i, rows 4s + e = bin columns j).  The PW-float rows of the pooled tile go
// out as 16-byte stores plus one tail store (PW = 14: 8 bytes from the last k-group).
__device__ __forceinline__ void store_tile_row(float* __restrict__ tile /* uniform */, f4 acc, int q, int s, int PH,
                                               int PW) {
  const unsigned voff = (unsigned)(q * PW + 4 * s) * 4u;
  const int n = PW - 4 * s;  // columns left in this row from the lane's first one
  const bool row_ok = q < PH;
  if (row_ok && n >= 4)
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(acc), "s"(tile) : "memory");
  const int tail = PW & 3;
  if (tail == 3) {
    if (row_ok && n == 3) {
      typedef float f3 __attribute__((ext_vector_type(3)));
      const f3 a3 = {acc.x, acc.y, acc.z};
      asm volatile("global_store_dwordx3 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(a3), "s"(tile) : "memory");
    }
  } else if (tail == 2) {
    if (row_ok && n == 2) {
      const f2 a2 = {acc.x, acc.y};
      asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(a2), "s"(tile) : "memory");
    }
  } else if (tail == 1) {
    if (row_ok && n == 1) asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(acc.x), "s"(tile) : "memory");
  }
}

// The channels of a RoI whose window is NY x NX blocks (tables already in registers).  Per channel:
//   V^T[x][i]   = sum_y F[y][x] Ay[i][y]       A = F^T (lane: column x, rows 4s+e), B = fy  -> acc (col i, rows x = 4s+e)
//   out^T[j][i] = sum_x Ax[j][x] V^T[x][i]     A = fx, B = V^T = that accumulator         -> acc (col i, rows j = 4s+e)
// Lane's slice of a feature block: column x = lane & 15, rows 4s .. 4s+3 -- four dword loads whose quarter-waves
// each read 64 contiguous bytes of one feature row.  (The other orientation, one 16-byte load of row q / columns
// 4s..4s+3, makes every quarter-wave touch 16 different rows: ~100 L1 accesses per load instruction.)
template <int NY, int NX, int NST>
__device__ __forceinline__ void fwd_channels(const float* __restrict__ img_c0, int HW, int W,
                                             const int (&cell)[NY][NX][4], const u4 (&bx)[NX], const u4 (&ay)[NY],
                                             float* __restrict__ out_c0, int PHPW, int PW, int PH, int nch,
                                             int lane) {
  const int q = lane & 15, s = lane >> 4;
  unsigned off[NY][NX][4];  // byte offsets of the lane's four feature cells inside a plane (loop-invariant)
  float nxt[NY][NX][4];
#pragma unroll
  for (int y = 0; y < NY; ++y)
#pragma unroll
    for (int x = 0; x < NX; ++x)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        off[y][x][e] = (unsigned)cell[y][x][e] * 4u;
        nxt[y][x][e] = 0.f;
      }
  auto issue = [&](const float* plane) {
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int x = 0; x < NX; ++x)
#pragma unroll
        for (int e = 0; e < 4; ++e) OVIS_FLOAD(nxt[y][x][e], off[y][x][e], plane);
  };
  issue(img_c0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int c = 0; c < nch; c += kFwdWaves) {
    f4 cur[NY][NX];
    // the wait naming every nxt register: nothing below may read them above this point
#pragma unroll
    for (int y = 0; y < NY; ++y)
#pragma unroll
      for (int x = 0; x < NX; ++x) {
        asm volatile("" : "+v"(nxt[y][x][0]), "+v"(nxt[y][x][1]), "+v"(nxt[y][x][2]), "+v"(nxt[y][x][3]));
        cur[y][x] = (f4){nxt[y][x][0], nxt[y][x][1], nxt[y][x][2], nxt[y][x][3]};
      }
    __builtin_amdgcn_sched_barrier(0);
    // next channel of this wave (the last iteration re-reads its own plane: the instruction count stays fixed)
    issue(img_c0 + (long)(c + kFwdWaves < nch ? c + kFwdWaves : c) * HW);
    __builtin_amdgcn_sched_barrier(0);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      f4 vt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int y = 0; y < NY; ++y) vt = mfma3(split_bf16(cur[y][x]), ay[y], vt);
      acc = mfma3(bx[x], split_bf16(vt), acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    store_tile_row(out_c0 + (long)c * PHPW, acc, q, s, PH, PW);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NST) : "memory");  // nxt landed; this channel's stores stay in flight
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ f4 load_block(const float* __restrict__ p, int W) {  // generic (large-window) path
  return (f4){p[0], p[W], p[2 * W], p[3 * W]};
}

template <int NY, int NX, int NST>
__device__ __forceinline__ void fwd_roi(const float* __restrict__ img_c0, int HW, int H, int W, const FwdHdr& h,
                                        const u4* __restrict__ fxr, const u4* __restrict__ fyr,
                                        float* __restrict__ out_c0, int PHPW, int PW, int PH, int nch, int lane) {
  // Lane's four feature cells per block: column ox + q, rows oy + 4s + e -- clamped into the cells the block owns
  // (a clamped lane re-reads a cell some other lane needs anyway; its table entry is zero), so the 16 x 16
  // footprint costs no cache lines beyond the RoI's actual window.
  const int q = lane & 15, s = lane >> 4;
  int cell[NY][NX][4];
  u4 bx[NX], ay[NY];
#pragma unroll
  for (int x = 0; x < NX; ++x) bx[x] = fxr[x * 64 + lane];
#pragma unroll
  for (int y = 0; y < NY; ++y) ay[y] = fyr[y * 64 + lane];
#pragma unroll
  for (int y = 0; y < NY; ++y)
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      const int col = min(max(block_origin(h.wx0, x, W) + q, h.wx0 + x * kT), min(h.wx0 + x * kT + kT - 1, h.wx1));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = min(max(block_origin(h.wy0, y, H) + 4 * s + e, h.wy0 + y * kT),
                            min(h.wy0 + y * kT + kT - 1, h.wy1));
        cell[y][x][e] = row * W + col;
      }
    }
  fwd_channels<NY, NX, NST>(img_c0, HW, W, cell, bx, ay, out_c0, PHPW, PW, PH, nch, lane);
}

template <int NST>
__global__ __launch_bounds__(kFwdWaves * 64) void roi_fwd_mfma_kernel(
    const float* __restrict__ in, const float* __restrict__ rois, const FwdHdr* __restrict__ hdr,
    const u4* __restrict__ fx, const u4* __restrict__ fy, float* __restrict__ out, int R, int batch, int C, int H,
    int W, int PH, int PW, int NXB, int NYB) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.x % R;  // channel-slab major: co-resident workgroups read the same planes
  const int c0 = (blockIdx.x / R) * kSlab + wave;
  const int HW = H * W, PHPW = PH * PW;
  const FwdHdr h = hdr[r];
  const int nch = min(kSlab, C - (c0 - wave)) - wave;  // this wave takes channels c0, c0 + 4, ... of the slab
  if (nch <= 0) return;
  float* out_c0 = out + ((long)r * C + c0) * PHPW;
  if (h.nxb == 0) {  // every sample is out of range (or the RoI is malformed): the reference emits zeros
    for (int c = 0; c < nch; c += kFwdWaves)
      for (int i = lane; i < PHPW; i += 64) out_c0[(long)c * PHPW + i] = 0.f;
    return;
  }
  const int b = (int)rois[(long)r * 5];
  const float* img_c0 = in + ((long)b * C + c0) * HW;
  const u4* fxr = fx + (long)r * NXB * 64;
  const u4* fyr = fy + (long)r * NYB * 64;
  if (h.nyb == 1 && h.nxb == 1) return fwd_roi<1, 1, NST>(img_c0, HW, H, W, h, fxr, fyr, out_c0, PHPW, PW, PH, nch, lane);
  if (h.nyb == 1 && h.nxb == 2) return fwd_roi<1, 2, NST>(img_c0, HW, H, W, h, fxr, fyr, out_c0, PHPW, PW, PH, nch, lane);
  if (h.nyb == 2 && h.nxb == 1) return fwd_roi<2, 1, NST>(img_c0, HW, H, W, h, fxr, fyr, out_c0, PHPW, PW, PH, nch, lane);
  if (h.nyb == 2 && h.nxb == 2) return fwd_roi<2, 2, NST>(img_c0, HW, H, W, h, fxr, fyr, out_c0, PHPW, PW, PH, nch, lane);
  // larger windows (RoIs beyond ~500 px): block loops with the tables re-read per block (L1-resident)
  const int q = lane & 15, s = lane >> 4;
  const int f_lane = 4 * s * W + q;
  for (int c = 0; c < nch; c += kFwdWaves) {
    const float* pc = img_c0 + (long)c * HW + f_lane;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int x = 0; x < h.nxb; ++x) {
      const int ox = block_origin(h.wx0, x, W);
      f4 vt = {0.f, 0.f, 0.f, 0.f};
      for (int y = 0; y < h.nyb; ++y)
        vt = mfma3(split_bf16(load_block(pc + block_origin(h.wy0, y, H) * W + ox, W)), fyr[y * 64 + lane], vt);
      acc = mfma3(fxr[x * 64 + lane], split_bf16(vt), acc);
    }
    if (q < PH) {  // compiler-counted stores here: this path mixes them with compiler-counted loads
      float* p = out_c0 + (long)c * PHPW + q * PW + 4 * s;
      if (4 * s + 0 < PW) p[0] = acc.x;
      if (4 * s + 1 < PW) p[1] = acc.y;
      if (4 * s + 2 < PW) p[2] = acc.z;
      if (4 * s + 3 < PW) p[3] = acc.w;
    }
  }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// workspace layout: hdr FwdHdr[R] (padded) | fx u4[R * NXB * 64] | fy u4[R * NYB * 64]
extern "C" size_t ovis_roi_align_forward_workspace_bytes(int num_rois, int height, int width) {
  if (num_rois <= 0 || height <= 0 || width <= 0) return 0;
  const size_t nxb = (size_t)ovis_ceil_div(width, kT), nyb = (size_t)ovis_ceil_div(height, kT);
  return align_up((size_t)num_rois * sizeof(FwdHdr), 256) + (size_t)num_rois * (nxb + nyb) * 64 * sizeof(u4);
}

// Shapes the matrix-core forward covers: pooled sizes up to one MFMA tile, maps of at least one 16 x 16 block.
extern "C" int ovis_roi_align_forward_mfma_supported(int height, int width, int pooled_h, int pooled_w) {
  return pooled_h >= 1 && pooled_w >= 1 && pooled_h <= kT && pooled_w <= kT && height >= kT && width >= kT;
}

int ovis_roi_align_forward_mfma_launch(const float* input, const float* rois, float* output, int num_rois, int batch,
                                       int channels, int height, int width, int pooled_h, int pooled_w,
                                       float spatial_scale, int sampling_ratio, void* workspace,
                                       size_t workspace_bytes, hipStream_t s) {
  const size_t need = ovis_roi_align_forward_workspace_bytes(num_rois, height, width);
  if (!workspace || workspace_bytes < need) return OVIS_ENOSPC;
  if (((uintptr_t)workspace & 255) != 0) return OVIS_EINVAL;
  const int NXB = ovis_ceil_div(width, kT), NYB = ovis_ceil_div(height, kT);
  char* w = (char*)workspace;
  FwdHdr* hdr = (FwdHdr*)w;
  w += align_up((size_t)num_rois * sizeof(FwdHdr), 256);
  u4* fx = (u4*)w;
  u4* fy = fx + (size_t)num_rois * NXB * 64;
  const long blocks = (long)num_rois * ovis_ceil_div(channels, kSlab);
  if (blocks > 0x7fffffffL) return OVIS_ERANGE;
  hipLaunchKernelGGL(roi_fwd_plan_kernel, dim3((unsigned)ovis_ceil_div(num_rois, 4)), dim3(256), 0, s, rois, num_rois,
                     batch, height, width, pooled_h, pooled_w, spatial_scale, sampling_ratio, hdr, fx, fy, NXB, NYB);
  OVIS_LAUNCH_CHECK();
  if (tile_stores(pooled_w) == 2)
    hipLaunchKernelGGL(roi_fwd_mfma_kernel<2>, dim3((unsigned)blocks), dim3(kFwdWaves * 64), 0, s, input, rois, hdr,
                       fx, fy, output, num_rois, batch, channels, height, width, pooled_h, pooled_w, NXB, NYB);
  else
    hipLaunchKernelGGL(roi_fwd_mfma_kernel<1>, dim3((unsigned)blocks), dim3(kFwdWaves * 64), 0, s, input, rois, hdr,
                       fx, fy, output, num_rois, batch, channels, height, width, pooled_h, pooled_w, NXB, NYB);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
