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
#include "roi_mfma.h"

namespace {
using namespace ovis_roi;

// Round geometry.  A round = kRI items between two workgroup barriers; its table blocks (2 KB per item) sit in one slot of a
// ring of kRing rounds in the 24 KB the planes leave of the CU's LDS: three slots of four items (tables requested two rounds
// ahead).  Two slots of six (kRI = 6, kRing = 2: a round's tables requested when the previous round
// starts, a third fewer barriers) measured 1 % faster at twelve G tiles / origins in flight, which costs scalar-register
// spills once anything else is added -- tools/experiments/patches/roi_bwd_probes.patch restores it as a build option.
constexpr int kRI = 4;        // items per round (one barrier per round)
constexpr int kGDepth = 2 * kRI;         // G tiles in flight per wave = two rounds
constexpr int kRing = 3;    // table ring depth in rounds
constexpr int kAhead = kRing - 1;        // a round's tables are requested this many rounds before it is consumed
constexpr int kRoundBytes = 2 * kRI * 1024;  // kRI tx blocks + kRI ty blocks
static_assert(kRing == 2 || kRing == 3, "ring depth");
static_assert(kRing * kRoundBytes <= 24 * 1024, "the table ring has 24 KB");
constexpr int kListPad = 4 * kGDepth;        // zero-contribution items after the last real one
constexpr int kPlanThreads = 256;
constexpr unsigned kReuseT = 0x80000000u;  // item flag, top bit of the footprint origin's cell index

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
__host__ __device__ inline long list_stride(int R, int NXB, int NYB) { return (long)R * NXB * NYB + kListPad + kGDepth; }

__global__ __launch_bounds__(kPlanThreads) void roi_bwd_plan_kernel(
    const float* __restrict__ rois, int R, int batch, int C, int H, int W, int PH, int PW, int bin_stride, float scale,
    int sampling_ratio, u4* __restrict__ list, int* __restrict__ counts, u4* __restrict__ tx,
    u4* __restrict__ ty, int NXB, int NYB, int small) {
  // small != 0 (tiles of at most 8 x 8 computed bins, gradient handed over as pre-split 256-byte tiles, see
  // roi_bwd_tiles_from_nhwc_kernel): the k-slots of the lane groups s and s + 2 mean the SAME four bins 4 (s & 1) .. + 3 --
  // the data operand carries its hi halves in groups 0, 1 and its lo halves in groups 2, 3, so ONE K = 16 product against the
  // table's hi halves (in every group) is hi.hi + lo.hi, and one against its lo halves is hi.lo + lo.lo.
  // bin_stride > 1: grad_output holds only the bins (bin_stride * i, bin_stride * j) the strided pooler produced
  // (roi_align_fwd_strided_nhwc_kernel), as [R, C, TH, TW] tiles; the tables carry those bins' weights, every other
  // bin's gradient is zero by construction and is never read.
  const int TH = (PH + bin_stride - 1) / bin_stride, TW = (PW + bin_stride - 1) / bin_stride;
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
      // A RoI's blocks are consecutive items, yb fastest.  An item's first stage depends on (RoI, xb) only -- T = G . Ax_xb --
      // so the top bit of the origin's cell index tells the main kernel that the previous item was (xb, yb - 1) of the
      // same RoI: its split first-stage result, still in registers, IS this item's -- no G split, no stage 1, no T split
      // (20 vector + 3 matrix instructions of an item's ~56; one wave-uniform branch per item pays for it).  Re-using the
      // split G tile across xb as well costs a second branch per item and measured as a loss.
      for (int xb = 0; xb < nbx; ++xb)
        for (int yb = 0; yb < nby; ++yb)
          my[o++] = (u4){small ? (unsigned)r * (unsigned)C * 256u : (unsigned)r * (unsigned)(C * TH * TW) * 4u,
                         (unsigned)(block_origin(g.wy0, yb, H) * W + block_origin(g.wx0, xb, W)) | (yb > 0 ? kReuseT : 0u),
                         (unsigned)(r * NXB + xb) * 1024u, (unsigned)(r * NYB + yb) * 1024u};
      base += all;
      __syncthreads();
    }
    // pad to a whole number of ring rounds plus the prefetch run-ahead with items that add exact zeros
    const int padded = (base + kGDepth - 1) / kGDepth * kGDepth;
    for (int i = base + threadIdx.x; i < padded + kListPad; i += kPlanThreads)
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
        // the main kernel's k-slot -> bin-column map: pulled-back last group / the same four bins in groups s and s + 2
        const int j = small ? 4 * (s & 1) + e : min(4 * s, TW - 4) + e;
        if (small ? j < TW : j >= 4 * s) v[e] = axis_weight(g.start_w, g.bin_w, g.gw, j * bin_stride, W, col);
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
        const int i = small ? 4 * (s & 1) + e : 4 * s + e;
        if (i < TH) v[e] = axis_weight(g.start_h, g.bin_h, g.gh, i * bin_stride, H, row) / g.count;
      }
    }
    const u4 sp = split_bf16(v);
    // small: the first-stage result arrives with its rows 0..7 repeated in rows 8..15 (lane groups 2, 3), so the table carries
    // its hi halves in groups 0, 1 and its lo halves in groups 2, 3: one product per half of the split T covers all four terms
    ty[((long)r * NYB + yb) * 64 + lane] = !small ? sp : (s < 2 ? (u4){sp.x, sp.y, 0u, 0u} : (u4){sp.z, sp.w, 0u, 0u});
  }
}

// ---------------------------------------------------------------------------------------------------
// Pre-split gradient tiles of the strided pooler's backward (tiles of at most 8 x 8 computed bins): the gradient arrives
// NHWC [R, TH, TW, C] fp32 from the producing data-gradient GEMM; this kernel is the layout change the plane-owner kernel
// needs anyway ([R, C, TH, TW]: one contiguous tile per (RoI, channel)) and hands every value over ALREADY SPLIT:
//   tile (r, c) = 256 bytes: hi[8][8] bf16 | lo[8][8] bf16, rows >= TH and columns >= TW zero
// so the main kernel's lanes load MFMA operands as they lie (8 bytes per lane, no hi/lo arithmetic per item and channel).
// One workgroup = (RoI, 64 channels): coalesced 256-byte row reads -> 2-byte LDS writes (tile stride 65 dwords: the 64 lanes
// of a row hit 64 banks) -> 16-byte chunks written back in address order (4 tiles = 1 KB per wave instruction).
// ---------------------------------------------------------------------------------------------------
constexpr int kTileBytes = 256;
__global__ __launch_bounds__(256) void roi_bwd_tiles_from_nhwc_kernel(const float* __restrict__ g, char* __restrict__ tiles,
                                                                     int C, int TH, int TW) {
  __shared__ unsigned lds[64 * 65];
  const int t = threadIdx.x;
  const int groups = (C + 63) / 64;
  const long r = blockIdx.x / groups;
  const int c0 = (int)(blockIdx.x % groups) * 64;
  for (int i = t; i < 64 * 65; i += 256) lds[i] = 0u;
  __syncthreads();
  const int c = t & 63;
  unsigned short* mine = (unsigned short*)(lds + c * 65);
  if (c0 + c < C) {
    for (int p = t >> 6; p < TH * TW; p += 4) {
      const float v = g[(r * TH * TW + p) * C + c0 + c];
      const unsigned hl = pack_bf16(v, 0.f);                              // low half: bf16(v)
      const unsigned lo = pack_bf16(v - __uint_as_float(hl << 16), 0.f);  // low half: bf16 of the exact remainder
      const int pos = (p / TW) * 8 + p % TW;
      mine[pos] = (unsigned short)hl;
      mine[64 + pos] = (unsigned short)lo;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int chunk = t + 256 * k, tile = chunk >> 4, part = chunk & 15;
    if (c0 + tile < C) {
      const unsigned* src = lds + tile * 65 + part * 4;
      *(u4*)(tiles + ((r * C + c0 + tile) * kTileBytes + part * 16)) = (u4){src[0], src[1], src[2], src[3]};
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Main kernel: one workgroup of NW waves owns NW consecutive channels of image n; wave w keeps the whole H x W
// gradient plane of channel c0 + w in LDS and is its only writer.
//
// What bounds this kernel is the CU's vector-memory path (64 B/clk): with every wave fetching its own copy of the
// 2 KB of table blocks per item, 3 KB per (item, channel) went through it and the kernel sat at ~0.30 ms with
// the HBM stream (0.16 ms alone) waiting behind table traffic.  So the table blocks -- identical for all
// channels -- are brought in ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) into a
// ring of three rounds of four items and read back with ds_read_b128; only the 784-byte G tile per (item,
// channel) still goes through the vector path.
//
// Per wave, per round r (items 4r .. 4r+3):
//   DMA the wave's share (8 / NW blocks) of round r+2's tables           -> ring slot (r+2) % 3
//   per item k: [plane reads of item k-1]  wait G(k)  ds_read tx, ty  stage 1 (3 MFMA)
//               [plane adds + writes of item k-1]  stage 2 (3 MFMA)  refill the G slot with item k+8
//   wait until the wave's DMAs for round r+1 have landed; s_barrier
// All vector loads of a wave complete in issue order, and the order is the same every round, so the waits are
// counted: when G(k) is needed, 7 younger G loads and the DMAs of two round starts are behind it
// (vmcnt(7 + 2P), P = 8 / NW); at the round end the DMAs for round r+1 have 8 G loads and P DMAs behind them
// (vmcnt(8 + P)) -- which are exactly the loads that would have been waited for anyway, so the deep G prefetch
// is never drained.  The loads are inline asm so that hipcc does not count them itself (left to itself it
// waits vmcnt(0) per item); rules followed (guide 5.7 form ii): G destinations are "+v", refilled in place,
// every consumer sits below a wait statement naming the register, sched_barrier pins statement order; the
// DMA statements save / restore M0 inside one statement.
// ---------------------------------------------------------------------------------------------------
#define OVIS_GLOAD4(dst, voff, sbase) \
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory")
// G tile through a raw buffer descriptor over the channel's tiles: the item's byte offset is the SCALAR offset operand, so a
// request costs no 64-bit address arithmetic (every instruction of the item loop is paid for in issue slots)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;
#define OVIS_BLOAD4(dst, voff, rsrc, soff) \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory")
#define OVIS_BLOAD2(dst, voff, rsrc, soff) \
  asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory")
#define OVIS_WAIT1(N, a) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "i"(N) : "memory")
template <bool SMALL> struct GTile { typedef f4 type; };
template <> struct GTile<true> { typedef u2 type; };

__device__ __forceinline__ void lds_dma16(const void* gsrc_lane, unsigned lds_dst_wave) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc_lane), "s"(lds_dst_wave)
      : "memory");
}

// SMALL: the gradient comes as the pre-split 256-byte tiles of roi_bwd_tiles_from_nhwc_kernel (`gout` points at them): a lane
// loads 8 bytes -- row q & 7, bins 4 (s & 1) .. + 3, hi halves in lane groups 0, 1 and lo halves in groups 2, 3 -- which IS the
// first stage's data operand; both stages are two K = 16 products (all four hi/lo terms), the only vector arithmetic left per
// item is the split of T: 39 instead of 53 instructions per item and channel.
template <int NW, bool FIT, int WC, bool SMALL = false>
__global__ __launch_bounds__(NW * 64) void roi_bwd_mfma_kernel(
    const float* __restrict__ gout, const u4* __restrict__ list, const int* __restrict__ counts,
    const u4* __restrict__ tx, const u4* __restrict__ ty, float* __restrict__ gin, int R, int batch,
    int C, int H, int W_rt, int PH, int PW, int NXB, int NYB, unsigned plane_stride) {
  // WC != 0: the map width is a compile-time constant, so the four plane rows of a footprint are immediate offsets
  // (ds_read2_b32 / ds_write2_b32 pairs, no address VALU); WC == 0: any width
  const int W = WC ? WC : W_rt;
  // no static LDS in this kernel: the dynamic segment starts at LDS address 0, which the DMA destinations rely on
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // table blocks each wave DMAs per round; when 2 * kRI is no multiple of NW the last waves request a block a second time
  // (same bytes to the same place), so that every wave has the same number of loads in flight (the waits are counted)
  constexpr int P = (2 * kRI + NW - 1) / NW;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x % batch;
  const int c_raw = (blockIdx.x / batch) * NW + wave;
  const bool live = c_raw < C;          // a partial channel group: the wave still loads, DMAs and meets barriers
  const int c = live ? c_raw : C - 1;
  const int HW = H * W;
  const int PHPW = PH * PW;
  float* plane = (float*)(smem + (size_t)wave * plane_stride);
  const unsigned ring_base = NW * plane_stride;
  for (int i = lane; i < HW; i += 64) plane[i] = 0.f;

  const int cnt = counts[n];  // padded to a multiple of kGDepth by the plan kernel
  const u4* my = list + (long)n * list_stride(R, NXB, NYB);
  const int q = lane & 15, s = lane >> 4;
  // Lane's slice of a G tile: row q, four columns starting at min(4s, PW - 4) -- the last k-group is pulled back
  // so that no lane reads past its row (the plan kernel zeroes the duplicated k-slots in tx).  Rows q >= PH re-read
  // row PH - 1: finite values that meet zero ty entries (i >= PH).
  const unsigned g_lane = SMALL ? (unsigned)((s >> 1) * 128 + (q & 7) * 16 + (s & 1) * 8)
                                : (unsigned)(min(q, PH - 1) * PW + min(4 * s, PW - 4)) * 4u;
  const unsigned long long gb = SMALL ? (unsigned long long)((const char*)gout + (long)c * kTileBytes)
                                      : (unsigned long long)(gout + (long)c * PHPW);
  const i4 rsrc = {(int)(unsigned)gb, (int)((unsigned)(gb >> 32) & 0xffffu), -1, 0x00020000};  // stride 0, no bound, 32-bit data
  const char* txl = (const char*)tx + lane * 16;
  const char* tyl = (const char*)ty + lane * 16;
  // LDS byte address of the lane's first footprint cell for an origin at cell 0 (plane base included)
  const unsigned lane_cell = (unsigned)wave * plane_stride + (unsigned)(4 * s * W + q) * 4u;

  // this wave's DMA duty for round `rr` into ring slot `slot`: blocks b = wave*P .. wave*P+P-1 (mod 2 kRI) of the round's
  // 2 kRI, block b = {tx, ty}[b / kRI] of item rr*kRI + b % kRI; lands at ring_base + slot*kRoundBytes + b*1024
  auto dma_round = [&](int rr, unsigned slot) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int b = (wave * P + p) % (2 * kRI);
      const u4 e = my[rr * kRI + b % kRI];
      const char* src = (b / kRI) ? tyl + e.w : txl + e.z;
      lds_dma16(src, ring_base + slot * kRoundBytes + (unsigned)b * 1024u);
    }
  };

  // lane's 4 grad_output values G[i = min(q, PH-1)][j = min(4s, PW-4) + e] of the item in slot d (SMALL: 4 bf16 halves)
  typename GTile<SMALL>::type rg[kGDepth];
  unsigned org[kGDepth];  // its footprint origin (cell index | reuse flag)
#pragma unroll
  for (int d = 0; d < kGDepth; ++d) rg[d] = (typename GTile<SMALL>::type)(0);
#define OVIS_FETCH(d, e)                                        \
  do {                                                          \
    if constexpr (SMALL) OVIS_BLOAD2(rg[d], g_lane, rsrc, (e).x); \
    else OVIS_BLOAD4(rg[d], g_lane, rsrc, (e).x);               \
    org[d] = (e).y;                                             \
  } while (0)

  // prologue, in the steady-state issue order: DMA G(0 .. kRI-1) DMA G(kRI .. 2 kRI-1).  Ring of three: the tables of
  // rounds 0 and 1; ring of two: round 0's tables twice (the second request keeps the count of loads in flight what the
  // counted waits of the first items assume; it rewrites slot 0 with the bytes it already holds)
  dma_round(0, 0);
#pragma unroll
  for (int d = 0; d < kRI; ++d) { const u4 e = my[d]; OVIS_FETCH(d, e); }
  dma_round(kAhead == 2 ? 1 : 0, kAhead == 2 ? 1 : 0);
#pragma unroll
  for (int d = kRI; d < kGDepth; ++d) { const u4 e = my[d]; OVIS_FETCH(d, e); }
  u4 epre = my[kGDepth];  // entry of the item to prefetch next
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(kGDepth + P) : "memory");  // round 0's tables landed
  __builtin_amdgcn_sched_barrier(0);

  // Software pipeline: the footprint of item k-1 is folded into the plane while item k runs through the matrix
  // pipe.  Starts with a zero footprint at cell 0 (adds 0.0f, FIT) / nothing (masked).
  f4 w_prev = {0.f, 0.f, 0.f, 0.f};
  u4 b2 = {0u, 0u, 0u, 0u};  // split first-stage result, kept for the RoI's further blocks of the same column strip
  unsigned cell_prev = lane_cell;
  bool on_prev = false;  // masked form only: does the lane own a column of the previous footprint
  unsigned slot = 0;     // ring slot of the round being consumed

  // cnt is a multiple of kGDepth; the list runs kListPad zero-contribution items past it
  const u4* cur = my + kGDepth;  // entries of the items the running round requests; advanced once per kGDepth items
  for (int k0 = 0; k0 < cnt; k0 += kGDepth, cur += kGDepth) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned slot2 = slot + kAhead >= kRing ? slot + kAhead - kRing : slot + kAhead;
        dma_round(k0 / kRI + half + kAhead, slot2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const char* tab = smem + ring_base + slot * kRoundBytes + lane * 16;
      // entries of the items this round requests (two rounds ahead): one base per round, constant offsets per item
      const u4* ent = cur + half * kRI;
#pragma unroll
      for (int i = 0; i < kRI; ++i) {
        const int d = half * kRI + i;
        const unsigned fl = org[d] & kReuseT;
        unsigned cell;  // (origin << 2) + lane_cell in ONE instruction; the flag bit leaves through the shift
        asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(cell) : "s"(org[d]), "v"(lane_cell));
        lds_f32* pp = (lds_f32*)(unsigned long)cell_prev;  // an LDS address as such: no symbol base to add
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (FIT) {
          v0 = pp[0]; v1 = pp[W]; v2 = pp[2 * W]; v3 = pp[3 * W];
        }
        const u4 bx = *(const u4*)(tab + i * 1024);
        const u4 ay = *(const u4*)(tab + (kRI + i) * 1024);
        __builtin_amdgcn_sched_barrier(0);
        OVIS_WAIT1(kGDepth - 1 + 2 * P, rg[d]);  // G(k) has landed
        __builtin_amdgcn_sched_barrier(0);
        // stage 1: T[i][x] = sum_j G[i][j] Ax[j][x], then its hi/lo split (T's accumulator layout -- col = lane & 15,
        // row = 4s + e -- is the B-operand layout of stage 2) -- unless the previous item (same RoI, same xb) left this
        // very operand in b2: ONE wave-uniform branch per item
        if (!(fl & kReuseT)) {
          if constexpr (SMALL) {
            b2 = split_bf16(mfma2x2(rg[d], rg[d], (u2){bx.x, bx.y}, (u2){bx.z, bx.w}));
          } else {
            const u4 a1 = split_bf16(rg[d]);
            b2 = split_bf16(mfma3(a1, bx, (f4){0.f, 0.f, 0.f, 0.f}));
          }
        }
        // previous item's footprint (its reads were issued above, their latency is behind stage 1 by now)
        if (FIT) {
          pp[0] = v0 + w_prev.x;
          pp[W] = v1 + w_prev.y;
          pp[2 * W] = v2 + w_prev.z;
          pp[3 * W] = v3 + w_prev.w;
        } else if (on_prev) {  // maps narrower / lower than one block: mask the footprint
          const int y = (int)((cell_prev - (unsigned)wave * plane_stride) >> 2) / W;
          if (y + 0 < H) pp[0] += w_prev.x;
          if (y + 1 < H) pp[W] += w_prev.y;
          if (y + 2 < H) pp[2 * W] += w_prev.z;
          if (y + 3 < H) pp[3 * W] += w_prev.w;
        }
        // stage 2: dW[y][x] = sum_i Ay[i][y] T[i][x]
        if constexpr (SMALL) w_prev = mfma2x2((u2){ay.x, ay.y}, (u2){ay.x, ay.y}, (u2){b2.x, b2.y}, (u2){b2.z, b2.w});
        else w_prev = mfma3(ay, b2, (f4){0.f, 0.f, 0.f, 0.f});
        cell_prev = cell;
        if (!FIT) on_prev = (int)(org[d] & ~kReuseT) % W + q < W;
        // G slot d is dead from here: refill it in place, kGDepth items ahead
        __builtin_amdgcn_sched_barrier(0);
        OVIS_FETCH(d, epre);
        __builtin_amdgcn_sched_barrier(0);
        epre = ent[i + 1];
      }
      // the wave's DMAs for the next round have landed (nothing younger than G(k+1) is forced); all waves have
      // finished reading this round's slot once they pass the barrier
      __builtin_amdgcn_sched_barrier(0);
      // the tables of the next round were requested kAhead round starts ago: what may still be in flight behind them are
      // the G refills since (kRI per round) and the table requests of the round starts after theirs
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(kRI * kAhead + (kAhead - 1) * P) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 >= kRing ? 0 : slot + 1;
    }
  }
  {  // drain the pipeline: the last item's footprint
    lds_f32* pp = (lds_f32*)(unsigned long)cell_prev;
    if (FIT) {
      pp[0] += w_prev.x;
      pp[W] += w_prev.y;
      pp[2 * W] += w_prev.z;
      pp[3 * W] += w_prev.w;
    } else if (on_prev) {
      const int y = (int)((cell_prev - (unsigned)wave * plane_stride) >> 2) / W;
      if (y + 0 < H) pp[0] += w_prev.x;
      if (y + 1 < H) pp[W] += w_prev.y;
      if (y + 2 < H) pp[2 * W] += w_prev.z;
      if (y + 3 < H) pp[3 * W] += w_prev.w;
    }
  }
  // the prefetches past the end still target the G registers / the ring: retire them before anything is reused
#pragma unroll
  for (int d = 0; d < kGDepth; ++d) OVIS_WAIT1(0, rg[d]);
#undef OVIS_FETCH
  if (live) {
    float* dst = gin + ((long)n * C + c) * HW;
    for (int i = lane; i < HW; i += 64) dst[i] = plane[i];
  }
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

// Waves (= channels) per workgroup for an H x W plane: all planes of the group plus the table ring must fit the
// CU's 160 KB of LDS.  0 = the plane is too large for this kernel.
static int plane_waves(int height, int width) {
  const size_t stride = align_up((size_t)height * width * sizeof(float), 16);
  const size_t room = 160 * 1024 - (size_t)kRing * kRoundBytes;
  if (8 * stride <= room) return 8;
  if (4 * stride <= room) return 4;
  return 0;
}

// The strided pooler's backward from an NHWC gradient (pre-split tiles, SMALL kernel): plan workspace + one 256-byte tile per
// (RoI, channel)
extern "C" size_t ovis_roi_align_backward_strided_nhwc_workspace_bytes(int num_rois, int batch, int channels, int height,
                                                                       int width) {
  const size_t plan = ovis_roi_align_backward_workspace_bytes(num_rois, batch, height, width);
  if (plan == 0 || channels <= 0) return 0;
  return align_up(plan, 256) + (size_t)num_rois * channels * kTileBytes;
}

// Whether the plane-owner kernel covers this shape (otherwise the caller falls back to the atomic kernels).
extern "C" int ovis_roi_align_backward_plane_supported(int height, int width, int pooled_h, int pooled_w) {
  return pooled_h >= 1 && pooled_w >= 4 && pooled_h <= kT && pooled_w <= kT && height < 65536 && width < 32768 &&
         plane_waves(height, width) != 0;
}

int ovis_roi_align_backward_plane_launch(const float* grad_output, const float* rois, float* grad_input,
                                         int num_rois, int batch, int channels, int height, int width,
                                         int pooled_h, int pooled_w, int bin_stride, float spatial_scale,
                                         int sampling_ratio, void* workspace, size_t workspace_bytes, hipStream_t s,
                                         int nhwc_small) {
  // nhwc_small: grad_output is the NHWC gradient [num_rois, tile_h, tile_w, channels] of at most 8 x 8 computed bins; it is
  // re-laid into pre-split tiles behind the plan workspace and the SMALL kernel consumes those
  const int tile_h = (pooled_h + bin_stride - 1) / bin_stride, tile_w = (pooled_w + bin_stride - 1) / bin_stride;
  if (nhwc_small && (tile_h > 8 || tile_w > 8 || !(height >= kT && width >= kT) || plane_waves(height, width) != 8))
    return OVIS_PLANE_TOO_BIG;
  // item byte offsets are 32-bit: grad_output and the tables must each stay below 4 GiB (else: atomic path)
  if ((double)num_rois * channels * (nhwc_small ? 64.0 : (double)tile_h * tile_w) * 4.0 >= 4294967296.0 ||
      ((double)num_rois * ovis_ceil_div(width, kT) + 1) * 1024.0 >= 4294967296.0 ||
      ((double)num_rois * ovis_ceil_div(height, kT) + 1) * 1024.0 >= 4294967296.0)
    return OVIS_PLANE_TOO_BIG;
  const size_t plan_bytes = ovis_roi_align_backward_workspace_bytes(num_rois, batch, height, width);
  const size_t need = nhwc_small ? ovis_roi_align_backward_strided_nhwc_workspace_bytes(num_rois, batch, channels, height, width)
                                 : plan_bytes;
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

  const int nw = plane_waves(height, width);
  const unsigned stride = (unsigned)align_up((size_t)height * width * sizeof(float), 16);
  const size_t lds = (size_t)nw * stride + (size_t)kRing * kRoundBytes;
  const long plan_blocks = (long)batch + ovis_ceil_div(num_rois, kPlanThreads / 64);
  const long blocks = (long)batch * ovis_ceil_div(channels, nw);
  if (plan_blocks > 0x7fffffffL || blocks > 0x7fffffffL) return OVIS_ERANGE;
  hipLaunchKernelGGL(roi_bwd_plan_kernel, dim3((unsigned)plan_blocks), dim3(kPlanThreads), 0, s, rois, num_rois,
                     batch, channels, height, width, pooled_h, pooled_w, bin_stride, spatial_scale, sampling_ratio, list,
                     counts, tx, ty, NXB, NYB, nhwc_small);
  OVIS_LAUNCH_CHECK();
  if (nhwc_small) {
    char* tiles = (char*)workspace + align_up(plan_bytes, 256);
    const long tblocks = (long)num_rois * ovis_ceil_div(channels, 64);
    if (tblocks > 0x7fffffffL) return OVIS_ERANGE;
    hipLaunchKernelGGL(roi_bwd_tiles_from_nhwc_kernel, dim3((unsigned)tblocks), dim3(256), 0, s, grad_output, tiles, channels,
                       tile_h, tile_w);
    OVIS_LAUNCH_CHECK();
    grad_output = (const float*)tiles;
  }
  static bool attr_set = false;
  if (!attr_set) {
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)roi_bwd_mfma_kernel<8, true, 84>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)roi_bwd_mfma_kernel<8, true, 84, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)roi_bwd_mfma_kernel<8, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)roi_bwd_mfma_kernel<8, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)roi_bwd_mfma_kernel<8, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)roi_bwd_mfma_kernel<4, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)roi_bwd_mfma_kernel<4, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const bool fit = height >= kT && width >= kT;
#define OVIS_BWD_LAUNCH(NW_, FIT_, WC_)                                                                                  \
  hipLaunchKernelGGL((roi_bwd_mfma_kernel<NW_, FIT_, WC_>), dim3((unsigned)blocks), dim3(NW_ * 64), lds, s, grad_output, \
                     list, counts, tx, ty, grad_input, num_rois, batch, channels, height, width, tile_h, tile_w,          \
                     NXB, NYB, stride)
  if (nhwc_small) {
#define OVIS_BWD_LAUNCH_SMALL(WC_)                                                                                           \
  hipLaunchKernelGGL((roi_bwd_mfma_kernel<8, true, WC_, true>), dim3((unsigned)blocks), dim3(8 * 64), lds, s, grad_output,  \
                     list, counts, tx, ty, grad_input, num_rois, batch, channels, height, width, tile_h, tile_w, NXB, NYB, \
                     stride)
    if (width == 84) OVIS_BWD_LAUNCH_SMALL(84); else OVIS_BWD_LAUNCH_SMALL(0);
#undef OVIS_BWD_LAUNCH_SMALL
  } else if (nw == 8) {
    if (fit && width == 84) OVIS_BWD_LAUNCH(8, true, 84);  // the C4 map of an 800 x 1333 batch
    else if (fit) OVIS_BWD_LAUNCH(8, true, 0);
    else OVIS_BWD_LAUNCH(8, false, 0);
  } else {
    if (fit) OVIS_BWD_LAUNCH(4, true, 0); else OVIS_BWD_LAUNCH(4, false, 0);
  }
#undef OVIS_BWD_LAUNCH
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
