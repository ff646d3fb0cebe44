// fp32-accurate GEMM / implicit-GEMM convolution on the bf16 matrix cores of gfx950, operands in "pair" layout.
//
//   C[M, N] = epilogue( sum_k A[m, k] * B[n, k] ),   A, B fp32 values carried as bf16 hi + lo
//
// x = hi + lo + O(2^-17 |x|), hi = bf16(x), lo = bf16(x - hi); the product keeps hi.hi + hi.lo + lo.hi, accumulated
// in fp32 by v_mfma_f32_16x16x32_bf16 (relative error ~4e-6, cf. 1.7e-6 for an fp32 GEMM).  The three products share
// their operand fragments: per 32-deep k-step a wave reads 4 fragment kinds (A_hi, A_lo, B_hi, B_lo) from LDS and
// issues 3 MFMAs per fragment pair, so the kernel moves 2/3 of the bytes and 4/9 of the LDS reads per MFMA of a plain
// bf16 GEMM over 3K -- that is what lets a simple one-barrier-per-k-step structure keep the matrix cores busy.
//
// PAIR LAYOUT (written by split_pair_kernel / the poolers / this kernel's epilogue): a row of K values (K % 32 == 0)
// is K/32 blocks of 128 bytes: [ hi(32 x bf16) | lo(32 x bf16) ].  One k-step of one row is one full 128-byte line,
// fetched by 8 lanes of a global_load_lds_dwordx4 (no VGPR round trip).
//
// Three ways the A operand is addressed (template parameter MODE):
//   PLAIN   : A is [M, ch] (+ an optional SECOND operand [M, ch2] whose k-steps follow: K = ch + ch2 against
//             B = [N, ch + ch2] -- conv3 and the projection shortcut of a bottleneck as ONE GEMM, no shortcut tensor).
//   SHIFTED : implicit convolution, any map size.  A is an NHWC tensor [R, H, W, ch]; k = (tap, channel block): row
//             m = (r, y, x) reads pixel (y + dy, x + dx) of tap (dy, dx) ("same" zero padding, stride 1; `flip`
//             negates the offsets = the data-gradient convolution); rows outside the map read a 128-byte zero line.
//   HALO    : implicit convolution on SMALL maps (every tap shift |dy*W + dx| <= 8 rows: the 7x7 maps of the res5
//             head).  k = (channel block, tap): the rows [m0 - 8, m0 + BM + 8) of one channel block are staged ONCE
//             and the 9 taps read their fragments from that halo tile at shifted LDS rows (out-of-map lanes read a
//             zero row kept in LDS) -- the A bytes that cross the L2 -> LDS path drop 9x (32 -> 18 KB per k-step),
//             which is what bounds the SHIFTED form.
// No im2col matrix exists anywhere.
//
// Tile: (WM x 64) x (WN x 64) per workgroup of WM*WN waves, wave tile 64 x 64 (4 x 4 MFMA tiles, 64 accumulator
// VGPRs); LDS stages of BM x 128 B (A) + BN x 128 B (B); the 16-byte chunk index of a row is XOR-ed with
// row & 7 (applied to the SOURCE address of the LDS-DMA, so the LDS image stays lane-linear) which makes every
// ds_read_b128 of a 16-row fragment conflict-free -- at ANY row offset, which the halo-tile 3x3 needs: its taps read the
// fragments dy * W + dx rows further; the earlier (row >> 1) & 7 was conflict-free only at offsets = 0 mod 4 and cost that
// kernel 2x the LDS cycles on six of its nine taps (tools/microbench/lds_shift_probe.hip).  NS = 2: the next stage loads under this stage's MFMAs, two
// workgroups per CU (small grids).  NS = 1: one stage, nothing overlaps inside a workgroup, three / four independent
// workgroups per CU hide each other's load phases (large grids: +7...17 %).
// Epilogue: the accumulators go through a wave-private LDS staging tile (16 rows x 64 columns, slots XOR-swizzled by
// row) so that bias / shortcut / gate reads and the fp32 / pair stores touch whole 256-byte row segments (16 lanes x
// 16 B per row) instead of 64-byte (fp32) and 32-byte (pair) pieces of 16 different rows per instruction.
// Small-M problems (a handful of workgroups, long K) are cut along K into slices whose raw partial tiles go to fp32
// slabs; split_gemm_finish_kernel sums them and applies the epilogue (deterministic, no atomics).
// Workgroups are renumbered so that each XCD owns a contiguous range of tiles (column tiles of one row tile are
// co-resident on one L2: the A rows are fetched from HBM once).
#include <stdlib.h>

#include "ovis_common.h"
constexpr int kGroupWidth = 4;   // column tiles per group of the tile order (16 measured slower, profiles/r4_column_group_ab.txt)

namespace {
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ char g_zero_line[128];  // zero-initialised: the line an out-of-map tap reads

enum { PLAIN = 0, SHIFTED = 1, HALO = 2, DEFORM = 3 };
constexpr int kHalo = 8;  // rows staged on either side of a HALO tile

struct SplitGemmArgs {
  const char* A; long a_rs;          // pair rows of `ch` values (bytes per row = 4 * ch for a dense tensor)
  const char* A2; long a2_rs;        // PLAIN: optional second operand, pair rows of `ch2` values (k-steps after A's)
  const char* B; long b_rs;          // [N] pair rows of K = T * ch (+ ch2) values
  float* C; long ldc;                // fp32 result (may be null)
  char* Cp; long cp_rs;              // pair result (may be null), bytes per row
  const float* bias; const float* res; long ldr;
  const char* resp; long resp_rs;    // shortcut given in PAIR layout (hi + lo is exact in fp32): the block input a bottleneck
                                     // already holds as its conv1 operand -- no fp32 copy of it has to exist
  const char* gate; long gate_rs;    // optional ReLU gate of a backward pass: pair rows of the forward activation (hi > 0)
  float* pool; int pool_rows; float pool_scale;  // POOL kernels: pool[m / pool_rows][n] += pool_scale * (final value)
  float* slab;                       // split-K: raw partial sums [kslices][M][N] (then C / Cp / bias / ... are unused here)
  long M; int N; int ch; int ch2; int T; int H; int W; int KH; int KW; int flip; int relu; int gw;
  int kslices; int steps_per_slice;  // k-steps (PLAIN / SHIFTED) or channel blocks (HALO) per slice
  // DEFORM (deformable convolution as an implicit GEMM, no column buffer): A is produced by bilinear sampling of an
  // NHWC fp32 image [B, H, W, ch] at (ho * stride - pad + ky * dil + offset) instead of being read; rows = B * Ho * Wo
  const float* dimg; const float* doff; const float* dmask;  // offsets [B, dg*2*T, Ho, Wo], mask [B, dg*T, Ho, Wo] or null
  int Ho; int Wo; int sh; int sw; int ph; int pw; int dlh; int dlw; int dg;
};

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

// Epilogue operand of 4 consecutive columns of one row, loaded ahead of its use.  A launch has at most ONE kind (the
// launcher checks): the fp32 shortcut (a | b = its 16 bytes); a shortcut given in pair layout (a = hi halves, b = lo
// halves, 64 bytes further); the hi halves of a ReLU gate (a).  All three are fetched by the same two 8-byte loads from
// a base / row stride / column offset / second-load offset chosen once per kernel (EpilogueSource), so that the loads
// are straight-line code whose results nothing touches before epilogue_apply4 -- a per-kind branch around them made the
// compiler wait for every load where it was issued.
struct EpilogueOperands {
  uint2 a, b;
};

struct EpilogueSource {
  const char* base;   // null: no operand
  long row_bytes;
  int second;         // byte offset of the second 8-byte load
};

__device__ __forceinline__ EpilogueSource epilogue_source(const SplitGemmArgs& p) {
  EpilogueSource s;
  s.base = p.res ? (const char*)p.res : (p.resp ? p.resp : p.gate);
  s.row_bytes = p.res ? p.ldr * 4 : (p.resp ? p.resp_rs : p.gate_rs);
  s.second = p.res ? 8 : (p.resp ? 64 : 0);
  return s;
}

// column byte offset of columns n..n+3 inside an operand row
__device__ __forceinline__ long epilogue_column_bytes(const SplitGemmArgs& p, int n, long poff) { return p.res ? (long)n * 4 : poff; }

__device__ __forceinline__ void epilogue_fetch4(const EpilogueSource& s, long m, long column_bytes, EpilogueOperands& o) {
  const char* q = s.base + m * s.row_bytes + column_bytes;
  o.a = *(const uint2*)q;
  o.b = *(const uint2*)(q + s.second);
}

// bias + shortcut + ReLU + gate + fp32 / pair stores of 4 consecutive columns n..n+3 of row m (operands already loaded)
// DUAL: a pair-layout shortcut (in `o`) AND a ReLU gate (hi halves in `gate2`) -- the input gradient of an identity
// bottleneck handed to the block below already gated and split.
template <bool DUAL = false>
__device__ __forceinline__ f32x4 epilogue_apply4(const SplitGemmArgs& p, long m, int n, long poff, f32x4 v, const f32x4& bias4,
                                                 const EpilogueOperands& o, uint2 gate2 = make_uint2(0u, 0u)) {
  if (p.bias) v += bias4;
  if (p.res) {
    v.x += __uint_as_float(o.a.x); v.y += __uint_as_float(o.a.y); v.z += __uint_as_float(o.b.x); v.w += __uint_as_float(o.b.y);
  } else if (p.resp) {
    const unsigned hx = o.a.x, hy = o.a.y, lx = o.b.x, ly = o.b.y;
    v.x += __uint_as_float(hx << 16) + __uint_as_float(lx << 16);
    v.y += __uint_as_float(hx & 0xffff0000u) + __uint_as_float(lx & 0xffff0000u);
    v.z += __uint_as_float(hy << 16) + __uint_as_float(ly << 16);
    v.w += __uint_as_float(hy & 0xffff0000u) + __uint_as_float(ly & 0xffff0000u);
  }
  if (p.relu) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  if (p.gate) {  // data gradient of a layer whose input went through a ReLU: zero where that activation was <= 0
    const unsigned gx = DUAL ? gate2.x : o.a.x, gy = DUAL ? gate2.y : o.a.y;
    const unsigned a0 = gx & 0xffffu, a1 = gx >> 16, a2 = gy & 0xffffu, a3 = gy >> 16;
    if (a0 == 0u || a0 >= 0x8000u) v.x = 0.f;
    if (a1 == 0u || a1 >= 0x8000u) v.y = 0.f;
    if (a2 == 0u || a2 >= 0x8000u) v.z = 0.f;
    if (a3 == 0u || a3 >= 0x8000u) v.w = 0.f;
  }
  if (p.C) *(f32x4*)(p.C + m * p.ldc + n) = v;
  if (p.Cp) {
    const unsigned h01 = pack_bf16(v.x, v.y), h23 = pack_bf16(v.z, v.w);
    const unsigned l01 = pack_bf16(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
    const unsigned l23 = pack_bf16(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
    char* d = p.Cp + m * p.cp_rs + poff;
    *(uint2*)d = make_uint2(h01, h23);
    *(uint2*)(d + 64) = make_uint2(l01, l23);
  }
  return v;
}

__device__ __forceinline__ void epilogue_store4(const SplitGemmArgs& p, long m, int n, f32x4 v) {
  const long poff = (long)(n >> 5) * 128 + (n & 31) * 2;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias4 = *(const f32x4*)(p.bias + n);
  const EpilogueSource src = epilogue_source(p);
  EpilogueOperands o;
  if (src.base) epilogue_fetch4(src, m, epilogue_column_bytes(p, n, poff), o);
  epilogue_apply4(p, m, n, poff, v, bias4, o);
}

// POOL (plain products, no split-K): the epilogue also adds pool_scale x the final values of every row group of
// p.pool_rows rows (the 7 x 7 map of one RoI) into p.pool[m / pool_rows][n] -- the head's average pooling
// (box_head/roi_box_predictors.py:62-66) without a second pass over the [R*49, 2048] result; with C and Cp both null the
// result itself is never written (the no-grad teacher pass reads nothing but the pooled rows).  A lane sums its 16 rows
// per map segment of its wave's 64-row span (at most 3 maps for pool_rows >= 32), the four lanes that share the columns
// are combined by two xor-shuffles, one lane of them issues <= 12 fp32 atomics.  A map meets at most two 64-row spans when
// pool_rows <= 64, so every pooled value is the sum of at most two addends onto zero: bit-reproducible.
template <int WM, int WN, int MODE, int NS, int OCC, bool DUAL = false, bool POOL = false>
__global__ __launch_bounds__(WM * WN * 64, OCC) void split_gemm_kernel(SplitGemmArgs p, int tiles_n, int ntiles) {
  constexpr int BM = WM * 64, BN = WN * 64, NW = WM * WN;
  constexpr int A_ROWS = MODE == HALO ? BM + 2 * kHalo + 2 : BM;  // HALO: + 256 bytes of zeros (one bank period)
  constexpr int ZROW = BM + 2 * kHalo;
  constexpr int A_BYTES = A_ROWS * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PIECES = (MODE == HALO ? BM + 2 * kHalo : BM) / 8;  // 8-row LDS-DMA pieces of the A tile
  constexpr int AI = (A_PIECES + NW - 1) / NW;                        // per wave
  constexpr int BI = BN / 8 / NW;
  static_assert(MODE != HALO || NS == 1, "the halo tile is single-staged");
  static_assert(NW * 4096 <= STAGE * NS, "epilogue staging fits the k-loop's LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware renumbering (bijective for any grid): workgroup b runs on XCD b % 8
  int id;
  {
    const int nblocks = gridDim.x;
    const int b = blockIdx.x, q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, loc = b >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int slice = id / ntiles;     // split-K slice (0 when kslices == 1)
  const int tile = id - slice * ntiles;
  // tile order: column tiles are taken in groups of `p.gw`; within a group row tile-major.  The workgroups resident
  // on one XCD then share gw weight tiles (gw x BN rows x 4K bytes: L2-resident) while the A rows stream through.
  const int gw = p.gw, tiles_m = ntiles / tiles_n;
  const int per_group = tiles_m * gw;
  const int grp = tile / per_group;
  const int in_grp = tile - grp * per_group;
  const int tile_m = in_grp / gw, tile_n = grp * gw + (in_grp - tile_m * gw);
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;
  const int Mi = (int)p.M;           // M < 2^31 (checked by the launcher)

  const int cpb = p.ch >> 5;         // 32-value blocks per tap (PLAIN: of the first operand)
  const int nk = p.T * cpb + (p.ch2 >> 5);
  int dy = 0, dx = 0;
  auto set_tap = [&](int t) {
    dy = t / p.KW - p.KH / 2;
    dx = t % p.KW - p.KW / 2;
    if (p.flip) { dy = -dy; dx = -dx; }
  };

  // ---- per-lane load geometry (no memory reads: hipcc would wait vmcnt(0) on them inside the DMA pipeline).
  // A lane's DMA state is (lr = row within an 8-row piece, s0 / s1 = its swizzled 16-byte chunk offset in even / odd
  // pieces); the 64-bit source addresses are formed at issue time.  The one-stage loops re-derive even that from an
  // opaque copy of the lane id in every iteration (a dozen VALU operations against 48 MFMAs): left to itself the
  // compiler keeps ~30 loop-invariant address registers alive, which costs the fourth workgroup per CU (128 VGPRs).
  // chunk swizzle of LDS row `row` = piece * 8 + lr: row & 7 = lr (the same in even and odd pieces: s0 == s1)
  struct LaneGeom { int lr, s0, s1; };
  auto lane_geom = [&](int ln) {
    const int lr = ln >> 3, lc = ln & 7;
    return LaneGeom{lr, (lc ^ lr) * 16, (lc ^ lr) * 16};
  };
  auto clamp_row = [&](int gm) { return gm < 0 ? 0 : (gm > Mi - 1 ? Mi - 1 : gm); };  // rows outside the matrix:
                                                                                       // valid memory, never used unmasked
  int a_yx[MODE == SHIFTED ? AI : 1];  // SHIFTED: map position of the lane's row in each of its A pieces
  if (MODE == SHIFTED) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int gm = clamp_row((int)m0 + (wave * AI + i) * 8 + (lane >> 3));
      a_yx[i] = (((gm / p.W) % p.H) << 16) | (gm % p.W);
    }
  }

  // ---- LDS-DMA issue of the tiles of one k-step ----
  auto issue_b = [&](int stage, int kb, const LaneGeom& g) {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int piece = wave * BI + i;
      int gn = n0 + piece * 8 + g.lr;
      if (gn > p.N - 1) gn = p.N - 1;
      glds16(p.B + (long)gn * p.b_rs + (long)kb * 128 + ((piece & 1) ? g.s1 : g.s0),
             smem + stage * STAGE + A_BYTES + piece * 1024);
    }
  };
  auto issue_a_plain = [&](int stage, int kb, const LaneGeom& g) {
    const bool first = kb < cpb;     // wave-uniform: which of the two K-concatenated operands this k-step reads
    const char* base = first ? p.A : p.A2;
    const long rs = first ? p.a_rs : p.a2_rs;
    const long koff = (long)(first ? kb : kb - cpb) * 128;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int piece = wave * AI + i;
      glds16(base + (long)clamp_row((int)m0 + piece * 8 + g.lr) * rs + koff + ((piece & 1) ? g.s1 : g.s0),
             smem + stage * STAGE + piece * 1024);
    }
  };
  const char* zero_line = g_zero_line;
  auto issue_a_shifted = [&](int stage, int cb, const LaneGeom& g) {  // tap state (dy, dx) set by the caller
    const long shift = ((long)dy * p.W + dx) * p.a_rs + (long)cb * 128;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int piece = wave * AI + i;
      const char* src = p.A + (long)clamp_row((int)m0 + piece * 8 + g.lr) * p.a_rs + shift + ((piece & 1) ? g.s1 : g.s0);
      const int y = (a_yx[MODE == SHIFTED ? i : 0] >> 16) + dy, x = (a_yx[MODE == SHIFTED ? i : 0] & 0xffff) + dx;
      if ((unsigned)y >= (unsigned)p.H || (unsigned)x >= (unsigned)p.W) src = zero_line + (g.s0 & 0x70);
      glds16(src, smem + stage * STAGE + piece * 1024);
    }
  };
  auto issue_a_halo = [&](int cb, const LaneGeom& g) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int piece = wave + NW * i;
      if (piece < A_PIECES)  // wave-uniform
        glds16(p.A + (long)clamp_row((int)m0 + piece * 8 + g.lr - kHalo) * p.a_rs + (long)cb * 128 + ((piece & 1) ? g.s1 : g.s0),
               smem + piece * 1024);
    }
  };

  // ---- DEFORM: the A tile of a k-step = (tap, 32-channel block) is SAMPLED, not loaded: the deformable-convolution
  // sample of modulated / plain DCN (deform_conv_kernel_cuda.cu:198-250, 578-644: bilinear with zero padding, the four
  // terms in the reference's order, times the mask) is split into bf16 hi / lo and written where the LDS-DMA of the
  // other modes would have put it.  The cells of step kb + 1 are fetched into registers under the MFMAs of step kb.
  // Thread t of the 256 samples rows (t >> 3) + 32 j, j < 4, and the four channels 4 (t & 7) .. + 3 of the 32-channel
  // block: eight neighbouring lanes read one whole 128-byte line of a cell (a lane owning 16 channels of one row had
  // every load instruction touch 32 different lines for 32 bytes each, and the kernel sat on the texture-address rate).
  // Per-(tap, deformable group) state of the thread's four rows, recomputed only when the tap or the group changes (a
  // tap spans ch / 32 k-steps): bilinear weights, element offsets of the four cells (-1: outside the image) and mask.
  constexpr int DR = MODE == DEFORM ? 4 : 1;
  float d_w[DR][4], d_mask[DR];
  int d_o[DR][4];
  int d_tap = -1, d_grp = -1;
  f32x4 d_v[DR][4];  // [row][cell]: the 4 channels of the k-step fetched ahead
  auto deform_fetch = [&](int tap, int cb) {
    if constexpr (MODE == DEFORM) {
      const int prow0 = threadIdx.x >> 3, pch = (threadIdx.x & 7) * 4;
      const int grp = (cb * 32) / (p.ch / p.dg);
      if (tap != d_tap || grp != d_grp) {  // wave-uniform
        d_tap = tap;
        d_grp = grp;
        const long plane_o = (long)p.Ho * p.Wo;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gm = clamp_row((int)m0 + prow0 + 32 * j);
          const int wo = gm % p.Wo, t1 = gm / p.Wo;
          const int ho = t1 % p.Ho, bimg = t1 / p.Ho;
          const long pix = (long)ho * p.Wo + wo;
          const float* off = p.doff + (((long)bimg * p.dg + grp) * 2 * p.T + 2 * tap) * plane_o + pix;
          const float h = (float)(ho * p.sh - p.ph + ky * p.dlh) + off[0];
          const float w = (float)(wo * p.sw - p.pw + kx * p.dlw) + off[plane_o];
          d_o[j][0] = d_o[j][1] = d_o[j][2] = d_o[j][3] = -1;
          d_w[j][0] = d_w[j][1] = d_w[j][2] = d_w[j][3] = 0.f;
          if (h > -1.f && w > -1.f && h < (float)p.H && w < (float)p.W) {
            const int hl = (int)floorf(h), wl = (int)floorf(w);
            const int hh = hl + 1, wh = wl + 1;
            const float lh = h - hl, lw = w - wl, uh = 1.f - lh, uw = 1.f - lw;
            d_w[j][0] = uh * uw; d_w[j][1] = uh * lw; d_w[j][2] = lh * uw; d_w[j][3] = lh * lw;
            const int img = bimg * p.H * p.W;  // the image has fewer than 2^31 elements (checked by the launcher)
            if (hl >= 0 && wl >= 0) d_o[j][0] = (img + hl * p.W + wl) * p.ch;
            if (hl >= 0 && wh <= p.W - 1) d_o[j][1] = (img + hl * p.W + wh) * p.ch;
            if (hh <= p.H - 1 && wl >= 0) d_o[j][2] = (img + hh * p.W + wl) * p.ch;
            if (hh <= p.H - 1 && wh <= p.W - 1) d_o[j][3] = (img + hh * p.W + wh) * p.ch;
          }
          d_mask[j] = p.dmask ? p.dmask[(((long)bimg * p.dg + grp) * p.T + tap) * plane_o + pix] : 1.f;
        }
      }
      const float* base = p.dimg + cb * 32 + pch;
      const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) d_v[j][c] = d_o[j][c] >= 0 ? *(const f32x4*)(base + d_o[j][c]) : z;
    }
  };
  // bilinear combination (the reference's four-term order), mask, bf16 hi / lo split and the LDS writes of the fetched step
  auto deform_commit = [&]() {
    if constexpr (MODE == DEFORM) {
      const int prow0 = threadIdx.x >> 3, chunk = (threadIdx.x & 7) >> 1, sub = (threadIdx.x & 1) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = d_w[j][0] * d_v[j][0] + d_w[j][1] * d_v[j][1] + d_w[j][2] * d_v[j][2] + d_w[j][3] * d_v[j][3];
        if (p.dmask) v *= d_mask[j];
        const unsigned h0 = pack_bf16(v.x, v.y), h1 = pack_bf16(v.z, v.w);
        const unsigned l0 = pack_bf16(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
        const unsigned l1 = pack_bf16(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
        const int prow = prow0 + 32 * j, sw = prow & 7;
        *(uint2*)(smem + prow * 128 + ((chunk ^ sw) << 4) + sub) = make_uint2(h0, h1);
        *(uint2*)(smem + prow * 128 + (((4 + chunk) ^ sw) << 4) + sub) = make_uint2(l0, l1);
      }
    }
  };

  // ---- fragment read addresses.  Fragment f of a wave covers tile rows .. + f*16 + frow: the swizzle term
  // (row & 7) does not depend on f, so the four fragments of an operand sit 2048 bytes apart. ----
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15, fc = lane >> 4;
  const int a_rd0 = (wm * 64 + frow) * 128 + ((fc ^ (frow & 7)) << 4);
  const int b_rd0 = A_BYTES + (wn * 64 + frow) * 128 + ((fc ^ (frow & 7)) << 4);
  // HALO: which taps of the lane's row (in each of its 4 fragments) fall inside the map: bit t of a 16-bit field
  unsigned h_ok[2] = {0u, 0u};
  if (MODE == HALO) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int gm = clamp_row((int)m0 + wm * 64 + f * 16 + frow);
      const int x = gm % p.W, y = (gm / p.W) % p.H;
      unsigned mask = 0u;
      for (int t = 0; t < p.T; ++t) {
        set_tap(t);
        if ((unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W) mask |= 1u << t;
      }
      h_ok[f >> 1] |= mask << ((f & 1) * 16);
    }
    if (threadIdx.x < 16) *(uint4*)(smem + ZROW * 128 + threadIdx.x * 16) = make_uint4(0u, 0u, 0u, 0u);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one k-step on the staged tiles: 16 fragment reads, 48 MFMAs (hi.hi, lo.hi, hi.lo per fragment pair); a0..a3 =
  // LDS addresses of the lane's four A fragments
  auto compute = [&](const char* base, int b0, int a0, int a1, int a2, int a3) {
    const int ard[4] = {a0, a1, a2, a3};
    bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bh[f] = *(const bf16x8*)(base + b0 + f * 2048);
      ah[f] = *(const bf16x8*)(base + ard[f]);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bl[f] = *(const bf16x8*)(base + ((b0 + f * 2048) ^ 64));
      al[f] = *(const bf16x8*)(base + (ard[f] ^ 64));
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[g], ah[f], acc[f][g], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[g], al[f], acc[f][g], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[g], ah[f], acc[f][g], 0, 0, 0);
  };
  auto compute_static = [&](const char* base) { compute(base, b_rd0, a_rd0, a_rd0 + 2048, a_rd0 + 4096, a_rd0 + 6144); };

  if (MODE == HALO) {
    // k = (channel block, tap): one halo tile per channel block, one B tile per tap
    const int cb0 = slice * p.steps_per_slice;
    int cb1 = cb0 + p.steps_per_slice;
    if (cb1 > cpb) cb1 = cpb;
    int t = 0, cb = cb0;
    for (int it = (cb1 - cb0) * p.T; it > 0; --it) {  // one flat loop: per-tap state is re-derived, not kept per tap
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const LaneGeom g = lane_geom(ln);
      const int fr = ln & 15, fq = ln >> 4;
      if (t == 0) issue_a_halo(cb, g);
      issue_b(0, t * cpb + cb, g);
      set_tap(t);
      // the tap's rows sit dy*W + dx LDS rows further (always inside the halo tile); lanes whose shifted pixel lies
      // outside the map read the zero row instead
      const int L0 = wm * 64 + fr + kHalo + dy * p.W + dx;
      // the zero block spans one whole bank period (256 B) and a lane reads it at its regular address modulo 256: the
      // same banks as the row it replaces, so the mix of zero and regular lanes of a border tap stays conflict-free
      // (one 128-byte zero row read at chunk fq cost the 3x3 kernel 31 % LDS bank-conflict cycles on the 7x7 maps)
      const int ard0 = L0 * 128 + ((fq ^ (L0 & 7)) << 4), zrd = ZROW * 128 + (ard0 & 255);
      const int b0 = A_BYTES + (wn * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4);
      int ard[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) ard[f] = ((h_ok[f >> 1] >> ((f & 1) * 16 + t)) & 1u) ? ard0 + f * 2048 : zrd;
      __syncthreads();
      compute(smem, b0, ard[0], ard[1], ard[2], ard[3]);
      __syncthreads();
      if (++t == p.T) {
        t = 0;
        ++cb;
      }
    }
  } else {
    const int kb0 = slice * p.steps_per_slice;
    int kb1 = kb0 + p.steps_per_slice;
    if (kb1 > nk) kb1 = nk;
    int ld_tap = 0, ld_cb = kb0;       // (tap, channel block) of the next stage to load (SHIFTED / DEFORM)
    if (MODE == SHIFTED || MODE == DEFORM) {
      ld_tap = kb0 / cpb;
      ld_cb = kb0 - ld_tap * cpb;
      set_tap(ld_tap);
    }
    auto issue = [&](int stage, int kb, const LaneGeom& g) {
      if (MODE == DEFORM) {
        // the A tile is written by deform_commit (its operands were fetched a k-step ahead): only B is loaded here
      } else if (MODE == SHIFTED) {
        issue_a_shifted(stage, ld_cb, g);
        if (++ld_cb == cpb) {
          ld_cb = 0;
          set_tap(++ld_tap);
        }
      } else {
        issue_a_plain(stage, kb, g);
      }
      issue_b(stage, kb, g);
    };
    if (MODE == DEFORM) {
      // sampled A tile: the cell loads of step kb + 1 are in flight under the MFMAs of step kb
      auto advance = [&]() {
        if (++ld_cb == cpb) {
          ld_cb = 0;
          ++ld_tap;
        }
      };
      if (kb0 < kb1) deform_fetch(ld_tap, ld_cb);
      for (int kb = kb0; kb < kb1; ++kb) {
        issue(0, kb, lane_geom(lane));  // the weight tile's DMA first: it lands under the sampling arithmetic
        deform_commit();
        __syncthreads();
        advance();
        if (kb + 1 < kb1) deform_fetch(ld_tap, ld_cb);
        compute_static(smem);
        __syncthreads();
      }
    } else if (NS == 1) {
      for (int kb = kb0; kb < kb1; ++kb) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        issue(0, kb, lane_geom(ln));
        __syncthreads();
        compute_static(smem);
        __syncthreads();
      }
    } else {
      // one __syncthreads per k-step (its fence drains this wave's DMAs: stage kb landed, stage kb-1's buffer free),
      // the next stage in flight under the MFMAs
      const LaneGeom g = lane_geom(lane);
      if (kb0 < kb1) issue(0, kb0, g);
      for (int kb = kb0; kb < kb1; ++kb) {
        __syncthreads();
        if (kb + 1 < kb1) issue((kb + 1 - kb0) & 1, kb + 1, g);
        compute_static(smem + ((kb - kb0) & 1) * STAGE);
      }
      __syncthreads();  // every wave is done with the stages: the epilogue reuses them
    }
  }

  // ---- epilogue through the wave-private staging tile: 16 rows x 64 columns fp32, 16-byte slot s of row r at
  // slot s ^ r (conflict-free for the fragment-order writes and the row-order reads) ----
  char* stg = smem + wave * 4096;
  const int er = lane >> 4, es = lane & 15;   // row within a group of 4, 4-column slot
  const int n = n0 + wn * 64 + es * 4;
  float* slab = p.slab ? p.slab + (long)slice * p.M * p.N : nullptr;
  // The shortcut / gate operands of row group f + 1 are requested before group f is stored and those of group 0 before
  // anything else: a lane's 16 (row, 4 columns) pieces would otherwise be 16 load -> wait -> store round trips in a row
  // (vmcnt also counts the stores), ~30 us per tile against ~10 us of matrix work for a K = 512 product.
  const bool col_ok = n < p.N;
  const long poff = (long)(n >> 5) * 128 + (n & 31) * 2;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (!slab && p.bias && col_ok) bias4 = *(const f32x4*)(p.bias + n);
  // pieces of 2 rows, double-buffered: 8 pieces per lane
  // NBUF - 1 pieces are in flight ahead of the one being stored; with two ahead the wait for a piece's operands no longer
  // includes the previous piece's stores (vmcnt retires in order).  The single-stage SHIFTED form has no registers for it.
  constexpr int NBUF = DUAL ? 2 : ((MODE == SHIFTED && NS == 1) ? 2 : 3);  // (deeper: no gain, tools/experiments/patches)
  EpilogueOperands ops[NBUF][2];
  uint2 gates[NBUF][2];  // DUAL only
  // The fetches are UNCONDITIONAL straight-line loads (rows / columns outside the problem re-read its last row / first
  // column; a launch without an operand, or a split-K slice, reads the first bytes of B with a zero row stride and
  // ignores them): a load under a branch makes the wait-count pass assume "nothing was issued since", i.e. vmcnt(0)
  // in front of every use, which also drains the prefetch for the next piece.
  EpilogueSource src = epilogue_source(p);
  long src_col = epilogue_column_bytes(p, col_ok ? n : 0, col_ok ? poff : 0);
  if (slab || !src.base) {
    src.base = p.B;
    src.row_bytes = 0;
    src.second = 8;
    src_col = 0;
  }
  const long m_last = p.M - 1;
  const char* gate_col = DUAL ? p.gate + (col_ok ? poff : 0) : nullptr;
  auto fetch = [&](int piece, EpilogueOperands (&o)[2], uint2 (&g)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long m = m0 + wm * 64 + (piece >> 1) * 16 + ((piece & 1) * 2 + j) * 4 + er;
      const long mc = m < m_last ? m : m_last;
      epilogue_fetch4(src, mc, src_col, o[j]);
      if constexpr (DUAL) g[j] = *(const uint2*)(gate_col + mc * p.gate_rs);
    }
  };
  // POOL: map segment boundaries inside this wave's 64 rows (rows [0, b1) belong to the first map, [b1, b2) to the second)
  f32x4 ps0 = {0.f, 0.f, 0.f, 0.f}, ps1 = ps0, ps2 = ps0;
  int pool_b1 = 0, pool_b2 = 0;
  long pool_map0 = 0;
  if constexpr (POOL) {
    const long mw = m0 + wm * 64;
    pool_map0 = mw / p.pool_rows;
    pool_b1 = (int)((pool_map0 + 1) * p.pool_rows - mw);
    pool_b2 = pool_b1 + p.pool_rows;
  }
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) fetch(i, ops[i], gates[i]);
#pragma unroll
  for (int f = 0; f < 4; ++f) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *(f32x4*)(stg + frow * 256 + (((g * 4 + fc) ^ frow) << 4)) = acc[f][g];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int piece = f * 2 + h;
      if (piece + NBUF - 1 < 8) fetch(piece + NBUF - 1, ops[(piece + NBUF - 1) % NBUF], gates[(piece + NBUF - 1) % NBUF]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = (h * 2 + j) * 4 + er;
        const f32x4 v = *(const f32x4*)(stg + r * 256 + ((es ^ r) << 4));
        const long m = m0 + wm * 64 + f * 16 + r;
        if (m < p.M && col_ok) {
          if (slab) *(f32x4*)(slab + m * p.N + n) = v;
          else {
            const f32x4 fin = epilogue_apply4<DUAL>(p, m, n, poff, v, bias4, ops[piece % NBUF][j], gates[piece % NBUF][j]);
            if constexpr (POOL) {
              const int rr = f * 16 + r;   // row inside the wave's 64-row span
              const f32x4 z = {0.f, 0.f, 0.f, 0.f};
              ps0 += rr < pool_b1 ? fin : z;
              ps1 += (rr >= pool_b1 && rr < pool_b2) ? fin : z;
              ps2 += rr >= pool_b2 ? fin : z;
            }
          }
        }
      }
    }
  }
  if constexpr (POOL) {
    f32x4* ps[3] = {&ps0, &ps1, &ps2};
#pragma unroll
    for (int sgm = 0; sgm < 3; ++sgm) {
      f32x4 v = *ps[sgm];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += __shfl_xor(v[e], 16, 64);
        v[e] += __shfl_xor(v[e], 32, 64);
      }
      const long first = (m0 + wm * 64) + (sgm == 0 ? 0 : (sgm == 1 ? pool_b1 : pool_b2));  // first row of the segment
      if (er == 0 && col_ok && first < p.M && (sgm == 0 || (sgm == 1 ? pool_b1 : pool_b2) < 64)) {
        float* d = p.pool + (pool_map0 + sgm) * p.N + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(d + e, v[e] * p.pool_scale);
      }
    }
  }
}

// Sum of the split-K slabs + the epilogue of split_gemm_kernel, 4 columns per thread.
__global__ __launch_bounds__(256) void split_gemm_finish_kernel(SplitGemmArgs p) {
  const int n4 = p.N >> 2;
  const long total = p.M * n4;
  const long slab = p.M * p.N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / n4;
    const int n = (int)(i - m * n4) * 4;
    const float* s = p.slab + m * p.N + n;
    f32x4 v = *(const f32x4*)s;
    for (int k = 1; k < p.kslices; ++k) v += *(const f32x4*)(s + k * slab);
    epilogue_store4(p, m, n, v);
  }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient: dW[n, (tap, c)] = sum_m G[m, n] * X[row(m, tap), c] -- a contraction over the ROWS of two
// row-major pair operands, so both MFMA operands are needed k(=m)-contiguous per lane while memory is n- / c-
// contiguous.  The tiles are staged as they lie ([32 rows][512 B], LDS-DMA) and read with the LDS transpose
// read ds_read_b64_tr_b16: a 16-lane group fetches a 4(m) x 16(n) block (lane p: row p/4, 8 bytes at (p%4)*8) and
// every lane receives the 4 m-values of its own column; two of them make one 16x16x32 operand.  Same three-term
// hi/lo product, same wave tiling as the forward kernel.  The 32-byte segment index of a row is XOR-ed with
// f(row) = ((row >> 3) & 1) << 2 | (row & 3) (on the DMA source address) so that the 8 rows a 32-lane half reads sit
// on 8 different bank octets.  The contraction is long (M = R*49 rows) and the output small, so the rows are cut
// into `slices` and every (tile, slice) workgroup writes its own fp32 slab; the caller sums the slabs
// (deterministic, no atomics).  For a 3x3 the X rows are read shifted by the tap (zero line outside the map, from a
// per-position tap mask table in LDS): no im2col rows are materialised for the weight gradient either.
// ---------------------------------------------------------------------------------------------------
typedef short bf16x4 __attribute__((ext_vector_type(4)));

struct SplitGemmTnArgs {
  const char* G; long g_rs;   // [M] pair rows of N values
  const char* X; long x_rs;   // [M] pair rows of ch values
  float* C;                   // [slices][N][T*ch]
  long M; int N; int ch; int T; int H; int W; int KH; int KW; int slices; int steps_per_slice;
  int g_major;                // tile order inside a row slice: G column tile major (else X column tile major), see the launcher
};

// The transpose reads are inline asm: hipcc treats the ds_read_tr builtin as possibly aliasing the in-flight LDS-DMA
// stage and would wait vmcnt(0) before the first one of every step, i.e. serialise the next stage's loads with this
// stage's MFMAs.  Inside asm it counts nothing, so the reads are followed by explicit lgkmcnt waits that name their
// destination registers ("+v") -- every consumer is ordered after its wait.
#define OVIS_TR_READ(dst, addr, off) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define OVIS_LGKM_WAIT8(N, a, b, c, d, e, f, g, h)                                                                \
  asm volatile("s_waitcnt lgkmcnt(%8)"                                                                             \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)                            \
               : "n"(N)                                                                                            \
               : "memory")

// SMALL (with CONV): maps of at most 64 pixels (the 7x7 maps of the res5 head).  Which rows of a map a tap reads inside the
// map is then one 64-bit mask (built by a ballot at kernel start) and a DMA row only tracks its pixel index -- ~10 vector
// instructions per row and step instead of ~45 for the (y, x) bookkeeping of the general form, which at 181 VALU + 73
// SALU per 48 MFMAs had the 3x3 weight gradient issue-bound (1000 TFLOP/s against 1260-1360 for the 1x1 forms).
template <bool CONV, bool SMALL = false>
__global__ __launch_bounds__(256, 2) void split_gemm_tn_kernel(SplitGemmTnArgs p, int tiles_i, int tiles_j,
                                                              int nblocks) {
  constexpr int TILE_BYTES = 32 * 512, STAGE = 2 * TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware renumbering INSIDE chunks of one resident round (2 workgroups x 256 CUs): every XCD gets a contiguous
  // run of a chunk's ids (neighbouring tiles share G / X blocks in its L2), while the chunks follow each other in
  // dispatch order -- so the workgroups resident at one time cover ~half of the row slices (ids are slice-major) and
  // their G / X working set (~200 MB on the res5 shapes) stays inside the 256 MB Infinity Cache.  Renumbering the whole
  // grid at once spread the resident set over ALL slices (410 MB) and the kernel fetched 2.3x its algorithmic bytes
  // from HBM (profiles/r1_pmc_step_hbm_traffic_student.json).
  int id;
  {
    constexpr int kChunk = 2 * OVIS_NUM_CU;
    const int b = blockIdx.x, base = b / kChunk * kChunk;
    const int nb = min(nblocks - base, kChunk), bl = b - base;
    const int q = nb >> 3, r = nb & 7, xcd = bl & 7, loc = bl >> 3;
    id = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int per_slice = tiles_i * tiles_j;
  const int slice = id / per_slice;
  const int rem = id - slice * per_slice;
  const int tile_j = p.g_major ? rem % tiles_j : rem / tiles_i;
  const int tile_i = p.g_major ? rem / tiles_j : rem - tile_j * tiles_i;
  const int i0 = tile_i * 128, j0 = tile_j * 128;
  const int tap = CONV ? j0 / p.ch : 0, c0 = CONV ? j0 - tap * p.ch : j0;
  // tap of this workgroup's column tile: X rows are read shifted by (tdy, tdx); a row whose shifted pixel lies outside
  // the map reads the zero line.  The (y, x) position of every DMA row is tracked incrementally (rows advance by 32
  // per step: x += 32 % W with carry into y), so maps of any size work and no table is needed.
  int off_rows = 0, tdy = 0, tdx = 0;
  if (CONV) {
    tdy = tap / p.KW - p.KH / 2;
    tdx = tap % p.KW - p.KW / 2;
    off_rows = tdy * p.W + tdx;
  }
  const long step0 = (long)slice * p.steps_per_slice;
  const long steps_total = (p.M + 31) >> 5;
  long step1 = step0 + p.steps_per_slice;
  if (step1 > steps_total) step1 = steps_total;
  const int nk = step1 > step0 ? (int)(step1 - step0) : 0;

  // ---- per-lane DMA geometry: 4 G pieces + 4 X pieces per wave per step, a piece = 2 rows x 512 B ----
  const int prow = lane >> 5, pseg = (lane & 31) >> 1, phalf = lane & 1;
  int rows[4];          // tile row of the lane in its 4 pieces
  long g_off[4], x_off[4];
  int x_y[4], x_x[4];   // map position of the X row, kept incrementally (SMALL: x_x = pixel index y * W + x)
  int m_row[4];
  const int HW = CONV ? p.H * p.W : 1;
  unsigned long long tap_mask = 0;  // SMALL: bit q set <=> pixel q of a map, shifted by this tap, is inside the map
  if (CONV && SMALL) {
    const int qy = lane / p.W, qx = lane - qy * p.W;
    tap_mask = __ballot(lane < HW && (unsigned)(qy + tdy) < (unsigned)p.H && (unsigned)(qx + tdx) < (unsigned)p.W);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave * 4 + i;
    const int row = piece * 2 + prow;
    rows[i] = row;
    const int fr = (((row >> 3) & 1) << 2) | (row & 3);
    const int seg = pseg ^ fr;
    const int m = (int)(step0 << 5) + row;
    m_row[i] = m;
    g_off[i] = (long)m * p.g_rs + (long)i0 * 4 + seg * 32 + phalf * 16;
    x_off[i] = ((long)m + off_rows) * p.x_rs + (long)c0 * 4 + seg * 32 + phalf * 16;
    x_x[i] = CONV ? (SMALL ? m % HW : m % p.W) : 0;
    x_y[i] = (CONV && !SMALL) ? (m / p.W) % p.H : 0;
  }
  const int step_q = (CONV && SMALL) ? 32 % HW : 0;
  const int step_dx = CONV ? 32 % p.W : 0, step_dy = CONV ? 32 / p.W : 0;
  const bool y_single = step_dy + 1 <= p.H;  // one conditional subtraction brings y back into [0, H)
  const char* zero_src = g_zero_line + (lane & 7) * 16;
  const int Mi = (int)p.M;

  auto issue = [&](int stage) {
    char* base = smem + stage * STAGE;
    bool x_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x_ok[i] = m_row[i] < Mi;
      if (CONV && SMALL) {
        x_ok[i] = x_ok[i] && ((tap_mask >> x_x[i]) & 1ull) != 0ull;
        int q = x_x[i] + step_q;
        if (q >= HW) q -= HW;
        x_x[i] = q;
      } else if (CONV) {
        x_ok[i] = x_ok[i] && (unsigned)(x_y[i] + tdy) < (unsigned)p.H && (unsigned)(x_x[i] + tdx) < (unsigned)p.W;
        int x = x_x[i] + step_dx, y = x_y[i] + step_dy;
        if (x >= p.W) { x -= p.W; ++y; }
        if (y_single) { if (y >= p.H) y -= p.H; } else { y %= p.H; }
        x_x[i] = x;
        x_y[i] = y;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* src = p.G + g_off[i];
      if (m_row[i] >= Mi) src = zero_src;
      glds16(src, base + (wave * 4 + i) * 1024);
      g_off[i] += 32 * p.g_rs;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* src = p.X + x_off[i];
      if (!x_ok[i]) src = zero_src;
      glds16(src, base + TILE_BYTES + (wave * 4 + i) * 1024);
      x_off[i] += 32 * p.x_rs;
      m_row[i] += 32;
    }
  };

  // ---- transpose-read addresses: lane (group gq, p): row 8*gq + p/4 (+4 second half), 8 bytes at (p%4)*8 of a segment
  const int wi = wave >> 1, wj = wave & 1;
  const int gq = lane >> 4, pp = lane & 15;
  const int rrow = 8 * gq + (pp >> 2);
  const int rfr = ((gq & 1) << 2) | (pp >> 2);
  int g_rd[4], x_rd[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int seg_g = (2 * wi + (f >> 1)) * 4 + (f & 1);
    g_rd[f] = rrow * 512 + ((seg_g ^ rfr) << 5) + (pp & 3) * 8;
    const int seg_x = (2 * wj + (f >> 1)) * 4 + (f & 1);
    x_rd[f] = TILE_BYTES + rrow * 512 + ((seg_x ^ rfr) << 5) + (pp & 3) * 8;
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // two LDS stages: the next stage loads under this stage's MFMAs, two workgroups per CU
  if (nk > 0) issue(0);
  for (int kb = 0; kb < nk; ++kb) {
    __syncthreads();
    if (kb + 1 < nk) issue((kb + 1) & 1);
    const int so = (kb & 1) * STAGE;  // the dynamic LDS segment starts at LDS address 0 (no static LDS in this TU's kernels)
    bf16x4 xa[4][2], xb[4][2], ga[4][2], gb[4][2];  // [fragment][hi/lo]: first / second 4 rows of the lane's 8 k values
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int ax = so + x_rd[f], ag = so + g_rd[f];
      OVIS_TR_READ(xa[f][0], ax, 0);
      OVIS_TR_READ(xb[f][0], ax, 2048);
      OVIS_TR_READ(ga[f][0], ag, 0);
      OVIS_TR_READ(gb[f][0], ag, 2048);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int ax = so + (x_rd[f] ^ 64), ag = so + (g_rd[f] ^ 64);  // lo half = segment + 2 (XOR commutes with the swizzle)
      OVIS_TR_READ(xa[f][1], ax, 0);
      OVIS_TR_READ(xb[f][1], ax, 2048);
      OVIS_TR_READ(ga[f][1], ag, 0);
      OVIS_TR_READ(gb[f][1], ag, 2048);
    }
    // 32 reads are outstanding; LDS returns in order: lgkmcnt(N) = all but the youngest N have landed
    OVIS_LGKM_WAIT8(15, xa[0][0], xb[0][0], ga[0][0], gb[0][0], xa[1][0], xb[1][0], ga[1][0], gb[1][0]);
    OVIS_LGKM_WAIT8(15, xa[2][0], xb[2][0], ga[2][0], gb[2][0], xa[3][0], xb[3][0], ga[3][0], gb[3][0]);
    bf16x8 gh[4], gl[4], xh[4], xl[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      xh[f] = __builtin_shufflevector(xa[f][0], xb[f][0], 0, 1, 2, 3, 4, 5, 6, 7);
      gh[f] = __builtin_shufflevector(ga[f][0], gb[f][0], 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[g], gh[f], acc[f][g], 0, 0, 0);
    OVIS_LGKM_WAIT8(0, xa[0][1], xb[0][1], ga[0][1], gb[0][1], xa[1][1], xb[1][1], ga[1][1], gb[1][1]);
    OVIS_LGKM_WAIT8(0, xa[2][1], xb[2][1], ga[2][1], gb[2][1], xa[3][1], xb[3][1], ga[3][1], gb[3][1]);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      xl[f] = __builtin_shufflevector(xa[f][1], xb[f][1], 0, 1, 2, 3, 4, 5, 6, 7);
      gl[f] = __builtin_shufflevector(ga[f][1], gb[f][1], 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[g], gl[f], acc[f][g], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[g], gh[f], acc[f][g], 0, 0, 0);
  }

  // D[j][i]: lane owns column i = (lane & 15) of G-block f and rows j = 4*(lane >> 4) + {0..3} of X-block g
  const long ld = (long)p.T * p.ch;
  float* out = p.C + (long)slice * p.N * ld;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int i = i0 + wi * 64 + f * 16 + (lane & 15);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j = j0 + wj * 64 + g * 16 + (lane >> 4) * 4;
      *(f32x4*)(out + (long)i * ld + j) = acc[f][g];
    }
  }
}

// fp32 [rows, cols] (row stride src_rs floats) -> pair layout [rows, cols/32 x (hi32 | lo32)] bf16; a thread owns 8
// values: 32 B read, 16 B + 16 B written.  HBM-bound: 8 B per element.
__global__ __launch_bounds__(256) void split_pair_kernel(const float* __restrict__ src, long src_rs,
                                                        char* __restrict__ dst, long rows, int cols) {
  const int oc = cols >> 3;  // 8-value groups per row
  const long total = rows * oc;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / oc;
    const int c = (int)(i - r * oc) * 8;
    const float4 v0 = *(const float4*)(src + r * src_rs + c), v1 = *(const float4*)(src + r * src_rs + c + 4);
    const unsigned h0 = pack_bf16(v0.x, v0.y), h1 = pack_bf16(v0.z, v0.w), h2 = pack_bf16(v1.x, v1.y),
                   h3 = pack_bf16(v1.z, v1.w);
    const unsigned l0 = pack_bf16(v0.x - __uint_as_float(h0 << 16), v0.y - __uint_as_float(h0 & 0xffff0000u));
    const unsigned l1 = pack_bf16(v0.z - __uint_as_float(h1 << 16), v0.w - __uint_as_float(h1 & 0xffff0000u));
    const unsigned l2 = pack_bf16(v1.x - __uint_as_float(h2 << 16), v1.y - __uint_as_float(h2 & 0xffff0000u));
    const unsigned l3 = pack_bf16(v1.z - __uint_as_float(h3 << 16), v1.w - __uint_as_float(h3 & 0xffff0000u));
    char* d = dst + r * 4L * cols + (long)(c >> 5) * 128 + (c & 31) * 2;
    *(uint4*)d = make_uint4(h0, h1, h2, h3);
    *(uint4*)(d + 64) = make_uint4(l0, l1, l2, l3);
  }
}

// ReLU gate of a backward pass fused with the operand split of the gated gradient: g = dy * (y > 0), written in
// pair layout (the operand of the dX / dW GEMMs) and optionally as fp32 (the shortcut branch needs it).  The gate
// reads either the saved fp32 output y or only the hi half of its pair form (bf16 rounding keeps the sign and maps
// no normal positive value to zero).  8 values per thread; 4 (+2 or +4) B read, 4 (+4) B written per element.
template <bool GATE_PAIR, bool WRITE_F32>
__global__ __launch_bounds__(256) void gate_split_pair_kernel(const float* __restrict__ dy, long dy_rs,
                                                             const void* __restrict__ gate, char* __restrict__ dst,
                                                             float* __restrict__ g32, long rows, int cols,
                                                             const float* __restrict__ gpool, int pool_rows,
                                                             float pool_scale, const float* __restrict__ gsel,
                                                             const int* __restrict__ group_slot) {
  const int oc = cols >> 3;
  const long total = rows * oc;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / oc;
    const int c = (int)(i - r * oc) * 8;
    float v[8];
    if (dy) {
      *(float4*)v = *(const float4*)(dy + r * dy_rs + c);
      *(float4*)(v + 4) = *(const float4*)(dy + r * dy_rs + c + 4);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
    if (gpool) {  // + the gradient of the mean over every `pool_rows` consecutive rows (average pooling behind this layer)
      const float* gp = gpool + (r / pool_rows) * (long)cols + c;
      float q[8];
      *(float4*)q = *(const float4*)gp;
      *(float4*)(q + 4) = *(const float4*)(gp + 4);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = dy ? v[k] + q[k] * pool_scale : q[k] * pool_scale;
    }
    if (gsel) {  // + the gradient of a row gather: groups (of pool_rows rows) that were selected carry their own dense rows
      const long grp = r / pool_rows;
      const int slot = group_slot[grp];
      if (slot >= 0) {
        const float* gs = gsel + ((long)slot * pool_rows + (r - grp * pool_rows)) * cols + c;
        float q[8];
        *(float4*)q = *(const float4*)gs;
        *(float4*)(q + 4) = *(const float4*)(gs + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += q[k];
      }
    }
    const long poff = r * 4L * cols + (long)(c >> 5) * 128 + (c & 31) * 2;
    if (gate) {
      if (GATE_PAIR) {
        const uint4 h = *(const uint4*)((const char*)gate + poff);  // 8 bf16 hi values: positive iff 0 < bits < 0x8000
        const unsigned w[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned a = w[k] & 0xffffu, b = w[k] >> 16;
          if (a == 0u || a >= 0x8000u) v[2 * k] = 0.f;
          if (b == 0u || b >= 0x8000u) v[2 * k + 1] = 0.f;
        }
      } else {
        float y[8];
        *(float4*)y = *(const float4*)((const float*)gate + r * (long)cols + c);
        *(float4*)(y + 4) = *(const float4*)((const float*)gate + r * (long)cols + c + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (!(y[k] > 0.f)) v[k] = 0.f;
      }
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      h[k] = pack_bf16(v[2 * k], v[2 * k + 1]);
      l[k] = pack_bf16(v[2 * k] - __uint_as_float(h[k] << 16), v[2 * k + 1] - __uint_as_float(h[k] & 0xffff0000u));
    }
    *(uint4*)(dst + poff) = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(dst + poff + 64) = make_uint4(l[0], l[1], l[2], l[3]);
    if (WRITE_F32) {
      *(float4*)(g32 + r * (long)cols + c) = *(float4*)v;
      *(float4*)(g32 + r * (long)cols + c + 4) = *(float4*)(v + 4);
    }
  }
}

// im2col of a strided convolution on an NCHW fp32 image straight into pair rows (the 7x7 / stride-2 stem,
// resnet.py:347-366, as a split GEMM: 3 input channels give the matrix cores nothing to do in a direct kernel):
// row m = (n, oy, ox), k = (ky*KW + kx)*C + c, zero-padded to kp (% 32 == 0) columns.  A thread builds 8 k values of
// one row (8 bounds-checked gathers, L1/L2-served) and writes 16 B hi + 16 B lo.
__global__ __launch_bounds__(256) void im2col_nchw_pair_kernel(const float* __restrict__ src, char* __restrict__ dst,
                                                              int N, int C, int H, int W, int KH, int KW, int stride,
                                                              int pad, int Ho, int Wo, int kp) {
  const int groups = kp >> 3;
  const long total = (long)N * Ho * Wo * groups;
  const int K = KH * KW * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int gidx = (int)(i % groups);
    const long m = i / groups;
    const int ox = (int)(m % Wo);
    const int oy = (int)((m / Wo) % Ho);
    const int n = (int)(m / ((long)Wo * Ho));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = gidx * 8 + j;
      float x = 0.f;
      if (k < K) {
        const int c = k % C, t = k / C;
        const int ky = t / KW, kx = t - ky * KW;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) x = src[(((long)n * C + c) * H + iy) * W + ix];
      }
      v[j] = x;
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = pack_bf16(v[2 * j], v[2 * j + 1]);
      l[j] = pack_bf16(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u));
    }
    const int c8 = gidx * 8;
    char* d = dst + m * 4L * kp + (long)(c8 >> 5) * 128 + (c8 & 31) * 2;
    *(uint4*)d = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(d + 64) = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

// Weight preparation of a trainable convolution in ONE launch: FrozenBN fold (per-output-channel scale), the tap-major
// matrix form and the operand split -- and the same for the transposed matrix the data gradient multiplies with.
//   w [N, C, T] f32 (T = KH*KW taps, contiguous)  ->  fwd  [N, T*C] pair rows, k  = t*C + c   (forward / dW layout)
//                                                      bwd  [C, T*N] pair rows, k' = t*N + n   (dX operand)
// both holding w[n, c, t] * scale[n] (scale may be null).  Replaces mul + permute-copy + split (+ transpose-copy +
// split in the backward) per convolution and step; weights are at most a few MB, the kernel is latency-trivial.
__global__ __launch_bounds__(256) void weight_prep_pair_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                              char* __restrict__ fwd, char* __restrict__ bwd, int N, int C,
                                                              int T) {
  const long nf = (long)N * T * (C >> 3);            // 8-value groups of the forward matrix
  const long nb = bwd ? (long)C * T * (N >> 3) : 0;  // of the transposed matrix
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nf + nb; i += (long)gridDim.x * 256) {
    float v[8];
    char* d;
    if (i < nf) {
      const int cg = (int)(i % (C >> 3));
      const long r = i / (C >> 3);
      const int t = (int)(r % T), n = (int)(r / T);
      const float sc = scale ? scale[n] : 1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = w[((long)n * C + cg * 8 + j) * T + t] * sc;
      const int k = t * C + cg * 8;
      d = fwd + (long)n * 4L * T * C + (long)(k >> 5) * 128 + (k & 31) * 2;
    } else {
      const long ib = i - nf;
      const int ng = (int)(ib % (N >> 3));
      const long r = ib / (N >> 3);
      const int t = (int)(r % T), c = (int)(r / T);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = ng * 8 + j;
        v[j] = w[((long)n * C + c) * T + t] * (scale ? scale[n] : 1.f);
      }
      const int k = t * N + ng * 8;
      d = bwd + (long)c * 4L * T * N + (long)(k >> 5) * 128 + (k & 31) * 2;
    }
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = pack_bf16(v[2 * j], v[2 * j + 1]);
      l[j] = pack_bf16(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u));
    }
    *(uint4*)d = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(d + 64) = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

// Sum of the weight-gradient slabs of split_gemm_tn_kernel, times the FrozenBN scale of the output channel, written in
// the weight's own [N, C, T] order: dw[n, c, t] = scale[n] * sum_s slabs[s, n, t*C + c].
// Four consecutive lanes share one group of 4 channels and take every fourth slice each (eight loads in flight per lane,
// summed as a fixed tree, the four partial sums combined by two lane exchanges): the loop of S dependent loads per thread it
// replaces ran one workgroup per CU at a memory round trip per slice -- 18 us for the 33 MB of a [256 x 1024] gradient's 32
// slabs, 38 launches per teacher step.  The order of the additions is fixed, so the result is reproducible.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ scale,
                                                         float* __restrict__ dw, int S, int N, int C, int T) {
  const long total = (long)N * T * (C >> 2);
  const long slab = (long)N * T * C;
  const int part = threadIdx.x & 3;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 2; i < total; i += ((long)gridDim.x * 256) >> 2) {
    const int cg = (int)(i % (C >> 2));
    const long r = i / (C >> 2);
    const int t = (int)(r % T), n = (int)(r / T);
    const float* p = slabs + ((long)n * T + t) * C + cg * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = part; s0 < S; s0 += 32) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s_ = s0 + 4 * u;
        v[u] = s_ < S ? *(const f32x4*)(p + s_ * slab) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] += __shfl_xor(a[e], 1, 64);
      a[e] += __shfl_xor(a[e], 2, 64);
    }
    if (part == 0) {
      const float sc = scale ? scale[n] : 1.f;
      float* o = dw + ((long)n * C + cg * 4) * T + t;
      o[0] = a.x * sc; o[T] = a.y * sc; o[2 * T] = a.z * sc; o[3 * T] = a.w * sc;
    }
  }
}

// Pair-layout im2col (only for the weight gradient of a 3x3, which contracts over the rows): src [R,H,W,C] pair
// layout -> dst [R*H*W, T*C] pair layout, tap-major; out-of-map taps are zero rows.  Pure 16-byte copies: a thread
// moves one 16-byte slot of one (pixel, tap).  4 B/element read (L2-served re-reads), 4*T B/element written.
__global__ __launch_bounds__(256) void im2col_pair_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                         long pixels, int H, int W, int C, int KH, int KW) {
  const int slots = C >> 2;  // 16-byte slots per pixel row (4*C bytes)
  const int T = KH * KW;
  const long total = pixels * T * slots;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int s = (int)(i % slots);
    const long mt = i / slots;
    const int t = (int)(mt % T);
    const long m = mt / T;
    const int x = (int)(m % W), y = (int)((m / W) % H);
    const int yy = y + t / KW - KH / 2, xx = x + t % KW - KW / 2;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
      v = *(const uint4*)(src + (m + (long)(yy - y) * W + (xx - x)) * 4L * C + s * 16L);
    *(uint4*)(dst + (m * T + t) * 4L * C + s * 16L) = v;
  }
}
}  // namespace

extern "C" int ovis_split_pair_f32(const float* src, long src_row_stride, void* dst_pair, long rows, int cols,
                                   void* stream) {
  if (rows < 0 || cols < 0) return OVIS_EINVAL;
  if (rows == 0 || cols == 0) return OVIS_OK;
  if (!src || !dst_pair) return OVIS_EINVAL;
  if (cols % 32 != 0 || src_row_stride % 4 != 0 || ((uintptr_t)src & 15) || ((uintptr_t)dst_pair & 15))
    return OVIS_ERANGE;
  const long total = rows * (cols / 8);
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 8L * OVIS_NUM_CU ? blocks : 8L * OVIS_NUM_CU);
  hipLaunchKernelGGL(split_pair_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_row_stride,
                     (char*)dst_pair, rows, cols);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_gate_split_pair_f32(const float* dy, long dy_row_stride, const void* gate, int gate_is_pair,
                                        void* dst_pair, float* g_f32, long rows, int cols, const float* g_pooled,
                                        int pool_rows, void* stream) {
  return ovis_gate_split_pair_rows_f32(dy, dy_row_stride, gate, gate_is_pair, dst_pair, g_f32, rows, cols, g_pooled,
                                       pool_rows, nullptr, nullptr, stream);
}

extern "C" int ovis_gate_split_pair_rows_f32(const float* dy, long dy_row_stride, const void* gate, int gate_is_pair,
                                             void* dst_pair, float* g_f32, long rows, int cols, const float* g_pooled,
                                             int pool_rows, const float* g_selected, const int32_t* group_slot,
                                             void* stream) {
  if (rows < 0 || cols < 0) return OVIS_EINVAL;
  if (rows == 0 || cols == 0) return OVIS_OK;
  if ((!dy && !g_pooled && !g_selected) || !dst_pair || ((g_pooled || g_selected) && (pool_rows <= 0 || rows % pool_rows != 0)) ||
      ((g_selected != nullptr) != (group_slot != nullptr)))
    return OVIS_EINVAL;
  if (cols % 32 != 0 || dy_row_stride % 4 != 0 || ((uintptr_t)dy & 15) || ((uintptr_t)dst_pair & 15) ||
      ((uintptr_t)gate & 15) || ((uintptr_t)g_f32 & 15) || ((uintptr_t)g_pooled & 15) || ((uintptr_t)g_selected & 15))
    return OVIS_ERANGE;
  const float pool_scale = g_pooled ? 1.f / (float)pool_rows : 0.f;
  const long total = rows * (cols / 8);
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 8L * OVIS_NUM_CU ? blocks : 8L * OVIS_NUM_CU);
  hipStream_t s = (hipStream_t)stream;
#define OVIS_GS(GP_, WF_)                                                                                          \
  hipLaunchKernelGGL((gate_split_pair_kernel<GP_, WF_>), dim3(grid), dim3(256), 0, s, dy, dy_row_stride, gate,      \
                     (char*)dst_pair, g_f32, rows, cols, g_pooled, pool_rows, pool_scale, g_selected, group_slot)
  if (gate_is_pair) { if (g_f32) OVIS_GS(true, true); else OVIS_GS(true, false); }
  else { if (g_f32) OVIS_GS(false, true); else OVIS_GS(false, false); }
#undef OVIS_GS
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_im2col_nchw_pair_f32(const float* src, void* dst_pair, int num, int channels, int height,
                                         int width, int kh, int kw, int stride, int pad, int k_padded, void* stream) {
  if (num < 0 || channels <= 0 || height <= 0 || width <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0)
    return OVIS_EINVAL;
  if (num == 0) return OVIS_OK;
  if (!src || !dst_pair) return OVIS_EINVAL;
  if (k_padded % 32 != 0 || k_padded < kh * kw * channels || ((uintptr_t)dst_pair & 15)) return OVIS_ERANGE;
  const int ho = (height + 2 * pad - kh) / stride + 1, wo = (width + 2 * pad - kw) / stride + 1;
  if (ho <= 0 || wo <= 0) return OVIS_EINVAL;
  const long total = (long)num * ho * wo * (k_padded / 8);
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 16L * OVIS_NUM_CU ? blocks : 16L * OVIS_NUM_CU);
  hipLaunchKernelGGL(im2col_nchw_pair_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (char*)dst_pair, num,
                     channels, height, width, kh, kw, stride, pad, ho, wo, k_padded);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_weight_prep_pair_f32(const float* weight, const float* scale, void* fwd_pair, void* bwd_pair,
                                        int out_channels, int in_channels, int taps, void* stream) {
  if (out_channels <= 0 || in_channels <= 0 || taps <= 0) return OVIS_EINVAL;
  if (!weight || !fwd_pair) return OVIS_EINVAL;
  if (in_channels % 32 != 0 || (bwd_pair && out_channels % 32 != 0) || ((uintptr_t)fwd_pair & 15) ||
      ((uintptr_t)bwd_pair & 15))
    return OVIS_ERANGE;
  const long total = (long)out_channels * taps * (in_channels / 8) + (bwd_pair ? (long)in_channels * taps * (out_channels / 8) : 0);
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 8L * OVIS_NUM_CU ? blocks : 8L * OVIS_NUM_CU);
  hipLaunchKernelGGL(weight_prep_pair_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, weight, scale, (char*)fwd_pair,
                     (char*)bwd_pair, out_channels, in_channels, taps);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_slab_reduce_f32(const float* slabs, const float* scale, float* dweight, int slices, int out_channels,
                                    int in_channels, int taps, void* stream) {
  if (slices <= 0 || out_channels <= 0 || in_channels <= 0 || taps <= 0) return OVIS_EINVAL;
  if (!slabs || !dweight) return OVIS_EINVAL;
  if (in_channels % 4 != 0 || ((uintptr_t)slabs & 15)) return OVIS_ERANGE;
  const long total = 4L * out_channels * taps * (in_channels / 4);   // four lanes per group of 4 channels
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 8L * OVIS_NUM_CU ? blocks : 8L * OVIS_NUM_CU);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slabs, scale, dweight, slices,
                     out_channels, in_channels, taps);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_im2col_pair(const void* src_pair, void* dst_pair, long num, int height, int width, int channels,
                                int kh, int kw, void* stream) {
  if (num < 0 || height <= 0 || width <= 0 || channels < 0 || kh <= 0 || kw <= 0 || !(kh & 1) || !(kw & 1))
    return OVIS_EINVAL;
  if (num == 0 || channels == 0) return OVIS_OK;
  if (!src_pair || !dst_pair) return OVIS_EINVAL;
  if (channels % 32 != 0 || ((uintptr_t)src_pair & 15) || ((uintptr_t)dst_pair & 15)) return OVIS_ERANGE;
  const long total = num * height * width * kh * kw * (channels / 4);
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 16L * OVIS_NUM_CU ? blocks : 16L * OVIS_NUM_CU);
  hipLaunchKernelGGL(im2col_pair_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)src_pair,
                     (char*)dst_pair, num * height * width, height, width, channels, kh, kw);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_split_gemm_tn_slices(long m, int n, int channels, int taps) {
  if (m <= 0 || n <= 0 || channels <= 0 || taps <= 0) return 1;
  const long tiles = (long)(n / 128) * ((long)taps * channels / 128);
  const long steps = (m + 31) / 32;
  if (tiles <= 0) return 1;
  // Two workgroups are resident per CU (64 KB of LDS each): the grid runs in rounds of 512 workgroups, and a last round
  // that is mostly empty costs as much as a full one (the 3x3 weight gradient of res5, 144 tiles: 8 slices = 1152
  // workgroups = 2.25 rounds ran at 75 %; 7 slices = 1008 fill two rounds to 98 %).  Among the slice counts of up to
  // ~4 rounds pick the one whose rounds are fullest, preferring about two rounds (enough parallelism, few slabs).
  const long slots = 2L * OVIS_NUM_CU;
  if (steps >= 128 && steps <= 1100 && tiles <= 64) {  // (the measured regime: medium tile counts go to the fill scorer)
    // Short contractions with few output tiles (the trainable trunk of the teacher step: M = 8400 / 33400 rows, 4-36
    // tiles): a workgroup's fixed costs (slab write, launch) and the slab reduction weigh as much as its k-steps, so ONE
    // round of workgroups with at least 16 k-steps each beats two rounds of shorter slices -- layer3's 3x3 (36 tiles):
    // 14 slices 43 us against 53 us with 28; its 1x1s (16 tiles): 16 slices 25 us against 28 with 32; layer2's 3x3
    // (9 tiles): 56 against 113 (profiles/r4_tn_slices_small_m.txt).
    long s = slots / tiles;
    if (s > steps / 16) s = steps / 16;
    if (s > 256) s = 256;
    return (int)(s < 1 ? 1 : s);
  }
  long s_max = (4 * slots + tiles - 1) / tiles;
  if (s_max > steps / 8) s_max = steps / 8;                              // at least 8 k-steps per slice
  if (s_max > 256) s_max = 256;
  if (s_max < 1) s_max = 1;
  long best = 1;
  double best_score = -1.0;
  for (long s = 1; s <= s_max; ++s) {
    const long blocks = s * tiles, rounds = (blocks + slots - 1) / slots;
    double score = (double)blocks / (double)(rounds * slots);            // fill of the rounds
    if (rounds == 1) score *= 0.97 * (double)blocks / (double)slots;     // a single round: no tail balancing, and an
                                                                         // underfilled one idles CUs twice over
    if (rounds > 2) score -= 0.01 * (double)(rounds - 2);                // slabs cost traffic: mild preference for 2 rounds
    if (score > best_score + 1e-9) { best_score = score; best = s; }
  }
  return (int)best;
}

extern "C" int ovis_split_gemm_pair_tn(const void* g_pair, long g_row_bytes, const void* x_pair, long x_row_bytes,
                                       float* c_slabs, int slices, long m, int n, int channels, int taps_h,
                                       int taps_w, int height, int width, void* stream) {
  if (m < 0 || n <= 0 || channels <= 0 || taps_h <= 0 || taps_w <= 0 || !(taps_h & 1) || !(taps_w & 1) || slices <= 0)
    return OVIS_EINVAL;
  if (!g_pair || !x_pair || !c_slabs || m > 0x7fffff00L) return OVIS_EINVAL;
  const int T = taps_h * taps_w;
  if (T > 1 && (height <= 0 || width <= 0)) return OVIS_EINVAL;
  if (n % 128 != 0 || channels % 128 != 0 ||
      g_row_bytes % 16 != 0 || x_row_bytes % 16 != 0 || ((uintptr_t)g_pair & 15) || ((uintptr_t)x_pair & 15) ||
      ((uintptr_t)c_slabs & 15))
    return OVIS_ERANGE;
  SplitGemmTnArgs p;
  p.G = (const char*)g_pair; p.g_rs = g_row_bytes; p.X = (const char*)x_pair; p.x_rs = x_row_bytes;
  p.C = c_slabs; p.M = m; p.N = n; p.ch = channels; p.T = T; p.H = height; p.W = width; p.KH = taps_h; p.KW = taps_w;
  p.slices = slices;
  const long steps = (m + 31) / 32;
  p.steps_per_slice = (int)((steps + slices - 1) / slices);
  const int tiles_i = n / 128, tiles_j = (int)((long)T * channels / 128);
  // An XCD owns a contiguous run of 64 tile ids of a slice: the operand whose column tile is the MAJOR index is read for
  // one or two of its column tiles per run, the other one whole.  Major = the operand with more DISTINCT column tiles
  // (the taps of a 3x3 re-read the same X tiles).  Same-box A/B (profiles/r5_tn_tile_order_ab.txt): the 3x3 weight
  // gradient fetches 27 % less from beyond L2 (2008 -> 1465 MB per launch, time unchanged: it is not fetch-bound), the
  // N = 2048 / K = 512 one runs 3 % faster G-major, the N = 512 / K = 2048 one 2 % slower (it keeps X-major).
  p.g_major = tiles_i >= channels / 128 ? 1 : 0;
  const long nblocks = (long)tiles_i * tiles_j * slices;
  if (nblocks > 0x7fffffffL) return OVIS_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  const int lds = 2 * 2 * 32 * 512;
  if (T > 1 && (long)height * width <= 64) {
    static bool attr_set_s = false;
    if (!attr_set_s) {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_tn_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_set_s = true;
    }
    hipLaunchKernelGGL((split_gemm_tn_kernel<true, true>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_i, tiles_j, (int)nblocks);
  } else if (T > 1) {
    static bool attr_set_c = false;
    if (!attr_set_c) {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_tn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_set_c = true;
    }
    hipLaunchKernelGGL(split_gemm_tn_kernel<true>, dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_i, tiles_j, (int)nblocks);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_tn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr_set = true;
    }
    hipLaunchKernelGGL(split_gemm_tn_kernel<false>, dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_i, tiles_j, (int)nblocks);
  }
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

// ---- launch plan of the NT kernel (shared by the launcher and the workspace query) ----
namespace {
struct SplitGemmPlan {
  int mode, narrow, stages, kslices, steps_per_slice, tiles_m, tiles_n;
};

// config: 0 = choose; otherwise a bit set for tests / A-B probes: 1 = one LDS stage, 2 = two stages, 4 = never use
// the halo form, 8 = no split-K; bits 8..15: that many K slices (tools/experiments/slices_probe.py); bit 16: the launch is
// co-scheduled with other work (choose slices for least total work, not least duration).
SplitGemmPlan split_gemm_plan(long m, int n, int channels, int channels2, int taps_h, int taps_w, int width, int config) {
  SplitGemmPlan q;
  const int T = taps_h * taps_w;
  q.narrow = n <= 64;                       // 64-column tiles (2 waves): layer1 / stem, no masked half tile
  const int bn = q.narrow ? 64 : 128;
  q.tiles_m = (int)((m + 127) / 128);
  q.tiles_n = (n + bn - 1) / bn;
  const long nb = (long)q.tiles_m * q.tiles_n;
  q.mode = PLAIN;
  if (T > 1) q.mode = (!q.narrow && !(config & 4) && (taps_h / 2) * width + taps_w / 2 <= kHalo && T <= 16) ? HALO : SHIFTED;
  const int cpb = channels / 32;
  const int units = q.mode == HALO ? cpb : T * cpb + channels2 / 32;   // what a K slice is counted in
  // split-K: a grid that leaves most CUs idle while every workgroup walks a long K
  q.kslices = 1;
  q.steps_per_slice = units;
  bool sliced_two_stage = false;   // the slice count below was chosen FOR the two-stage kernel
  if (!(config & 8) && !q.narrow && nb <= OVIS_NUM_CU / 2) {
    const int min_units = q.mode == HALO ? 1 : 8;
    long s = units / min_units;
    const long want = (2L * OVIS_NUM_CU + nb - 1) / nb;
    if (s > want) s = want;
    if (s > 64) s = 64;
    if (s >= 2) {
      q.steps_per_slice = (int)((units + s - 1) / s);
      q.kslices = (units + q.steps_per_slice - 1) / q.steps_per_slice;
    }
  } else if (!(config & 8) && (config & 0x10000) && !q.narrow && q.mode != HALO) {
    // CO-SCHEDULED launches (config bit 16: the frozen half of the student-teacher step, which runs on a side stream beside
    // the student backward's full-machine GEMMs): another stream fills the CUs a tail leaves idle, so what counts is the
    // launch's TOTAL work, not its duration alone -- the coarser cut of rounds 2-5 (fewer slabs to write and reduce).  Same-box
    // A/B of the pipelined student step: 33.4 ms with this rule, 33.8 ms with the latency model below on every launch.
    int sl = 1;
    if (nb <= OVIS_NUM_CU && units >= 64) sl = 4;
    else if (nb <= OVIS_NUM_CU && units >= 32) sl = 2;
    else if (nb < 3L * OVIS_NUM_CU && units >= 128) sl = 2;
    if (sl > 1) {
      q.steps_per_slice = (units + sl - 1) / sl;
      q.kslices = (units + q.steps_per_slice - 1) / q.steps_per_slice;
    }
  } else if (!(config & 8) && !q.narrow && q.mode != HALO && nb < 3L * OVIS_NUM_CU && units >= 16) {
    // Grids of one or two workgroups per CU that walk a long K (the 50 x 84 maps of layer3 and the RPN head: 66 row tiles;
    // the 100 x 167 maps of layer2: 261): the CUs with one workgroup more than the others set the kernel's duration, and a
    // round of resident workgroups that overflows by a few items pays a whole extra item (528 tiles x 288 k-steps of the RPN
    // head in two slices = 1056 items on 1024 one-stage slots: 509 us; in four slices on the 512 two-stage slots 421 us =
    // the library's rate).  The slice count is chosen by a cost model of the TWO-STAGE kernel fitted to
    // tools/experiments/slices_probe.py (profiles/r6_slices_probe.txt): an item of s k-steps takes 12.8 + 1.0 s us beside a
    // second workgroup on its CU and 8 + 0.785 s us alone (a lone workgroup reaches 64 % of a CU's rate); a launch is
    // floor(items / 512) full rounds plus one partial round, plus the slabs' write + reduce.
    const double slab_unit = (double)m * n * 4.0 / 34.4e6;
    double best = 1e30;
    int best_sl = 1;
    for (int sl : {1, 2, 3, 4, 6, 8}) {
      const int steps = (units + sl - 1) / sl;
      if (sl > 1 && steps < 8) break;
      const long items = nb * ((units + steps - 1) / steps);
      const long full = items / (2L * OVIS_NUM_CU), rest = items % (2L * OVIS_NUM_CU);
      const double t2 = 12.8 + 1.0 * steps, t1 = 8.0 + 0.785 * steps;
      const double t = full * t2 + (rest == 0 ? 0.0 : (rest > OVIS_NUM_CU ? t2 : t1)) + (sl > 1 ? 4.0 + 0.9 * sl * slab_unit : 0.0);
      if (t < best) { best = t; best_sl = sl; }
    }
    if (best_sl > 1) {
      q.steps_per_slice = (units + best_sl - 1) / best_sl;
      q.kslices = (units + q.steps_per_slice - 1) / q.steps_per_slice;
      sliced_two_stage = true;
    }
  }
  if (const int forced = (config >> 8) & 0xff; forced > 0 && !q.narrow) {
    const int sl = forced < units ? forced : units;
    q.steps_per_slice = (units + sl - 1) / sl;
    q.kslices = (units + q.steps_per_slice - 1) / q.steps_per_slice;
  }
  // Large grids: ONE LDS stage and three / four workgroups per CU instead of two double-buffered ones -- nothing
  // overlaps inside a workgroup, but more independent workgroups hide each other's load phases better (measured on
  // M = 50176: N = 2048 1x1 +9...17 %, 3x3 +3 %; on M = 100352: every 1x1 +7...16 %, 3x3 +7 %; grids below ~2 rounds
  // of two-stage workgroups lose with one stage -- except when they overflow the two-stage round by a few workgroups
  // and fit one single-stage round).
  const long total = nb * q.kslices;
  q.stages = 2;
  if (q.mode == HALO || q.narrow) q.stages = 1;
  else if (sliced_two_stage) q.stages = 2;
  else if (total >= 8L * OVIS_NUM_CU || (total >= 4L * OVIS_NUM_CU && (T > 1 || q.tiles_n >= 8))) q.stages = 1;
  else {
    const long slots2 = 2L * OVIS_NUM_CU, slots1 = 4L * OVIS_NUM_CU;
    if (total > slots2 && total <= slots2 + slots2 / 4 && total <= slots1) q.stages = 1;
    // two two-stage rounds + a few items against one single-stage round + a few (33400 x 512 x 128: 1044 tiles, 26.5 vs 29.7 us)
    if (total > slots1 && total <= slots1 + slots1 / 16) q.stages = 1;
  }
  if (config & 1) q.stages = 1;
  if ((config & 2) && q.mode != HALO && !q.narrow) q.stages = 2;
  return q;
}
}  // namespace

extern "C" size_t ovis_split_gemm_pair_workspace_bytes_ex(long m, int n, int channels, int channels2, int taps_h,
                                                          int taps_w, int width, int config) {
  if (m <= 0 || n <= 0 || channels <= 0 || taps_h <= 0 || taps_w <= 0 || config < 0 || config > 0x1ffff) return 0;
  const SplitGemmPlan q = split_gemm_plan(m, n, channels, channels2, taps_h, taps_w, width, config);
  return q.kslices > 1 ? (size_t)q.kslices * (size_t)m * (size_t)n * sizeof(float) : 0;
}

extern "C" size_t ovis_split_gemm_pair_workspace_bytes(long m, int n, int channels, int channels2, int taps_h,
                                                       int taps_w, int width) {
  return ovis_split_gemm_pair_workspace_bytes_ex(m, n, channels, channels2, taps_h, taps_w, width, 0);
}

static int split_gemm_pair_impl(const void* a_pair, long a_row_bytes, const void* a2_pair, long a2_row_bytes,
                                const void* b_pair, long b_row_bytes, float* c, long ldc, void* c_pair,
                                long c_pair_row_bytes, const float* bias, const float* residual, long ldr,
                                const void* residual_pair, long residual_pair_row_bytes,
                                const void* gate_pair, long gate_row_bytes, long m, int n, int channels, int channels2,
                                int taps_h, int taps_w, int height, int width, int flip, int relu, void* workspace,
                                size_t workspace_bytes, int config, void* stream, float* pool = nullptr, int pool_rows = 0,
                                float pool_scale = 1.f) {
  if (m < 0 || n < 0 || channels < 0 || channels2 < 0 || taps_h <= 0 || taps_w <= 0 || !(taps_h & 1) || !(taps_w & 1))
    return OVIS_EINVAL;
  if (m == 0 || n == 0) return OVIS_OK;
  if (!a_pair || !b_pair || (!c && !c_pair && !pool) || channels == 0 || m > 0x7fffff00L) return OVIS_EINVAL;
  if (pool && (pool_rows < 32 || pool_rows > 64 || ((uintptr_t)pool & 15))) return OVIS_ERANGE;
  const int T = taps_h * taps_w;
  if (T > 1 && (height <= 0 || width <= 0 || height > 32767 || width > 32767 || channels2 != 0)) return OVIS_EINVAL;
  if (channels2 != 0 && !a2_pair) return OVIS_EINVAL;
  // one epilogue operand -- or the pair shortcut together with a gate (the DUAL kernels, plain products only)
  const bool dual = residual_pair && gate_pair && !residual;
  if (!dual && (residual != nullptr) + (residual_pair != nullptr) + (gate_pair != nullptr) > 1) return OVIS_EINVAL;
  if (dual && T > 1) return OVIS_ERANGE;
  if (channels % 32 != 0 || channels2 % 32 != 0 || n % 4 != 0 || (c_pair && n % 32 != 0) || a_row_bytes % 16 != 0 ||
      a2_row_bytes % 16 != 0 || b_row_bytes % 16 != 0 || ((uintptr_t)a_pair & 15) || ((uintptr_t)a2_pair & 15) ||
      ((uintptr_t)b_pair & 15) || ((uintptr_t)c & 15) || ((uintptr_t)c_pair & 15) || ((uintptr_t)bias & 15) ||
      ((uintptr_t)residual & 15) || ((uintptr_t)workspace & 15) || ldc % 4 != 0 || ldr % 4 != 0 ||
      c_pair_row_bytes % 16 != 0)
    return OVIS_ERANGE;
  if (config < 0 || config > 0x1ffff) return OVIS_ERANGE;
  SplitGemmPlan q = split_gemm_plan(m, n, channels, channels2, taps_h, taps_w, width, config);
  if (q.kslices > 1 && (!workspace || workspace_bytes < (size_t)q.kslices * (size_t)m * (size_t)n * sizeof(float))) {
    // no (or too small a) workspace: the un-split grid
    q = split_gemm_plan(m, n, channels, channels2, taps_h, taps_w, width, config | 8);
  }
  SplitGemmArgs p;
  p.A = (const char*)a_pair; p.a_rs = a_row_bytes;
  p.A2 = (const char*)a2_pair; p.a2_rs = a2_row_bytes;
  p.B = (const char*)b_pair; p.b_rs = b_row_bytes;
  p.C = c; p.ldc = ldc; p.Cp = (char*)c_pair; p.cp_rs = c_pair_row_bytes;
  p.bias = bias; p.res = residual; p.ldr = ldr;
  p.resp = (const char*)residual_pair; p.resp_rs = residual_pair_row_bytes;
  p.gate = (const char*)gate_pair; p.gate_rs = gate_row_bytes;
  p.slab = q.kslices > 1 ? (float*)workspace : nullptr;
  p.pool = pool; p.pool_rows = pool_rows; p.pool_scale = pool_scale;
  p.M = m; p.N = n; p.ch = channels; p.ch2 = channels2; p.T = T; p.H = height; p.W = width; p.KH = taps_h; p.KW = taps_w;
  p.flip = flip; p.relu = relu;
  p.kslices = q.kslices; p.steps_per_slice = q.steps_per_slice;
  p.gw = (q.tiles_n % kGroupWidth == 0) ? kGroupWidth : (q.tiles_n % 4 == 0) ? 4 : q.tiles_n;   // column groups of 4 weight tiles (2 MB) stay in an XCD's L2
  const long ntiles = (long)q.tiles_m * q.tiles_n;
  const long nblocks = ntiles * q.kslices;
  if (nblocks > 0x7fffffffL) return OVIS_ERANGE;
  hipStream_t s = (hipStream_t)stream;
#define OVIS_SG_LAUNCH__(WM_, WN_, MODE_, NS_, OCC_, DUAL_, POOL_)                                                   \
  do {                                                                                                              \
    constexpr int a_rows = MODE_ == HALO ? WM_ * 64 + 2 * kHalo + 2 : WM_ * 64;                                      \
    constexpr int lds = NS_ * (a_rows * 128 + WN_ * 64 * 128);                                                       \
    static bool attr_set = false;                                                                                   \
    if (!attr_set) {                                                                                                \
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<WM_, WN_, MODE_, NS_, OCC_, DUAL_, POOL_>,    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));                           \
      attr_set = true;                                                                                              \
    }                                                                                                               \
    hipLaunchKernelGGL((split_gemm_kernel<WM_, WN_, MODE_, NS_, OCC_, DUAL_, POOL_>), dim3((unsigned)nblocks),      \
                       dim3(WM_ * WN_ * 64), lds, s, p, q.tiles_n, (int)ntiles);                                    \
  } while (0)
#define OVIS_SG_LAUNCH_(WM_, WN_, MODE_, NS_, OCC_, DUAL_) OVIS_SG_LAUNCH__(WM_, WN_, MODE_, NS_, OCC_, DUAL_, false)
#define OVIS_SG_LAUNCH(WM_, WN_, MODE_, NS_, OCC_) OVIS_SG_LAUNCH_(WM_, WN_, MODE_, NS_, OCC_, false)
  if (pool) {
    // pooled epilogue: plain 128-column products without split-K (124 VGPRs: still four workgroups per CU)
    if (dual || q.narrow || q.mode != PLAIN || q.kslices > 1) return OVIS_ERANGE;
    if (q.stages == 1) OVIS_SG_LAUNCH__(2, 2, PLAIN, 1, 4, false, true); else OVIS_SG_LAUNCH__(2, 2, PLAIN, 2, 2, false, true);
  } else if (dual) {
    if (q.narrow || q.mode != PLAIN || q.kslices > 1) return OVIS_ERANGE;
    if (q.stages == 1) OVIS_SG_LAUNCH_(2, 2, PLAIN, 1, 4, true); else OVIS_SG_LAUNCH_(2, 2, PLAIN, 2, 2, true);
  } else if (q.narrow) {
    if (q.mode == PLAIN) OVIS_SG_LAUNCH(2, 1, PLAIN, 1, 2); else OVIS_SG_LAUNCH(2, 1, SHIFTED, 1, 2);
  } else if (q.mode == HALO) {
    OVIS_SG_LAUNCH(2, 2, HALO, 1, 4);
  } else if (q.mode == SHIFTED) {
    if (q.stages == 1) OVIS_SG_LAUNCH(2, 2, SHIFTED, 1, 4); else OVIS_SG_LAUNCH(2, 2, SHIFTED, 2, 2);
  } else {
    if (q.stages == 1) OVIS_SG_LAUNCH(2, 2, PLAIN, 1, 4); else OVIS_SG_LAUNCH(2, 2, PLAIN, 2, 2);
  }
#undef OVIS_SG_LAUNCH__
#undef OVIS_SG_LAUNCH_
#undef OVIS_SG_LAUNCH
  OVIS_LAUNCH_CHECK();
  if (q.kslices > 1) {
    p.slab = (float*)workspace;
    const long total = m * (n / 4);
    const long blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 8L * OVIS_NUM_CU ? blocks : 8L * OVIS_NUM_CU);
    hipLaunchKernelGGL(split_gemm_finish_kernel, dim3(grid), dim3(256), 0, s, p);
    OVIS_LAUNCH_CHECK();
  }
  return OVIS_OK;
}

extern "C" int ovis_split_gemm_pair(const void* a_pair, long a_row_bytes, const void* a2_pair, long a2_row_bytes,
                                    const void* b_pair, long b_row_bytes, float* c, long ldc, void* c_pair,
                                    long c_pair_row_bytes, const float* bias, const float* residual, long ldr, long m,
                                    int n, int channels, int channels2, int taps_h, int taps_w, int height, int width,
                                    int flip, int relu, void* workspace, size_t workspace_bytes, int config,
                                    void* stream) {
  return split_gemm_pair_impl(a_pair, a_row_bytes, a2_pair, a2_row_bytes, b_pair, b_row_bytes, c, ldc, c_pair,
                              c_pair_row_bytes, bias, residual, ldr, nullptr, 0, nullptr, 0, m, n, channels, channels2,
                              taps_h, taps_w, height, width, flip, relu, workspace, workspace_bytes, config, stream);
}

// The same with the shortcut operand given in pair layout ([m, n] pair rows, n % 32 == 0): residual = hi + lo, exact in
// fp32.  A bottleneck holds its input as the pair operand of conv1 anyway, so the identity shortcut needs no fp32 copy of
// the block input -- the producing block writes its result in pair layout only (4 instead of 8 bytes per element).
extern "C" int ovis_split_gemm_pair_rp(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                       float* c, long ldc, void* c_pair, long c_pair_row_bytes, const float* bias,
                                       const void* residual_pair, long residual_pair_row_bytes, long m, int n,
                                       int channels, int relu, void* workspace, size_t workspace_bytes, int config,
                                       void* stream) {
  if (!residual_pair || n % 32 != 0 || residual_pair_row_bytes % 16 != 0 || ((uintptr_t)residual_pair & 15)) return OVIS_ERANGE;
  return split_gemm_pair_impl(a_pair, a_row_bytes, nullptr, 0, b_pair, b_row_bytes, c, ldc, c_pair, c_pair_row_bytes, bias,
                              nullptr, 0, residual_pair, residual_pair_row_bytes, nullptr, 0, m, n, channels, 0, 1, 1, 0, 0,
                              0, relu, workspace, workspace_bytes, config, stream);
}

// conv3 + pair shortcut + ReLU of the LAST bottleneck of a res5 chain with the head's average pooling in the epilogue:
// pooled [m / pool_rows, n] (ZERO-FILLED by the caller) += pool_scale * result, summed over the pool_rows rows of every
// map; c and c_pair may both be NULL (nothing but the pooled rows is produced).  OVIS_ERANGE when the launch plan of this
// shape is not the plain un-split 128-column form (ovis_split_gemm_pair_pool_supported tells in advance).
extern "C" int ovis_split_gemm_pair_pool_supported(long m, int n, int channels, int pool_rows) {
  if (m <= 0 || n <= 0 || channels <= 0 || channels % 32 || n % 32 || pool_rows < 32 || pool_rows > 64) return 0;
  const SplitGemmPlan q = split_gemm_plan(m, n, channels, 0, 1, 1, 0, 0);   // small-M shapes keep their split-K plan
  return !q.narrow && q.mode == PLAIN && q.kslices == 1;
}

extern "C" int ovis_split_gemm_pair_rp_pool(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                            float* c, long ldc, void* c_pair, long c_pair_row_bytes, const float* bias,
                                            const void* residual_pair, long residual_pair_row_bytes, long m, int n,
                                            int channels, int relu, float* pooled, int pool_rows, float pool_scale,
                                            void* stream) {
  if (!pooled) return OVIS_EINVAL;
  if (n % 32 != 0 || (residual_pair && (residual_pair_row_bytes % 16 != 0 || ((uintptr_t)residual_pair & 15)))) return OVIS_ERANGE;
  return split_gemm_pair_impl(a_pair, a_row_bytes, nullptr, 0, b_pair, b_row_bytes, c, ldc, c_pair, c_pair_row_bytes, bias,
                              nullptr, 0, residual_pair, residual_pair_row_bytes, nullptr, 0, m, n, channels, 0, 1, 1, 0, 0,
                              0, relu, nullptr, 0, 8, stream, pooled, pool_rows, pool_scale);
}

extern "C" int ovis_split_gemm_pair_gated(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                          float* c, long ldc, void* c_pair, long c_pair_row_bytes,
                                          const void* gate_pair, long gate_row_bytes, long m, int n, int channels,
                                          int taps_h, int taps_w, int height, int width, int flip, int config,
                                          void* stream) {
  if (!gate_pair || gate_row_bytes % 16 != 0 || ((uintptr_t)gate_pair & 15) || n % 32 != 0) return OVIS_ERANGE;
  return split_gemm_pair_impl(a_pair, a_row_bytes, nullptr, 0, b_pair, b_row_bytes, c, ldc, c_pair, c_pair_row_bytes,
                              nullptr, nullptr, 0, nullptr, 0, gate_pair, gate_row_bytes, m, n, channels, 0, taps_h, taps_w,
                              height, width, flip, 0, nullptr, 0, config | 8, stream);
}

// The same with a split-K workspace (ovis_split_gemm_pair_workspace_bytes of the same problem): under-filled grids -- the
// data gradients of the teacher step's trainable trunk, 132 tiles walking 72 k-steps -- take the plan's K slices like
// the forward products do; the gate is applied by the slab reduction's epilogue.
extern "C" int ovis_split_gemm_pair_gated_ws(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                             float* c, long ldc, void* c_pair, long c_pair_row_bytes,
                                             const void* gate_pair, long gate_row_bytes, long m, int n, int channels,
                                             int taps_h, int taps_w, int height, int width, int flip, void* workspace,
                                             size_t workspace_bytes, int config, void* stream) {
  if (!gate_pair || gate_row_bytes % 16 != 0 || ((uintptr_t)gate_pair & 15) || n % 32 != 0) return OVIS_ERANGE;
  return split_gemm_pair_impl(a_pair, a_row_bytes, nullptr, 0, b_pair, b_row_bytes, c, ldc, c_pair, c_pair_row_bytes,
                              nullptr, nullptr, 0, nullptr, 0, gate_pair, gate_row_bytes, m, n, channels, 0, taps_h, taps_w,
                              height, width, flip, 0, workspace, workspace_bytes, config, stream);
}

// The input gradient of an identity bottleneck, ready for the block below: (A @ B^T + shortcut gradient given in pair
// layout) * (block input > 0), written in pair layout (and / or fp32).  See include/ovis_hip.h.
extern "C" int ovis_split_gemm_pair_rp_gated(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                             float* c, long ldc, void* c_pair, long c_pair_row_bytes,
                                             const void* residual_pair, long residual_pair_row_bytes,
                                             const void* gate_pair, long gate_row_bytes, long m, int n, int channels,
                                             int config, void* stream) {
  if (!residual_pair || !gate_pair || n % 32 != 0 || residual_pair_row_bytes % 16 != 0 || gate_row_bytes % 16 != 0 ||
      ((uintptr_t)residual_pair & 15) || ((uintptr_t)gate_pair & 15))
    return OVIS_ERANGE;
  if (n < 128) return OVIS_ERANGE;  // the narrow-tile kernels have no DUAL form
  return split_gemm_pair_impl(a_pair, a_row_bytes, nullptr, 0, b_pair, b_row_bytes, c, ldc, c_pair, c_pair_row_bytes,
                              nullptr, nullptr, 0, residual_pair, residual_pair_row_bytes, gate_pair, gate_row_bytes, m, n,
                              channels, 0, 1, 1, 0, 0, 0, 0, nullptr, 0, config | 8, stream);
}

// Deformable convolution (v1 / modulated v2) forward as ONE implicit GEMM: rows = output pixels, K = (tap, channel), the
// A tile of every k-step sampled in the kernel (split_gemm_kernel<.., DEFORM, ..>) -- no column buffer.
// See include/ovis_hip.h.
extern "C" int ovis_deform_conv_implicit_f32(const float* input_nhwc, const float* offset, const float* mask,
                                             const void* weight_pair, long weight_row_bytes, const float* bias,
                                             float* output, long ldc, int batch, int channels, int height, int width,
                                             int out_channels, int out_h, int out_w, int kernel_h, int kernel_w,
                                             int stride_h, int stride_w, int pad_h, int pad_w, int dil_h, int dil_w,
                                             int deformable_group, void* stream) {
  if (batch < 0 || channels <= 0 || height <= 0 || width <= 0 || out_channels <= 0 || out_h <= 0 || out_w <= 0 ||
      kernel_h <= 0 || kernel_w <= 0 || stride_h <= 0 || stride_w <= 0 || dil_h <= 0 || dil_w <= 0 || deformable_group <= 0)
    return OVIS_EINVAL;
  const long m = (long)batch * out_h * out_w;
  if (m == 0) return OVIS_OK;
  if (!input_nhwc || !offset || !weight_pair || !output) return OVIS_EINVAL;
  if ((long)batch * height * width * channels > 0x7fffffffL) return OVIS_ERANGE;
  if (channels % deformable_group != 0 || (channels / deformable_group) % 32 != 0 || out_channels % 4 != 0 || ldc % 4 != 0 ||
      weight_row_bytes % 16 != 0 || m > 0x7fffff00L || ((uintptr_t)input_nhwc & 15) || ((uintptr_t)weight_pair & 15) ||
      ((uintptr_t)output & 15) || ((uintptr_t)bias & 15))
    return OVIS_ERANGE;
  SplitGemmArgs p = {};
  p.B = (const char*)weight_pair; p.b_rs = weight_row_bytes;
  p.C = output; p.ldc = ldc; p.bias = bias;
  p.M = m; p.N = out_channels; p.ch = channels; p.ch2 = 0; p.T = kernel_h * kernel_w; p.H = height; p.W = width;
  p.KH = kernel_h; p.KW = kernel_w; p.flip = 0; p.relu = 0;
  p.kslices = 1; p.steps_per_slice = p.T * (channels / 32);
  p.dimg = input_nhwc; p.doff = offset; p.dmask = mask;
  p.Ho = out_h; p.Wo = out_w; p.sh = stride_h; p.sw = stride_w; p.ph = pad_h; p.pw = pad_w; p.dlh = dil_h; p.dlw = dil_w;
  p.dg = deformable_group;
  const int tiles_m = (int)((m + 127) / 128), tiles_n = (out_channels + 127) / 128;
  p.gw = (tiles_n % kGroupWidth == 0) ? kGroupWidth : (tiles_n % 4 == 0) ? 4 : tiles_n;
  const long ntiles = (long)tiles_m * tiles_n;
  if (ntiles > 0x7fffffffL) return OVIS_ERANGE;
  constexpr int lds = 128 * 128 + 128 * 128;
  static bool attr_set = false;
  if (!attr_set) {
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<2, 2, DEFORM, 1, 2>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((split_gemm_kernel<2, 2, DEFORM, 1, 2>), dim3((unsigned)ntiles), dim3(256), lds, (hipStream_t)stream, p,
                     tiles_n, (int)ntiles);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
