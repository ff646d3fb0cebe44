// fp32-accurate GEMM / implicit-GEMM convolution on the bf16 matrix cores of gfx950, operands in "pair" layout.
//
//   C[M, N] = epilogue( sum_k A[m, k] * B[n, k] ),   A, B fp32 values carried as bf16 hi + lo
//
// x = hi + lo + O(2^-17 |x|), hi = bf16(x), lo = bf16(x - hi); the product keeps hi.hi + hi.lo + lo.hi, accumulated
// in fp32 by v_mfma_f32_16x16x32_bf16 (relative error ~4e-6, cf. 1.7e-6 for an fp32 GEMM).  Unlike the K-concatenated
// form ([hi | hi | lo] x [hi | lo | hi] through a library GEMM, split_bf16.hip) the three products share their
// operand fragments here: per 32-deep k-step a wave reads 4 fragments kinds (A_hi, A_lo, B_hi, B_lo) from LDS and
// issues 3 MFMAs per fragment pair, so the kernel moves 2/3 of the bytes and 4/9 of the LDS reads per MFMA of a plain
// bf16 GEMM over 3K -- that is what lets a simple one-barrier-per-k-step structure keep the matrix cores busy.
//
// PAIR LAYOUT (written by split_pair_kernel / im2col_pair_kernel / this kernel's epilogue): a row of K values
// (K % 32 == 0) is K/32 blocks of 128 bytes: [ hi(32 x bf16) | lo(32 x bf16) ].  One k-step of one row is one full
// 128-byte line, fetched by 8 lanes of a global_load_lds_dwordx4 (no VGPR round trip).
//
// IMPLICIT CONVOLUTION: with T = KH*KW > 1 the A operand is an NHWC tensor [R, H, W, ch] in pair layout and
// k = (tap, channel): row m = (r, y, x) reads pixel (y + dy, x + dx) of tap (dy, dx) ("same" zero padding, stride 1;
// `flip` negates the offsets = the data-gradient convolution).  Rows outside the map read a 128-byte line of zeros.
// No im2col matrix exists anywhere: the 9 shifted reads of a pixel hit L2.
//
// Tile: (WM x 64) x 128 per workgroup of 2*WM waves, wave tile 64 x 64 (4 x 4 MFMA tiles, 64 accumulator VGPRs),
// two LDS stages of BM x 128 B (A) + 128 x 128 B (B); the 16-byte chunk index of a row is XOR-ed with (row >> 1) & 7
// (applied to the SOURCE address of the LDS-DMA, so the LDS image stays lane-linear) which makes every ds_read_b128
// of a 16-row fragment conflict-free.  Accumulators are computed transposed (D[n][m]) so that a lane owns 4
// consecutive columns of C: 16-byte stores, float4 bias / residual reads, 8-byte pair stores.
// Workgroups are renumbered so that each XCD owns a contiguous range of tiles (column tiles of one row tile are
// co-resident on one L2: the A rows are fetched from HBM once).
#include <stdlib.h>

#include "ovis_common.h"

namespace {
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ char g_zero_line[128];  // zero-initialised: the line an out-of-map tap reads

struct SplitGemmArgs {
  const char* A; long a_rs;          // pair rows of `ch` values (bytes per row = 4 * ch for a dense tensor)
  const char* B; long b_rs;          // [N] pair rows of K = T * ch values
  float* C; long ldc;                // fp32 result (may be null)
  char* Cp; long cp_rs;              // pair result (may be null), bytes per row
  const float* bias; const float* res; long ldr;
  const char* gate; long gate_rs;    // optional ReLU gate of a backward pass: pair rows of the forward activation (hi > 0)
  long M; int N; int ch; int T; int H; int W; int KH; int KW; int flip; int relu; int gw;
  int bm_eff;  // rows a tile really covers (<= BM, % 8 == 0; = BM unless the OVIS_SG_BALANCE probe is on)
};

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

template <int WM, bool CONV, int NS, int ABL = 0>
__global__ __launch_bounds__(WM * 128, (NS == 1 && ABL == 7) ? 4 : 1) void split_gemm_kernel(SplitGemmArgs p, int tiles_n, int nblocks) {
  constexpr int BM = WM * 64, BN = 128, NW = WM * 2;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int AI = BM / 8 / NW;  // LDS-DMA instructions per wave per stage for A (8 rows each): 4
  constexpr int BI = BN / 8 / NW;  // for B: 4 (WM = 2) or 2 (WM = 4)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware renumbering (bijective for any nblocks): workgroup b runs on XCD b % 8
  int tile;
  {
    const int b = blockIdx.x, q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, loc = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // tile order: column tiles are taken in groups of `p.gw`; within a group row tile-major.  The workgroups resident
  // on one XCD then share gw weight tiles (gw x 128 rows x 4K bytes: L2-resident) while the A rows stream through.
  const int gw = p.gw, tiles_m = nblocks / tiles_n;
  const int per_group = tiles_m * gw;
  const int grp = tile / per_group;
  const int in_grp = tile - grp * per_group;
  const int tile_m = in_grp / gw, tile_n = grp * gw + (in_grp - tile_m * gw);
  const long m0 = (long)tile_m * p.bm_eff;
  const int n0 = tile_n * BN;

  // ---- per-lane load geometry (no memory reads: hipcc would wait vmcnt(0) on them inside the DMA pipeline) ----
  const int lrow = lane >> 3;        // row within an 8-row DMA piece
  const int lchunk = lane & 7;       // 16-byte slot within the 128-byte LDS row
  constexpr bool conv = CONV;
  long a_off[AI];                    // byte offset of the lane's source row (chunk swizzle included)
  int a_yx[AI];                      // (y << 16) | x of that row, for the tap bounds test
  int a_lds[AI];                     // wave-uniform LDS offset of the piece inside a stage
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int piece = wave * AI + i;           // 8-row piece of the A tile
    const int row = piece * 8 + lrow;
    int gm = (int)m0 + row;                    // M < 2^31 (checked by the launcher)
    if (gm > (int)p.M - 1) gm = (int)p.M - 1;  // tail rows: valid memory, results never stored
    const int chunk = lchunk ^ ((row >> 1) & 7);
    a_off[i] = (long)gm * p.a_rs + chunk * 16;
    a_lds[i] = piece * 1024;
    int x = 0, y = 0;
    if (conv) {
      x = gm % p.W;
      y = (gm / p.W) % p.H;
    }
    a_yx[i] = (y << 16) | x;
  }
  long b_off[BI];
  int b_lds[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int piece = wave * BI + i;
    const int row = piece * 8 + lrow;
    int gn = n0 + row;
    if (gn > p.N - 1) gn = p.N - 1;
    const int chunk = lchunk ^ ((row >> 1) & 7);
    b_off[i] = (long)gn * p.b_rs + chunk * 16;
    b_lds[i] = A_BYTES + piece * 1024;
  }
  const char* zero_src = g_zero_line + lchunk * 16;
  uint4 sink = make_uint4(0u, 0u, 0u, 0u);

  const int cpb = p.ch >> 5;         // 32-value blocks per tap
  const int nk = p.T * cpb;
  int ld_tap = 0, ld_cb = 0;         // (tap, channel block) of the next stage to load
  int dy = 0, dx = 0;
  auto set_tap = [&](int t) {
    if (conv) {
      dy = t / p.KW - p.KH / 2;
      dx = t % p.KW - p.KW / 2;
      if (p.flip) { dy = -dy; dx = -dx; }
    }
  };
  set_tap(0);

  // one LDS-DMA piece of the A / B tile of k-step kb into `stage` (the tap state is the one of kb; `advance` steps it)
  auto issue_a = [&](int stage, int kb, int i) {
    if ((wave * AI + i) * 8 >= p.bm_eff) return;  // wave-uniform: this 8-row piece lies beyond the shrunk tile
    char* base = smem + stage * STAGE;
    const long a_shift = conv ? ((long)dy * p.W + dx) * p.a_rs + (long)ld_cb * 128 : (long)kb * 128;
    const char* src = p.A + a_off[i] + a_shift;
    if (conv) {
      const int y = (a_yx[i] >> 16) + dy, x = (a_yx[i] & 0xffff) + dx;
      if ((unsigned)y >= (unsigned)p.H || (unsigned)x >= (unsigned)p.W) src = zero_src;
    }
    if (ABL == 5) {  // probe: the same bytes as plain 16-byte loads into registers (no LDS write)
      const uint4 v = *(const uint4*)src;
      sink.x ^= v.x; sink.y ^= v.y; sink.z ^= v.z; sink.w ^= v.w;
    } else {
      glds16(src, base + a_lds[i]);
    }
  };
  auto issue_b = [&](int stage, int kb, int i) {
    const char* src = p.B + b_off[i] + (long)kb * 128;
    if (ABL == 5) {
      const uint4 v = *(const uint4*)src;
      sink.x ^= v.x; sink.y ^= v.y; sink.z ^= v.z; sink.w ^= v.w;
    } else {
      glds16(src, smem + stage * STAGE + b_lds[i]);
    }
  };
  auto advance = [&]() {
    if (++ld_cb == cpb) {
      ld_cb = 0;
      set_tap(++ld_tap);
    }
  };
  auto issue = [&](int stage, int kb) {
#pragma unroll
    for (int i = 0; i < AI; ++i) issue_a(stage, kb, i);
#pragma unroll
    for (int i = 0; i < BI; ++i) issue_b(stage, kb, i);
    advance();
  };

  // ---- fragment read addresses ----
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fc = lane >> 4;
  int a_rd[4], b_rd[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int ra = wm * 64 + f * 16 + frow;
    a_rd[f] = ra * 128 + ((fc ^ ((ra >> 1) & 7)) << 4);
    const int rb = wn * 64 + f * 16 + frow;
    b_rd[f] = A_BYTES + rb * 128 + ((fc ^ ((rb >> 1) & 7)) << 4);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ABL == 6 (NS == 3): FRAGMENT DOUBLE BUFFERING.  One 8-wave workgroup per CU has every wave at the same barrier,
  // so the LDS reads of a k-step are exposed (no second workgroup fills the matrix cores meanwhile).  Here the
  // fragments of stage kb+1 are read into a second register set right after the barrier of step kb, and the 48 MFMAs
  // of step kb run on the set that was read one step earlier: the matrix cores never wait for LDS.  Three LDS
  // buffers: compute reads none, fragment reads take stage kb+1, the DMA fills stage kb+2.
  if (NS == 1) {
    // single LDS stage, three workgroups per CU: no overlap inside a workgroup (load -> barrier -> compute ->
    // barrier), the other two workgroups fill the gaps.  Probe (tile_m code 1128).
    for (int kb = 0; kb < nk; ++kb) {
      issue(0, kb);
      __syncthreads();
      const char* base = smem;
      bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        bh[f] = *(const bf16x8*)(base + b_rd[f]);
        ah[f] = *(const bf16x8*)(base + a_rd[f]);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        bl[f] = *(const bf16x8*)(base + (b_rd[f] ^ 64));
        al[f] = *(const bf16x8*)(base + (a_rd[f] ^ 64));
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
      __syncthreads();
    }
  } else
  if (NS == 3 && ABL == 6) {
    bf16x8 f0[16], f1[16];
    auto read_frags = [&](bf16x8* fr, int buf) {
      const char* base = smem + buf * STAGE;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fr[8 + f] = *(const bf16x8*)(base + b_rd[f]);
        fr[f] = *(const bf16x8*)(base + a_rd[f]);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fr[12 + f] = *(const bf16x8*)(base + (b_rd[f] ^ 64));
        fr[4 + f] = *(const bf16x8*)(base + (a_rd[f] ^ 64));
      }
    };
    auto mfmas = [&](const bf16x8* fr) {
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[8 + g], fr[f], acc[f][g], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[8 + g], fr[4 + f], acc[f][g], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[12 + g], fr[f], acc[f][g], 0, 0, 0);
    };
    issue(0, 0);
    __syncthreads();
    if (nk > 1) issue(1, 1);
    read_frags(f0, 0);
    int b1 = 1, b2 = 2;  // ring slots of stages kb+1 and kb+2
    for (int kb = 0; kb < nk; kb += 2) {
      if (kb + 1 < nk) {
        __syncthreads();                       // stage kb+1 landed everywhere; slot b2 (stage kb-1) no longer read
        if (kb + 2 < nk) issue(b2, kb + 2);
        read_frags(f1, b1);
      }
      mfmas(f0);
      b1 = b2; b2 = b2 == 2 ? 0 : b2 + 1;
      if (kb + 1 < nk) {
        if (kb + 2 < nk) {
          __syncthreads();
          if (kb + 3 < nk) issue(b2, kb + 3);
          read_frags(f0, b1);
        }
        mfmas(f1);
        b1 = b2; b2 = b2 == 2 ? 0 : b2 + 1;
      }
    }
  } else {
  // NS == 2: one __syncthreads per k-step (its fence drains this wave's DMAs: stage kb landed, stage kb-1's buffer
  // free), the next stage in flight under the MFMAs.  NS == 3: a ring with TWO stages in flight -- the wait is a
  // counted vmcnt that leaves the younger stage outstanding, and the barrier is a raw s_barrier (a __syncthreads
  // would drain the DMA queue: an LDS-DMA is a pending LDS write on the VM counter).
  issue(0, 0);
  if (NS == 3 && nk > 1) issue(1, 1);
  int st = 0;  // ring slot of stage kb
  for (int kb = 0; kb < nk; ++kb) {
    // ABL == 3 (A/B probe): the 8 DMA pieces of the next stage spread over the MFMA groups below (one piece per 4
    // MFMAs, order pinned by sched_barrier) instead of issued back to back after the barrier.  Measured 0-8 % SLOWER
    // than the burst on the res5 shapes (pinning the order costs more than the cheaper DMA issue slots give back).
    constexpr bool ILV = NS == 2 && ABL == 3;
    const bool more = kb + 1 < nk;
    if (NS == 2) {
      __syncthreads();
      if (!ILV && more && (ABL != 1 || kb == 0)) issue((kb + 1) & 1, kb + 1);
    } else {
      if (kb + 1 < nk && ABL != 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kb + 2 < nk && ABL != 1) issue(st == 0 ? 2 : st - 1, kb + 2);
    }
    const char* base = smem + st * STAGE;
    st = (st + 1 == NS) ? 0 : st + 1;
    if (ABL == 4 || ABL == 5) continue;  // ablation: the DMA / load stream alone (no LDS reads, no MFMAs)
    bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bh[f] = *(const bf16x8*)(base + b_rd[f]);
      ah[f] = *(const bf16x8*)(base + a_rd[f]);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bl[f] = *(const bf16x8*)(base + (b_rd[f] ^ 64));
      al[f] = *(const bf16x8*)(base + (a_rd[f] ^ 64));
    }
    if (ABL == 2) {  // ablation: loads + LDS reads only
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        acc[f][0] += __builtin_bit_cast(f32x4, ah[f]);
        acc[f][1] += __builtin_bit_cast(f32x4, al[f]);
        acc[f][2] += __builtin_bit_cast(f32x4, bh[f]);
        acc[f][3] += __builtin_bit_cast(f32x4, bl[f]);
      }
      continue;
    }
    if (ILV) {
      const int nst = (kb + 1) & 1;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[g], ah[f], acc[f][g], 0, 0, 0);
        if (more && f < AI) issue_a(nst, kb + 1, f);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[g], al[f], acc[f][g], 0, 0, 0);
        if (more && f < BI) issue_b(nst, kb + 1, f);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) advance();
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[g], ah[f], acc[f][g], 0, 0, 0);
      continue;
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
  }

  }  // legacy loop
  if (ABL == 5) acc[0][0].x += __uint_as_float(sink.x ^ sink.y ^ sink.z ^ sink.w);
  // ---- epilogue: lane owns row m = .. + (lane & 15), columns n = .. + (lane >> 4) * 4 + {0..3} of each tile ----
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const long m = m0 + wm * 64 + f * 16 + frow;
    if (m >= p.M || wm * 64 + f * 16 + frow >= p.bm_eff) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * 64 + g * 16 + fc * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[f][g];
      if (p.bias) {
        const f32x4 b = *(const f32x4*)(p.bias + n);
        v += b;
      }
      if (p.res) {
        const f32x4 r = *(const f32x4*)(p.res + m * p.ldr + n);
        v += r;
      }
      if (p.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (p.gate) {  // data gradient of a layer whose input went through a ReLU: zero where that activation was <= 0
        const uint2 h = *(const uint2*)(p.gate + m * p.gate_rs + (long)(n >> 5) * 128 + (n & 31) * 2);
        const unsigned a0 = h.x & 0xffffu, a1 = h.x >> 16, a2 = h.y & 0xffffu, a3 = h.y >> 16;
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
        char* d = p.Cp + m * p.cp_rs + (long)(n >> 5) * 128 + (n & 31) * 2;
        *(uint2*)d = make_uint2(h01, h23);
        *(uint2*)(d + 64) = make_uint2(l01, l23);
      }
    }
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

template <bool CONV, int NS = 2>
__global__ __launch_bounds__(256, NS == 1 ? 3 : 1) void split_gemm_tn_kernel(SplitGemmTnArgs p, int tiles_i, int tiles_j,
                                                                            int nblocks) {
  constexpr int TILE_BYTES = 32 * 512, STAGE = 2 * TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int id;
  {
    const int b = blockIdx.x, q = nblocks >> 3, r = nblocks & 7, xcd = b & 7, loc = b >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int per_slice = tiles_i * tiles_j;
  const int slice = id / per_slice;
  const int rem = id - slice * per_slice;
  const int tile_j = rem / tiles_i, tile_i = rem - tile_j * tiles_i;
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
  int x_y[4], x_x[4];   // map position of the X row, kept incrementally
  int m_row[4];
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
    x_x[i] = CONV ? m % p.W : 0;
    x_y[i] = CONV ? (m / p.W) % p.H : 0;
  }
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
      if (CONV) {
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

  // NS == 2: the next stage loads under this stage's MFMAs, two workgroups per CU.  NS == 1: one stage, no overlap
  // inside the workgroup, three workgroups per CU (<= 168 VGPRs) fill each other's load phases.
  if (NS == 2 && nk > 0) issue(0);
  for (int kb = 0; kb < nk; ++kb) {
    if (NS == 1) {
      if (kb > 0) __syncthreads();  // everyone is done reading the stage
      issue(0);
    }
    __syncthreads();
    if (NS == 2 && kb + 1 < nk) issue((kb + 1) & 1);
    const int so = NS == 1 ? 0 : (kb & 1) * STAGE;  // the dynamic LDS segment starts at LDS address 0 (no static LDS in this TU's kernels)
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
                                                             float pool_scale) {
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
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ scale,
                                                         float* __restrict__ dw, int S, int N, int C, int T) {
  const long total = (long)N * T * (C >> 2);
  const long slab = (long)N * T * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % (C >> 2));
    const long r = i / (C >> 2);
    const int t = (int)(r % T), n = (int)(r / T);
    const float* p = slabs + ((long)n * T + t) * C + cg * 4;
    f32x4 a = *(const f32x4*)p;
    for (int s_ = 1; s_ < S; ++s_) a += *(const f32x4*)(p + s_ * slab);
    const float sc = scale ? scale[n] : 1.f;
    float* o = dw + ((long)n * C + cg * 4) * T + t;
    o[0] = a.x * sc; o[T] = a.y * sc; o[2 * T] = a.z * sc; o[3 * T] = a.w * sc;
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
  if (rows < 0 || cols < 0) return OVIS_EINVAL;
  if (rows == 0 || cols == 0) return OVIS_OK;
  if ((!dy && !g_pooled) || !dst_pair || (g_pooled && (pool_rows <= 0 || rows % pool_rows != 0))) return OVIS_EINVAL;
  if (cols % 32 != 0 || dy_row_stride % 4 != 0 || ((uintptr_t)dy & 15) || ((uintptr_t)dst_pair & 15) ||
      ((uintptr_t)gate & 15) || ((uintptr_t)g_f32 & 15) || ((uintptr_t)g_pooled & 15))
    return OVIS_ERANGE;
  const float pool_scale = g_pooled ? 1.f / (float)pool_rows : 0.f;
  const long total = rows * (cols / 8);
  const long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < 8L * OVIS_NUM_CU ? blocks : 8L * OVIS_NUM_CU);
  hipStream_t s = (hipStream_t)stream;
#define OVIS_GS(GP_, WF_)                                                                                          \
  hipLaunchKernelGGL((gate_split_pair_kernel<GP_, WF_>), dim3(grid), dim3(256), 0, s, dy, dy_row_stride, gate,      \
                     (char*)dst_pair, g_f32, rows, cols, g_pooled, pool_rows, pool_scale)
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
  const long total = (long)out_channels * taps * (in_channels / 4);
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
  long slots = 2L * OVIS_NUM_CU;
  if (const char* e = getenv("OVIS_TN_STAGES")) { if (atoi(e) == 1) slots = 3L * OVIS_NUM_CU; }  // probe: single-stage variant
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
  const long nblocks = (long)tiles_i * tiles_j * slices;
  if (nblocks > 0x7fffffffL) return OVIS_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  int lds = 2 * 2 * 32 * 512;
  if (const char* e = getenv("OVIS_TN_STAGES")) {
    if (atoi(e) == 1) {
      lds = 2 * 32 * 512;
      if (T > 1) hipLaunchKernelGGL((split_gemm_tn_kernel<true, 1>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_i, tiles_j, (int)nblocks);
      else hipLaunchKernelGGL((split_gemm_tn_kernel<false, 1>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_i, tiles_j, (int)nblocks);
      OVIS_LAUNCH_CHECK();
      return OVIS_OK;
    }
  }
  if (T > 1) {
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

static int split_gemm_pair_impl(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                float* c, long ldc, void* c_pair, long c_pair_row_bytes, const float* bias,
                                const float* residual, long ldr, const void* gate_pair, long gate_row_bytes, long m,
                                int n, int channels, int taps_h, int taps_w, int height, int width, int flip, int relu,
                                int tile_m, void* stream);

extern "C" int ovis_split_gemm_pair(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                    float* c, long ldc, void* c_pair, long c_pair_row_bytes, const float* bias,
                                    const float* residual, long ldr, long m, int n, int channels, int taps_h,
                                    int taps_w, int height, int width, int flip, int relu, int tile_m,
                                    void* stream) {
  return split_gemm_pair_impl(a_pair, a_row_bytes, b_pair, b_row_bytes, c, ldc, c_pair, c_pair_row_bytes, bias, residual,
                              ldr, nullptr, 0, m, n, channels, taps_h, taps_w, height, width, flip, relu, tile_m, stream);
}

extern "C" int ovis_split_gemm_pair_gated(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                          float* c, long ldc, void* c_pair, long c_pair_row_bytes,
                                          const void* gate_pair, long gate_row_bytes, long m, int n, int channels,
                                          int taps_h, int taps_w, int height, int width, int flip, void* stream) {
  if (!gate_pair || gate_row_bytes % 16 != 0 || ((uintptr_t)gate_pair & 15) || n % 32 != 0) return OVIS_ERANGE;
  return split_gemm_pair_impl(a_pair, a_row_bytes, b_pair, b_row_bytes, c, ldc, c_pair, c_pair_row_bytes, nullptr,
                              nullptr, 0, gate_pair, gate_row_bytes, m, n, channels, taps_h, taps_w, height, width, flip,
                              0, 0, stream);
}

static int split_gemm_pair_impl(const void* a_pair, long a_row_bytes, const void* b_pair, long b_row_bytes,
                                float* c, long ldc, void* c_pair, long c_pair_row_bytes, const float* bias,
                                const float* residual, long ldr, const void* gate_pair, long gate_row_bytes, long m,
                                int n, int channels, int taps_h, int taps_w, int height, int width, int flip, int relu,
                                int tile_m, void* stream) {
  if (m < 0 || n < 0 || channels < 0 || taps_h <= 0 || taps_w <= 0 || !(taps_h & 1) || !(taps_w & 1))
    return OVIS_EINVAL;
  if (m == 0 || n == 0) return OVIS_OK;
  if (!a_pair || !b_pair || (!c && !c_pair) || channels == 0 || m > 0x7fffff00L) return OVIS_EINVAL;
  const int T = taps_h * taps_w;
  if (T > 1 && (height <= 0 || width <= 0 || height > 32767 || width > 32767)) return OVIS_EINVAL;
  if (channels % 32 != 0 || n % 4 != 0 || (c_pair && n % 32 != 0) || a_row_bytes % 16 != 0 || b_row_bytes % 16 != 0 ||
      ((uintptr_t)a_pair & 15) || ((uintptr_t)b_pair & 15) || ((uintptr_t)c & 15) || ((uintptr_t)c_pair & 15) ||
      ((uintptr_t)bias & 15) || ((uintptr_t)residual & 15) || ldc % 4 != 0 || ldr % 4 != 0 ||
      c_pair_row_bytes % 16 != 0)
    return OVIS_ERANGE;
  int stages = 2, abl = 0;
  if (tile_m >= 10000) { abl = tile_m / 10000; tile_m %= 10000; }  // ablation probes (tools/experiments only)
  if (tile_m >= 1000) { stages = tile_m / 1000; tile_m %= 1000; }
  if ((tile_m != 0 && tile_m != 128 && tile_m != 256) || (stages != 1 && stages != 2 && stages != 3)) return OVIS_ERANGE;
  SplitGemmArgs p;
  p.A = (const char*)a_pair; p.a_rs = a_row_bytes;
  p.B = (const char*)b_pair; p.b_rs = b_row_bytes;
  p.C = c; p.ldc = ldc; p.Cp = (char*)c_pair; p.cp_rs = c_pair_row_bytes;
  p.bias = bias; p.res = residual; p.ldr = ldr;
  p.gate = (const char*)gate_pair; p.gate_rs = gate_row_bytes;
  p.M = m; p.N = n; p.ch = channels; p.T = T; p.H = height; p.W = width; p.KH = taps_h; p.KW = taps_w;
  p.flip = flip; p.relu = relu;
  const int tiles_n = (n + 127) / 128;
  {
    int gw = 4;
    if (const char* e = getenv("OVIS_SG_GW")) gw = atoi(e);
    if (gw <= 0 || gw > tiles_n || tiles_n % gw != 0) gw = tiles_n;
    p.gw = gw;
  }
  hipStream_t s = (hipStream_t)stream;
  int bm = tile_m;
  if (bm == 0) {
    bm = 128;  // independent 4-wave workgroups beat one 8-wave 256-row workgroup per CU on every shape measured
    // Large grids: ONE LDS stage and three / four workgroups per CU instead of two double-buffered ones -- nothing
    // overlaps inside a workgroup, but more independent workgroups hide each other's load phases better (measured on
    // M = 50176: N = 2048 1x1 +9...17 %, 3x3 +3 %, N = 512 1x1 +-2 %; on M = 100352: every 1x1 +7...16 %, 3x3 +7 %;
    // grids below ~2 rounds lose: they keep 2 stages).
    const long nb = ((m + 127) / 128) * tiles_n;
    const char* e = getenv("OVIS_SG_STAGES");
    if (e) stages = atoi(e) == 1 ? 1 : 2;
    else if ((nb >= 8L * OVIS_NUM_CU && !getenv("OVIS_SG_NARROW")) || (nb >= 4L * OVIS_NUM_CU && (T > 1 || tiles_n >= 8))) stages = 1;
    if (stages == 1 && T == 1) abl = 7;  // the plain kernel fits 128 VGPRs: four workgroups per CU
  }
  // OVIS_SG_BALANCE=1 (probe): spread the rows over ceil(blocks / resident workgroups) FULL rounds of slightly shorter
  // tiles.  Measured 0-12 % SLOWER on the res5 shapes (M = 50176 / 100352): the last, mostly empty round of the
  // uniform tiling costs less than the extra weight-tile traffic of more, shorter row tiles.  Off by default.
  p.bm_eff = bm;
  if (bm == 128 && stages == 2 && !abl) {
    const long slots = 2L * OVIS_NUM_CU;
    const long blocks0 = ((m + 127) / 128) * tiles_n;
    const long rounds = (blocks0 + slots - 1) / slots;
    const char* e = getenv("OVIS_SG_BALANCE");
    if (!e || atoi(e) == 0) goto no_balance;
    {
      const long rt = rounds * slots / tiles_n;            // row tiles that fill `rounds` rounds
      if (rt > 0) {
        long be = ((m + rt - 1) / rt + 7) / 8 * 8;
        if (be >= 64 && be < 128) p.bm_eff = (int)be;
      }
    }
  }
no_balance:
  const long tiles_m = (m + p.bm_eff - 1) / p.bm_eff;
  const long nblocks = tiles_m * tiles_n;
  if (nblocks > 0x7fffffffL) return OVIS_ERANGE;
 #define OVIS_SG_LAUNCH(WM_, CONV_, NS_)                                                                         \
  do {                                                                                                          \
    constexpr int lds = NS_ * (WM_ * 64 * 128 + 128 * 128);                                                     \
    static bool attr_set = false;                                                                               \
    if (!attr_set) {                                                                                            \
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<WM_, CONV_, NS_>,                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));                       \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL((split_gemm_kernel<WM_, CONV_, NS_>), dim3((unsigned)nblocks), dim3(WM_ * 128), lds, s,  \
                       p, tiles_n, (int)nblocks);                                                               \
  } while (0)
  if (stages == 1 && bm == 128) {
    constexpr int lds1 = 128 * 128 + 128 * 128;
    if (abl == 7) {  // 4 waves per SIMD (<= 128 VGPRs): four workgroups per CU
      if (T > 1) hipLaunchKernelGGL((split_gemm_kernel<2, true, 1, 7>), dim3((unsigned)nblocks), dim3(256), lds1, s, p, tiles_n, (int)nblocks);
      else hipLaunchKernelGGL((split_gemm_kernel<2, false, 1, 7>), dim3((unsigned)nblocks), dim3(256), lds1, s, p, tiles_n, (int)nblocks);
    } else {
      if (T > 1) hipLaunchKernelGGL((split_gemm_kernel<2, true, 1, 0>), dim3((unsigned)nblocks), dim3(256), lds1, s, p, tiles_n, (int)nblocks);
      else hipLaunchKernelGGL((split_gemm_kernel<2, false, 1, 0>), dim3((unsigned)nblocks), dim3(256), lds1, s, p, tiles_n, (int)nblocks);
    }
  } else if (stages == 1) {
    return OVIS_ERANGE;
  } else if (abl == 6 && bm == 256 && stages == 3) {
    constexpr int lds3 = 3 * (256 * 128 + 128 * 128);
    if (T > 1) {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<4, true, 3, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
      hipLaunchKernelGGL((split_gemm_kernel<4, true, 3, 6>), dim3((unsigned)nblocks), dim3(512), lds3, s, p, tiles_n, (int)nblocks);
    } else {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<4, false, 3, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
      hipLaunchKernelGGL((split_gemm_kernel<4, false, 3, 6>), dim3((unsigned)nblocks), dim3(512), lds3, s, p, tiles_n, (int)nblocks);
    }
  } else if (abl && bm == 256 && stages == 3 && T == 1) {  // ablations of the 256-row, 3-stage ring (probe only)
    constexpr int lds3 = 3 * (256 * 128 + 128 * 128);
#define OVIS_ABL3(A_)                                                                                                    \
  do {                                                                                                                   \
    OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<4, false, 3, A_>,                                    \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds3));                                 \
    hipLaunchKernelGGL((split_gemm_kernel<4, false, 3, A_>), dim3((unsigned)nblocks), dim3(512), lds3, s, p, tiles_n,    \
                       (int)nblocks);                                                                                    \
  } while (0)
    if (abl == 1) OVIS_ABL3(1); else if (abl == 2) OVIS_ABL3(2); else OVIS_ABL3(4);
#undef OVIS_ABL3
  } else if (abl) {
    constexpr int lds = 2 * (128 * 128 + 128 * 128);
    if (abl == 1) {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<2, false, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL((split_gemm_kernel<2, false, 2, 1>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_n, (int)nblocks);
    } else if (abl == 4) {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<2, false, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL((split_gemm_kernel<2, false, 2, 4>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_n, (int)nblocks);
    } else if (abl == 5) {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<2, false, 2, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL((split_gemm_kernel<2, false, 2, 5>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_n, (int)nblocks);
    } else if (abl == 3) {
      if (T > 1) {
        OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<2, true, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((split_gemm_kernel<2, true, 2, 3>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_n, (int)nblocks);
      } else {
        OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<2, false, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((split_gemm_kernel<2, false, 2, 3>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_n, (int)nblocks);
      }
    } else {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)split_gemm_kernel<2, false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL((split_gemm_kernel<2, false, 2, 2>), dim3((unsigned)nblocks), dim3(256), lds, s, p, tiles_n, (int)nblocks);
    }
  } else if (bm == 256) {
    if (stages == 3) { if (T > 1) OVIS_SG_LAUNCH(4, true, 3); else OVIS_SG_LAUNCH(4, false, 3); }
    else { if (T > 1) OVIS_SG_LAUNCH(4, true, 2); else OVIS_SG_LAUNCH(4, false, 2); }
  } else {
    if (stages == 3) { if (T > 1) OVIS_SG_LAUNCH(2, true, 3); else OVIS_SG_LAUNCH(2, false, 3); }
    else { if (T > 1) OVIS_SG_LAUNCH(2, true, 2); else OVIS_SG_LAUNCH(2, false, 2); }
  }
#undef OVIS_SG_LAUNCH
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
