// Device-side training-target construction for the RoI heads: the chains of ~100 tiny tensor ops per
// image that the reference runs for box matching / delta encoding and for the mask targets, as ONE launch each.
// Both follow the reference's fp32 operation order (FP contraction is off for this library), so they reproduce the
// tensor-op formulation value for value (log / division are the device's correctly rounded forms).
//
//   match_encode : boxlist_iou (mb/structures/boxlist_ops.py:53-89) -> Matcher (mb/modeling/matcher.py:42-81, no
//                  low-quality matches) -> labels (box_head/loss.py:46-77, mask_head/loss.py:60-77) -> BoxCoder.encode
//                  (mb/modeling/box_coder.py:22-52).  HBM-trivial; latency-bound: one thread per proposal, the G
//                  ground-truth boxes are broadcast reads.
//   project_masks: project_masks_on_boxes (mask_head/loss.py:11-42) for binary 'mask'-mode targets: crop to the
//                  rounded box (segmentation_mask.py:117-136) and bilinear resize to MxM (F.interpolate,
//                  align_corners=False; :138-156), cast back to the mask dtype.  One thread per output pixel.
//   sample_fg_bg : BalancedPositiveNegativeSampler (mb/modeling/balanced_positive_negative_sampler.py:19-68) for one
//                  image in ONE launch: counts, a uniformly random num_pos-subset of the positives and num_neg-subset of
//                  the negatives (the reference's randperm()[:k] picks exactly such subsets; the random stream itself
//                  is torch-version dependent and cannot be matched), compacted in ascending index order like the
//                  reference's nonzero(pos_mask | neg_mask) -- no nonzero / randperm / host read.
//   project_pasted_masks: project_masks on masks that exist only as (MxM probability map, box) pairs -- the pseudo
//                  labels' masks (st_generalized_rcnn.py:266-271).  A pixel of the full-resolution binary mask the
//                  reference builds with Masker (mask_head/inference.py:100-160: pad by 1, expand the box, bilinear
//                  resize to the integer box, > 0.5, paste) is evaluated on the fly, so the H x W canvases never exist.
#include "ovis_common.h"

namespace {
__global__ __launch_bounds__(256) void match_encode_kernel(const float* __restrict__ gt, const long* __restrict__ gt_labels,
                                                          const float* __restrict__ prop, int G, int P, float high,
                                                          float low, int between_label_mode, float wx, float wy,
                                                          float ww, float wh, long* __restrict__ matched_idx,
                                                          long* __restrict__ labels, float* __restrict__ reg) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const float4 b = *(const float4*)(prop + 4 * p);
  const float area_b = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  float best = -1.f;
  int arg = 0;
  for (int g = 0; g < G; ++g) {
    const float4 a = *(const float4*)(gt + 4 * g);
    const float area_a = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x) + 1.f, 0.f);
    const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y) + 1.f, 0.f);
    const float inter = w * h;
    const float iou = inter / (area_a + area_b - inter);
    if (iou > best) {  // first maximum wins
      best = iou;
      arg = g;
    }
  }
  long lab = gt_labels[arg];
  int idx = arg;
  if (best < low) {
    lab = 0;
    idx = 0;  // matched.clamp(min=0) of BELOW_LOW_THRESHOLD
  } else if (best < high) {
    idx = 0;  // clamp(min=0) of BETWEEN_THRESHOLDS
    lab = between_label_mode == 0 ? -1 : gt_labels[0];  // box head: ignore (-1); mask head: the label of gt 0, as upstream
  }
  matched_idx[p] = idx;
  labels[p] = lab;
  if (reg) {
    const float4 a = *(const float4*)(gt + 4 * idx);
    const float ex_w = b.z - b.x + 1.f, ex_h = b.w - b.y + 1.f;
    const float ex_cx = b.x + 0.5f * ex_w, ex_cy = b.y + 0.5f * ex_h;
    const float gt_w = a.z - a.x + 1.f, gt_h = a.w - a.y + 1.f;
    const float gt_cx = a.x + 0.5f * gt_w, gt_cy = a.y + 0.5f * gt_h;
    float4 r;
    r.x = wx * (gt_cx - ex_cx) / ex_w;
    r.y = wy * (gt_cy - ex_cy) / ex_h;
    r.z = ww * logf(gt_w / ex_w);
    r.w = wh * logf(gt_h / ex_h);
    *(float4*)(reg + 4 * p) = r;
  }
}

// ---- RPN targets: anchors x ground truth with the Matcher's LOW-QUALITY matches (rpn/loss.py:21-89; matcher.py:42-112) --------
// IoU of one anchor / ground-truth pair, the expression of boxlist_iou (structures/boxlist_ops.py:53-89); both passes below
// evaluate it with the same instructions, so "IoU == the ground truth's best IoU" is an exact comparison as in the reference.
__device__ __forceinline__ float anchor_iou(const float4 a, const float area_a, const float4 b, const float area_b) {
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x) + 1.f, 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y) + 1.f, 0.f);
  const float inter = w * h;
  return inter / (area_a + area_b - inter);
}

// pass 1: best[g] = max over the anchors of IoU(g, anchor) (bit pattern of a non-negative float: unsigned max).  A block
// walks a strided share of the anchors and issues ONE atomic per ground truth: all G words share a cache line, and with an
// atomic per wave (985 waves x G at 63 000 anchors) the kernel spent 80 us queueing on that line.
__global__ __launch_bounds__(256) void rpn_best_per_gt_kernel(const float* __restrict__ gt, const float* __restrict__ anchors,
                                                             int G, int A, unsigned* __restrict__ best) {
  __shared__ float red[4];
  for (int g = 0; g < G; ++g) {
    const float4 a = *(const float4*)(gt + 4 * g);
    const float area_a = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
    float v = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < A; p += gridDim.x * 256) {
      const float4 b = *(const float4*)(anchors + 4 * p);
      v = fmaxf(v, anchor_iou(a, area_a, b, (b.z - b.x + 1.f) * (b.w - b.y + 1.f)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(best + g, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    __syncthreads();
  }
}

// pass 2: per anchor the Matcher (argmax, thresholds, low-quality restore), the label 1 / 0 / -1 with the visibility and the
// between-thresholds rules in the reference's order (rpn/loss.py:60-77), and BoxCoder.encode of the matched ground truth
__global__ __launch_bounds__(256) void rpn_match_encode_kernel(const float* __restrict__ gt, const float* __restrict__ anchors,
                                                              const unsigned char* __restrict__ visible, int G, int A,
                                                              float high, float low, int allow_low_quality,
                                                              const unsigned* __restrict__ best_per_gt, float wx, float wy,
                                                              float ww, float wh, long* __restrict__ labels,
                                                              float* __restrict__ reg) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= A) return;
  const float4 b = *(const float4*)(anchors + 4 * p);
  const float area_b = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  float best = -1.f;
  int arg = 0;
  bool tied = false;
  for (int g = 0; g < G; ++g) {
    const float4 a = *(const float4*)(gt + 4 * g);
    const float area_a = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
    const float iou = anchor_iou(a, area_a, b, area_b);
    if (iou > best) {  // first maximum wins
      best = iou;
      arg = g;
    }
    tied |= allow_low_quality && __float_as_uint(iou) == best_per_gt[g];
  }
  int matched = arg;
  if (best < low) matched = -1;          // BELOW_LOW_THRESHOLD
  else if (best < high) matched = -2;    // BETWEEN_THRESHOLDS
  if (tied) matched = arg;               // an anchor that ties a ground truth's best IoU keeps its argmax
  long lab = matched >= 0 ? 1 : 0;       // (matched >= 0); BELOW -> 0
  if (!visible[p]) lab = -1;             // anchors that straddle the image border are ignored
  if (matched == -2) lab = -1;           // between the thresholds: ignored
  labels[p] = lab;
  const float4 a = *(const float4*)(gt + 4 * (matched < 0 ? 0 : matched));   // matched.clamp(min=0)
  const float ex_w = b.z - b.x + 1.f, ex_h = b.w - b.y + 1.f;
  const float ex_cx = b.x + 0.5f * ex_w, ex_cy = b.y + 0.5f * ex_h;
  const float gt_w = a.z - a.x + 1.f, gt_h = a.w - a.y + 1.f;
  const float gt_cx = a.x + 0.5f * gt_w, gt_cy = a.y + 0.5f * gt_h;
  float4 r;
  r.x = wx * (gt_cx - ex_cx) / ex_w;
  r.y = wy * (gt_cy - ex_cy) / ex_h;
  r.z = ww * logf(gt_w / ex_w);
  r.w = wh * logf(gt_h / ex_h);
  *(float4*)(reg + 4 * p) = r;
}

__global__ __launch_bounds__(256) void project_masks_kernel(const unsigned char* __restrict__ masks,
                                                           const long* __restrict__ gt_index,
                                                           const float* __restrict__ boxes, int P, int H, int W, int M,
                                                           int is_bool, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)P * M * M;
  if (i >= total) return;
  const int dx = (int)(i % M), dy = (int)((i / M) % M), p = (int)(i / ((long)M * M));
  const float4 bx = *(const float4*)(boxes + 4 * p);
  const float b0 = rintf(bx.x), b1 = rintf(bx.y), b2 = rintf(bx.z), b3 = rintf(bx.w);  // python round(): half to even
  const float xmin = fminf(fmaxf(b0, 0.f), (float)(W - 1)), ymin = fminf(fmaxf(b1, 0.f), (float)(H - 1));
  const float xmax = fmaxf(fminf(fmaxf(b2, 0.f), (float)W), xmin + 1.f);
  const float ymax = fmaxf(fminf(fmaxf(b3, 0.f), (float)H), ymin + 1.f);
  const float w = xmax - xmin, h = ymax - ymin;
  // scale = in / out as ONE correctly rounded division (area_pixel_compute_scale of F.interpolate; segmentation_mask.py:
  // 150-155): a crop whose size is a multiple of M must give integer source positions -- zero weight on the second tap --
  // which size * (1 / M) misses by an ulp (tests/test_step_golden.py: the reference's own step disagreed there)
  const float sy = fmaxf(((float)dy + 0.5f) * (h / (float)M) - 0.5f, 0.f);
  const float sx = fmaxf(((float)dx + 0.5f) * (w / (float)M) - 0.5f, 0.f);
  float y0 = floorf(sy), x0 = floorf(sx);
  const float ly = sy - y0, lx = sx - x0;
  y0 = fminf(y0, h - 1.f);
  x0 = fminf(x0, w - 1.f);
  const float y1 = fminf(y0 + 1.f, h - 1.f), x1 = fminf(x0 + 1.f, w - 1.f);
  const unsigned char* m = masks + (long)gt_index[p] * H * W;
  const long r0 = (long)(y0 + ymin) * W, r1 = (long)(y1 + ymin) * W;
  const long c0 = (long)(x0 + xmin), c1 = (long)(x1 + xmin);
  const float v00 = (float)m[r0 + c0], v01 = (float)m[r0 + c1], v10 = (float)m[r1 + c0], v11 = (float)m[r1 + c1];
  const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  out[i] = is_bool ? (v != 0.f ? 1.f : 0.f) : (float)(unsigned char)v;
}
// ---- fg / bg sampler ------------------------------------------------------------------------------------------------
constexpr int kSampThreads = 1024;

__device__ __forceinline__ unsigned sample_key(unsigned long long seed, unsigned i) {  // splitmix64 of (seed, index)
  unsigned long long z = seed + ((unsigned long long)i + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)((z ^ (z >> 31)) >> 32);
}

// class of a matched label: 1 positive (label >= 1), 0 negative (label == 0), 2 ignored (< 0)
__device__ __forceinline__ int sample_class(long lab) { return lab >= 1 ? 1 : (lab == 0 ? 0 : 2); }

// exclusive block scan of one int per thread (kSampThreads threads); returns the prefix, *total = the sum.  Wave scans by
// shuffles, the 16 wave totals through LDS: two barriers per scan (the 10-step LDS ladder it replaces had twenty).
__device__ __forceinline__ int block_exscan(int v, int* sh, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int base = 0, sum = 0;
#pragma unroll
  for (int w = 0; w < kSampThreads / 64; ++w) {
    const int t = sh[w];
    base += w < wave ? t : 0;
    sum += t;
  }
  *total = sum;
  __syncthreads();
  return base + incl - v;
}

// One workgroup samples one image.  Keys are a hash of (seed, index); a class (positives / negatives) that has more members
// than wanted keeps its k smallest keys (ties by index).  The k-th key is found by a radix select over the digits
// [31:24] [23:18] [17:9] [8:0]; CACHE: every element's class and the TOP 14 BITS of its key sit in LDS (2 bytes per element,
// filled once with coalesced label reads), so the 64-bit hash is evaluated once per element instead of in each of the seven
// passes over the labels -- on the 63 000 anchors of the RPN sampler the one-CU kernel was bound by exactly that arithmetic
// (157 us per image); only the handful of elements whose cached bits equal the threshold's are hashed again.  The selection
// is the same set in the same order as the uncached form (P above the LDS budget).
template <bool CACHE>
__global__ __launch_bounds__(kSampThreads) void sample_fg_bg_kernel(const long* __restrict__ labels, int P, int batch_size,
                                                                    int max_pos, unsigned long long seed,
                                                                    long* __restrict__ sel_idx, long* __restrict__ pos_slot,
                                                                    int* __restrict__ counts) {
  extern __shared__ unsigned short cache[];  // CACHE: (class << 14) | (key >> 18) per element
  __shared__ int sh[kSampThreads / 64];
  __shared__ unsigned hist[2][512];
  __shared__ unsigned prefix[2], want[2];  // per class: key prefix fixed so far, how many keys below it are still wanted
  const int tid = threadIdx.x;
  // contiguous slices (the scans below are in index order); long slices are a multiple of 8 elements = 16 bytes of cache
  int chunk = (P + kSampThreads - 1) / kSampThreads;
  const bool vec = CACHE && chunk >= 16;
  if (vec) chunk = (chunk + 7) & ~7;
  const int i0 = min(tid * chunk, P), i1 = min(i0 + chunk, P);
  // f(i, entry) over the thread's slice, entry = (class << 14) | (key >> 18 when cached); long cached slices: eight entries
  // per LDS read
  auto for_slice = [&](auto&& f) {
    if (CACHE && !vec) {
      for (int i = i0; i < i1; ++i) f(i, (unsigned)cache[i]);
    } else if (CACHE) {
      for (int b = i0; b < i1; b += 8) {
        const uint4 q = *(const uint4*)(cache + b);
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = b + u;
          if (i < i1) f(i, (w[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
        }
      }
    } else {
      for (int i = i0; i < i1; ++i) f(i, (unsigned)sample_class(labels[i]) << 14);
    }
  };
  // class counts
  int c1 = 0, c0 = 0;
  if (CACHE) {
    for (int base = 0; base < P; base += 8 * kSampThreads) {  // coalesced, eight loads in flight per thread
      long v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * kSampThreads + tid;
        v[u] = i < P ? labels[i] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + u * kSampThreads + tid;
        if (i < P) {
          const int c = sample_class(v[u]);
          cache[i] = (unsigned short)(c << 14);
          c1 += c == 1;
          c0 += c == 0;
        }
      }
    }
  } else {
    for_slice([&](int, unsigned e) {
      c1 += (e >> 14) == 1;
      c0 += (e >> 14) == 0;
    });
  }
  int npos, nneg;
  block_exscan(c1, sh, &npos);
  block_exscan(c0, sh, &nneg);
  const int k_pos = min(npos, max_pos);
  const int k_neg = min(nneg, batch_size - k_pos);
  const int kk[2] = {k_neg, k_pos};
  const bool need[2] = {k_neg < nneg, k_pos < npos};  // the class needs a threshold (else everything of it is taken)
  if (CACHE) {
    for (int i = tid; i < P; i += kSampThreads) {  // a thread revisits the entries it wrote itself
      const int c = cache[i] >> 14;
      if (c != 2 && need[c]) cache[i] = (unsigned short)((c << 14) | (sample_key(seed, (unsigned)i) >> 18));
    }
  }
  if (tid < 2) {
    prefix[tid] = 0;
    want[tid] = (unsigned)kk[tid];
  }
  __syncthreads();
  // radix select, most significant digit first: afterwards every key below thr[c] is selected and `want[c]` of the keys
  // equal to it (lowest indices first)
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = pass == 0 ? 24 : (pass == 1 ? 18 : (pass == 2 ? 9 : 0));
    const int bits = pass == 0 ? 8 : (pass == 1 ? 6 : 9);
    const unsigned dmask = (1u << bits) - 1u;
    for (int b = tid; b < 1024; b += kSampThreads) hist[b >> 9][b & 511] = 0;
    __syncthreads();
    const unsigned pf0 = prefix[0], pf1 = prefix[1];
    for_slice([&](int i, unsigned e) {
      const int c = (int)(e >> 14);
      if (c == 2 || !need[c]) return;
      const unsigned pf = c ? pf1 : pf0;
      unsigned key;
      if (CACHE) {
        const unsigned top = e & 0x3fffu;                            // key >> 18
        if (pass == 1 && (top >> 6) != (pf >> 24)) return;
        if (pass >= 2 && top != (pf >> 18)) return;
        key = pass < 2 ? top << 18 : sample_key(seed, (unsigned)i);
      } else {
        key = sample_key(seed, (unsigned)i);
      }
      if (pass == 0 || (key >> (shift + bits)) == (pf >> (shift + bits))) atomicAdd(&hist[c][(key >> shift) & dmask], 1u);
    });
    __syncthreads();
    // first digit whose bucket crosses the number of keys still wanted: wave c searches class c's histogram -- eight bins
    // per lane, a wave scan of the lane sums, the crossing lane walks its own bins (one thread walking 512 LDS words cost
    // ~25 us per pass)
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < 2 && need[wave]) {  // wave-uniform
      const int nb = (int)dmask + 1, per = (nb + 63) >> 6;
      unsigned loc[8], sum = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int b = lane * per + j;
        loc[j] = (j < per && b < nb) ? hist[wave][b] : 0u;
        sum += loc[j];
      }
      unsigned incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      const unsigned w = want[wave];
      const unsigned long long crossing = __ballot(incl > w);
      const int first = crossing ? __ffsll((long long)crossing) - 1 : 63;
      if (lane == first) {
        unsigned acc = incl - sum;
        int d = lane * per;
        if (crossing) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (j < per && acc + loc[j] <= w) {
              acc += loc[j];
              d = lane * per + j + 1;
            } else {
              break;
            }
          }
        } else {  // never for a class that needs a threshold; kept as the scalar walk's end state
          acc = incl;
          d = nb;
        }
        d = min(d, (int)dmask);
        prefix[wave] |= (unsigned)d << shift;
        want[wave] = w - acc;
      }
    }
    __syncthreads();
  }
  const unsigned thr[2] = {prefix[0], prefix[1]};
  const unsigned ties_wanted[2] = {want[0], want[1]};
  // -1 below the threshold, 0 equal, +1 above (the cached bits decide all but the few elements that share the threshold's)
  auto versus_threshold = [&](int i, unsigned e, int c) -> int {
    if (CACHE) {
      const unsigned top = e & 0x3fffu, ttop = thr[c] >> 18;
      if (top != ttop) return top < ttop ? -1 : 1;
    }
    const unsigned key = sample_key(seed, (unsigned)i);
    return key < thr[c] ? -1 : (key == thr[c] ? 0 : 1);
  };
  // ties (key == thr[c]) are granted in index order: exclusive count of the thread's preceding ties, per class
  int t0 = 0, t1 = 0;
  for_slice([&](int i, unsigned e) {
    const int c = (int)(e >> 14);
    if (c == 2 || !need[c]) return;
    if (versus_threshold(i, e, c) == 0) (c ? t1 : t0) += 1;
  });
  int tot;
  int tie_before[2];
  tie_before[0] = block_exscan(t0, sh, &tot);
  tie_before[1] = block_exscan(t1, sh, &tot);
  // selection flags -> ascending compaction (and the positives' slots inside the compacted list)
  int n_sel = 0, n_ps = 0;
  int seen[2] = {tie_before[0], tie_before[1]};
  auto taken = [&](int i, unsigned e, int c) -> bool {
    if (!need[c]) return true;
    const int v = versus_threshold(i, e, c);
    return v < 0 || (v == 0 && (unsigned)(seen[c]++) < ties_wanted[c]);
  };
  for_slice([&](int i, unsigned e) {
    const int c = (int)(e >> 14);
    if (c == 2) return;
    const bool take = taken(i, e, c);
    n_sel += take;
    n_ps += take && c == 1;
  });
  int total_sel, total_pos;
  int at = block_exscan(n_sel, sh, &total_sel);
  int pat = block_exscan(n_ps, sh, &total_pos);
  seen[0] = tie_before[0];
  seen[1] = tie_before[1];
  for_slice([&](int i, unsigned e) {
    const int c = (int)(e >> 14);
    if (c == 2) return;
    if (taken(i, e, c)) {
      if (c == 1) pos_slot[pat++] = at;
      sel_idx[at++] = i;
    }
  });
  for (int j = total_sel + tid; j < batch_size; j += kSampThreads) sel_idx[j] = 0;
  for (int j = total_pos + tid; j < batch_size; j += kSampThreads) pos_slot[j] = 0;
  if (tid == 0) {
    counts[0] = total_sel;
    counts[1] = total_pos;
  }
}

// ---- mask targets from (probability map, box) pairs ---------------------------------------------------------------------
// Pixel (Y, X) of the binary image mask Masker would paste for ground truth g (mask_head/inference.py:100-160 with
// padding 1): the M x M map, zero-padded to (M+2)^2, is resized bilinearly (align_corners=False) to the integer box
// obtained by expanding the box by (M+2)/M about its centre and truncating, thresholded, and pasted clipped to the image.
__device__ __forceinline__ float pasted_pixel(const float* __restrict__ prob, int M, int4 bx, int bw, int bh, float thr,
                                              int im_h, int im_w, int Y, int X) {
  if (Y < 0 || X < 0 || Y >= im_h || X >= im_w || Y < bx.y || Y > bx.w || X < bx.x || X > bx.z) return 0.f;
  const int S = M + 2;
  // upsample_bilinear2d, align_corners=False: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out
  const float scale_y = (float)S / (float)bh, scale_x = (float)S / (float)bw;
  const float sy = fmaxf(scale_y * ((float)(Y - bx.y) + 0.5f) - 0.5f, 0.f);
  const float sx = fmaxf(scale_x * ((float)(X - bx.x) + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < S - 1), x1 = x0 + (x0 < S - 1);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  auto at = [&](int y, int x) {  // the padded map: a zero border of one pixel
    return (y >= 1 && y <= M && x >= 1 && x <= M) ? prob[(y - 1) * M + (x - 1)] : 0.f;
  };
  const float v = (1.f - ly) * ((1.f - lx) * at(y0, x0) + lx * at(y0, x1)) + ly * ((1.f - lx) * at(y1, x0) + lx * at(y1, x1));
  return v > thr ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void project_pasted_masks_kernel(const float* __restrict__ probs,
                                                                  const float* __restrict__ gt_boxes,
                                                                  const long* __restrict__ gt_index,
                                                                  const float* __restrict__ boxes, int P, int H, int W,
                                                                  int Mp, int M, float thr, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)P * M * M;
  if (i >= total) return;
  const int dx = (int)(i % M), dy = (int)((i / M) % M), p = (int)(i / ((long)M * M));
  // the crop window and the resize taps: project_masks_kernel, expression by expression
  const float4 bb = *(const float4*)(boxes + 4 * p);
  const float b0 = rintf(bb.x), b1 = rintf(bb.y), b2 = rintf(bb.z), b3 = rintf(bb.w);
  const float xmin = fminf(fmaxf(b0, 0.f), (float)(W - 1)), ymin = fminf(fmaxf(b1, 0.f), (float)(H - 1));
  const float xmax = fmaxf(fminf(fmaxf(b2, 0.f), (float)W), xmin + 1.f);
  const float ymax = fmaxf(fminf(fmaxf(b3, 0.f), (float)H), ymin + 1.f);
  const float w = xmax - xmin, h = ymax - ymin;
  const float sy = fmaxf(((float)dy + 0.5f) * (h / (float)M) - 0.5f, 0.f);
  const float sx = fmaxf(((float)dx + 0.5f) * (w / (float)M) - 0.5f, 0.f);
  float y0 = floorf(sy), x0 = floorf(sx);
  const float ly = sy - y0, lx = sx - x0;
  y0 = fminf(y0, h - 1.f);
  x0 = fminf(x0, w - 1.f);
  const float y1 = fminf(y0 + 1.f, h - 1.f), x1 = fminf(x0 + 1.f, w - 1.f);
  // the ground truth's pasted mask: integer box of the expanded pseudo box (Masker: scale (M+2)/M, truncation)
  const long g = gt_index[p];
  const float4 gb = *(const float4*)(gt_boxes + 4 * g);
  const float scale = (float)(Mp + 2) / (float)Mp;
  const float w_half = (gb.z - gb.x) * 0.5f * scale, h_half = (gb.w - gb.y) * 0.5f * scale;
  const float x_c = (gb.z + gb.x) * 0.5f, y_c = (gb.w + gb.y) * 0.5f;
  const int4 bx = make_int4((int)(x_c - w_half), (int)(y_c - h_half), (int)(x_c + w_half), (int)(y_c + h_half));
  const int bw = max(bx.z - bx.x + 1, 1), bh = max(bx.w - bx.y + 1, 1);
  const float* pr = probs + g * Mp * Mp;
  const int Y0 = (int)(y0 + ymin), Y1 = (int)(y1 + ymin), X0 = (int)(x0 + xmin), X1 = (int)(x1 + xmin);
  const float v00 = pasted_pixel(pr, Mp, bx, bw, bh, thr, H, W, Y0, X0), v01 = pasted_pixel(pr, Mp, bx, bw, bh, thr, H, W, Y0, X1);
  const float v10 = pasted_pixel(pr, Mp, bx, bw, bh, thr, H, W, Y1, X0), v11 = pasted_pixel(pr, Mp, bx, bw, bh, thr, H, W, Y1, X1);
  const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  out[i] = v != 0.f ? 1.f : 0.f;
}
// rows `index[0..n)` of up to two [P, 4] f32 arrays and two [P] int64 arrays in one launch: the sampled proposals'
// boxes / regression targets / labels / matched ground truths (box_head/loss.py:112-121: proposals_per_image[img_sampled_inds])
__global__ __launch_bounds__(256) void gather_rows_kernel(const long long* __restrict__ index, int n,
                                                         const float4* __restrict__ fa, float4* __restrict__ fa_out,
                                                         const float4* __restrict__ fb, float4* __restrict__ fb_out,
                                                         const long long* __restrict__ ia, long long* __restrict__ ia_out,
                                                         const long long* __restrict__ ib, long long* __restrict__ ib_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long r = index[i];
  if (fa) fa_out[i] = fa[r];
  if (fb) fb_out[i] = fb[r];
  if (ia) ia_out[i] = ia[r];
  if (ib) ib_out[i] = ib[r];
}

}  // namespace

extern "C" int ovis_sample_fg_bg(const int64_t* labels, int num, int batch_size, int max_positives, uint64_t seed,
                                 int64_t* selected, int64_t* positive_slots, int32_t* counts, void* stream) {
  if (num < 0 || batch_size <= 0 || max_positives < 0 || max_positives > batch_size) return OVIS_EINVAL;
  if (!selected || !positive_slots || !counts || (num > 0 && !labels)) return OVIS_EINVAL;
  constexpr int kCacheBytes = 150 * 1024;  // 2 bytes per element next to the kernel's ~4 KB of static LDS
  if (2L * num + 16 <= kCacheBytes) {
    static bool attr_set = false;
    if (!attr_set) {
      OVIS_HIP_TRY(hipFuncSetAttribute((const void*)sample_fg_bg_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kCacheBytes));
      attr_set = true;
    }
    const int lds = ((2 * num + 16 + 255) / 256) * 256;  // a slice reads whole 16-byte groups
    hipLaunchKernelGGL(sample_fg_bg_kernel<true>, dim3(1), dim3(kSampThreads), lds, (hipStream_t)stream, (const long*)labels, num,
                       batch_size, max_positives, (unsigned long long)seed, (long*)selected, (long*)positive_slots, counts);
  } else {
    hipLaunchKernelGGL(sample_fg_bg_kernel<false>, dim3(1), dim3(kSampThreads), 0, (hipStream_t)stream, (const long*)labels, num,
                       batch_size, max_positives, (unsigned long long)seed, (long*)selected, (long*)positive_slots, counts);
  }
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_project_pasted_masks_f32(const float* mask_probs, const float* gt_boxes, const int64_t* gt_index,
                                             const float* boxes, int num, int image_height, int image_width,
                                             int prob_resolution, int resolution, float threshold, float* out,
                                             void* stream) {
  if (num < 0 || image_height <= 0 || image_width <= 0 || prob_resolution <= 0 || resolution <= 0) return OVIS_EINVAL;
  if (num == 0) return OVIS_OK;
  if (!mask_probs || !gt_boxes || !gt_index || !boxes || !out) return OVIS_EINVAL;
  if (((uintptr_t)boxes & 15) || ((uintptr_t)gt_boxes & 15)) return OVIS_ERANGE;
  const long total = (long)num * resolution * resolution;
  hipLaunchKernelGGL(project_pasted_masks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     mask_probs, gt_boxes, (const long*)gt_index, boxes, num, image_height, image_width, prob_resolution,
                     resolution, threshold, out);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_match_encode_f32(const float* gt_boxes, const int64_t* gt_labels, const float* proposals,
                                     int num_gt, int num_proposals, float high_threshold, float low_threshold,
                                     int between_keeps_label, float wx, float wy, float ww, float wh,
                                     int64_t* matched_idx, int64_t* labels, float* regression_targets, void* stream) {
  if (num_gt <= 0 || num_proposals < 0) return OVIS_EINVAL;
  if (num_proposals == 0) return OVIS_OK;
  if (!gt_boxes || !gt_labels || !proposals || !matched_idx || !labels) return OVIS_EINVAL;
  if (((uintptr_t)gt_boxes & 15) || ((uintptr_t)proposals & 15) || ((uintptr_t)regression_targets & 15)) return OVIS_ERANGE;
  hipLaunchKernelGGL(match_encode_kernel, dim3((num_proposals + 255) / 256), dim3(256), 0, (hipStream_t)stream, gt_boxes,
                     (const long*)gt_labels, proposals, num_gt, num_proposals, high_threshold, low_threshold,
                     between_keeps_label, wx, wy, ww, wh, (long*)matched_idx, (long*)labels, regression_targets);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_rpn_match_encode_f32(const float* gt_boxes, const float* anchors, const uint8_t* visibility, int num_gt,
                                         int num_anchors, float high_threshold, float low_threshold,
                                         int allow_low_quality_matches, float wx, float wy, float ww, float wh,
                                         uint32_t* best_per_gt_scratch, int64_t* labels, float* regression_targets,
                                         void* stream) {
  if (num_gt <= 0 || num_anchors < 0) return OVIS_EINVAL;
  if (num_anchors == 0) return OVIS_OK;
  if (!gt_boxes || !anchors || !visibility || !best_per_gt_scratch || !labels || !regression_targets) return OVIS_EINVAL;
  if (((uintptr_t)gt_boxes & 15) || ((uintptr_t)anchors & 15) || ((uintptr_t)regression_targets & 15)) return OVIS_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((num_anchors + 255) / 256);
  if (allow_low_quality_matches) {
    OVIS_HIP_TRY(hipMemsetAsync(best_per_gt_scratch, 0, sizeof(uint32_t) * (size_t)num_gt, s));
    hipLaunchKernelGGL(rpn_best_per_gt_kernel, dim3(blocks < 64u ? blocks : 64u), dim3(256), 0, s, gt_boxes, anchors, num_gt,
                       num_anchors, best_per_gt_scratch);
    OVIS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(rpn_match_encode_kernel, dim3(blocks), dim3(256), 0, s, gt_boxes, anchors, visibility, num_gt,
                     num_anchors, high_threshold, low_threshold, allow_low_quality_matches, best_per_gt_scratch, wx, wy, ww, wh,
                     (long*)labels, regression_targets);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_project_masks_f32(const uint8_t* masks, const int64_t* gt_index, const float* boxes, int num,
                                      int height, int width, int resolution, int masks_are_bool, float* out,
                                      void* stream) {
  if (num < 0 || height <= 0 || width <= 0 || resolution <= 0) return OVIS_EINVAL;
  if (num == 0) return OVIS_OK;
  if (!masks || !gt_index || !boxes || !out) return OVIS_EINVAL;
  if ((uintptr_t)boxes & 15) return OVIS_ERANGE;
  const long total = (long)num * resolution * resolution;
  hipLaunchKernelGGL(project_masks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, masks,
                     (const long*)gt_index, boxes, num, height, width, resolution, masks_are_bool, out);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_gather_rows(const int64_t* index, int num, const float* boxes_a, float* boxes_a_out,
                                const float* boxes_b, float* boxes_b_out, const int64_t* ints_a, int64_t* ints_a_out,
                                const int64_t* ints_b, int64_t* ints_b_out, void* stream) {
  if (num < 0) return OVIS_EINVAL;
  if (num == 0) return OVIS_OK;
  if (!index || (boxes_a && !boxes_a_out) || (boxes_b && !boxes_b_out) || (ints_a && !ints_a_out) || (ints_b && !ints_b_out))
    return OVIS_EINVAL;
  if ((((uintptr_t)boxes_a | (uintptr_t)boxes_a_out | (uintptr_t)boxes_b | (uintptr_t)boxes_b_out) & 15) != 0) return OVIS_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ovis_ceil_div(num, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)index, num, (const float4*)boxes_a, (float4*)boxes_a_out, (const float4*)boxes_b,
                     (float4*)boxes_b_out, (const long long*)ints_a, (long long*)ints_a_out, (const long long*)ints_b,
                     (long long*)ints_b_out);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
