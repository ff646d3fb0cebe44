// The backward of an identity bottleneck inside a pair-only chain, behind ONE native call, for gfx950 (MI355X).
// maskrcnn_benchmark/modeling/backbone/resnet.py:290-342 (Bottleneck.forward; autograd derives its backward): with the
// gradient w.r.t. the block's output arriving gated and split from the block above (g3), the block's backward is nine
// launches in a fixed order --
//     dW3 = g3^T o2            g2 = (g3 W3) * (o1' > 0)         [1x1]
//     dW2 = g2^T o1 (3x3)      g1 = (g2 (*) W2) * (o1 > 0)      [3x3, flipped taps]
//     dW1 = g1^T x             gx = (g1 W1 + g3) * (x > 0)      [1x1 + shortcut, handed to the block below]
// -- each of which already has its entry point in this library.  Issued one by one from the host language they cost ~15 us
// of host time apiece (150 us per block against 225 us of kernels on one stream, profiles/r6_ab_trunk_dw_beside.txt), which
// is what kept the three weight gradients from running BESIDE the data-gradient chain: on two streams the kernels' critical
// path is ~110 us and the host became the pace.  Here the host pays one call; the weight gradients go to `side_stream`
// (NULL: everything on `stream`), ordered against the chain by events, and `stream` waits for them before the call returns
// its place in the queue -- the caller needs no cross-stream bookkeeping.  No new kernel: the file calls the library's own
// entry points (split_gemm.hip) with sub-allocated workspaces.
#include "ovis_common.h"

namespace {

inline size_t bb_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct BbPlan {
  size_t chain_bytes;   // workspace of the data-gradient GEMMs (they run one after the other on `stream`)
  size_t slab_bytes;    // slabs of the weight-gradient products (one after the other on the side stream)
  int s1, s2, s3;       // K slices of dW1, dW2, dW3
};

BbPlan bb_plan(long m, int cin, int mid, int kh, int kw, int width, int config) {
  BbPlan p;
  const size_t a = ovis_split_gemm_pair_workspace_bytes_ex(m, mid, cin, 0, 1, 1, 0, config);        // g2: N = mid, K = cin
  const size_t b = ovis_split_gemm_pair_workspace_bytes_ex(m, mid, mid, 0, kh, kw, width, config);  // g1: 3x3
  p.chain_bytes = bb_align(a > b ? a : b);
  p.s3 = ovis_split_gemm_tn_slices(m, cin, mid, 1);        // dW3 [cin, mid]
  p.s2 = ovis_split_gemm_tn_slices(m, mid, mid, kh * kw);  // dW2 [mid, mid, kh, kw]
  p.s1 = ovis_split_gemm_tn_slices(m, mid, cin, 1);        // dW1 [mid, cin]
  const size_t w3 = (size_t)p.s3 * cin * mid * 4, w2 = (size_t)p.s2 * mid * mid * kh * kw * 4, w1 = (size_t)p.s1 * mid * cin * 4;
  size_t mx = w3 > w2 ? w3 : w2;
  if (w1 > mx) mx = w1;
  p.slab_bytes = bb_align(mx);
  return p;
}

hipEvent_t* bb_events() {
  static hipEvent_t ev[4];
  static bool made = false;
  if (!made) {
    for (int i = 0; i < 4; ++i)
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    made = true;
  }
  return ev;
}

}  // namespace

extern "C" size_t ovis_bottleneck_identity_backward_workspace_bytes(long m, int channels, int mid_channels, int taps_h,
                                                                    int taps_w, int width, int config) {
  if (m <= 0 || channels <= 0 || mid_channels <= 0 || taps_h <= 0 || taps_w <= 0) return 0;
  const BbPlan p = bb_plan(m, channels, mid_channels, taps_h, taps_w, width, config);
  return p.chain_bytes + p.slab_bytes;
}

extern "C" int ovis_bottleneck_identity_backward(
    const void* g3_pair, long g3_row_bytes, const void* x_pair, long x_row_bytes, const void* o1_pair, long o1_row_bytes,
    const void* o2_pair, long o2_row_bytes, const void* t1_pair, long t1_row_bytes, const void* t2_pair, long t2_row_bytes,
    const void* t3_pair, long t3_row_bytes, const float* scale1, const float* scale2, const float* scale3, long m, int channels,
    int mid_channels, int taps_h, int taps_w, int height, int width, void* g2_pair, void* g1_pair, void* gx_pair, float* dw1,
    float* dw2, float* dw3, void* workspace, size_t workspace_bytes, int config, void* stream, void* side_stream) {
  if (m <= 0 || channels <= 0 || mid_channels <= 0 || taps_h <= 0 || taps_w <= 0) return OVIS_EINVAL;
  if (!g3_pair || !x_pair || !o1_pair || !o2_pair || !t1_pair || !t2_pair || !t3_pair || !g2_pair || !g1_pair || !gx_pair ||
      !dw1 || !dw2 || !dw3 || !workspace)
    return OVIS_EINVAL;
  if (channels % 128 || mid_channels % 128 || !(taps_h & 1) || !(taps_w & 1)) return OVIS_ERANGE;
  const BbPlan p = bb_plan(m, channels, mid_channels, taps_h, taps_w, width, config);
  if (workspace_bytes < p.chain_bytes + p.slab_bytes) return OVIS_ENOSPC;
  hipStream_t s = (hipStream_t)stream;
  hipStream_t side = side_stream ? (hipStream_t)side_stream : s;
  const bool two = side != s;
  hipEvent_t* ev = two ? bb_events() : nullptr;
  if (two && !ev) return OVIS_EINVAL;
  char* chain_ws = (char*)workspace;
  float* slabs = (float*)((char*)workspace + p.chain_bytes);
  const int cin = channels, mid = mid_channels, taps = taps_h * taps_w;
  int rc;
#define BB_TRY(call)            \
  do {                          \
    rc = (call);                \
    if (rc != OVIS_OK) return rc; \
  } while (0)
#define BB_AFTER_MAIN(i)                                              \
  if (two) {                                                          \
    OVIS_HIP_TRY(hipEventRecord(ev[i], s));                           \
    OVIS_HIP_TRY(hipStreamWaitEvent(side, ev[i], 0));                 \
  }
  // dW3 = g3^T o2 beside g2 = (g3 W3) gated by relu(o2)
  BB_AFTER_MAIN(0)
  BB_TRY(ovis_split_gemm_pair_tn(g3_pair, g3_row_bytes, o2_pair, o2_row_bytes, slabs, p.s3, m, cin, mid, 1, 1, 0, 0, side));
  BB_TRY(ovis_slab_reduce_f32(slabs, scale3, dw3, p.s3, cin, mid, 1, side));
  BB_TRY(ovis_split_gemm_pair_gated_ws(g3_pair, g3_row_bytes, t3_pair, t3_row_bytes, nullptr, mid, g2_pair, 4L * mid, o2_pair,
                                       o2_row_bytes, m, mid, cin, 1, 1, 0, 0, 0, chain_ws, p.chain_bytes, config, s));
  // dW2 = g2^T o1 (taps) beside g1 = (g2 (*) W2, flipped) gated by relu(o1)
  BB_AFTER_MAIN(1)
  BB_TRY(ovis_split_gemm_pair_tn(g2_pair, 4L * mid, o1_pair, o1_row_bytes, slabs, p.s2, m, mid, mid, taps_h, taps_w, height,
                                 width, side));
  BB_TRY(ovis_slab_reduce_f32(slabs, scale2, dw2, p.s2, mid, mid, taps, side));
  BB_TRY(ovis_split_gemm_pair_gated_ws(g2_pair, 4L * mid, t2_pair, t2_row_bytes, nullptr, mid, g1_pair, 4L * mid, o1_pair,
                                       o1_row_bytes, m, mid, mid, taps_h, taps_w, height, width, 1, chain_ws, p.chain_bytes,
                                       config, s));
  // dW1 = g1^T x beside gx = (g1 W1 + g3) gated by relu(x): the block below's g3
  BB_AFTER_MAIN(2)
  BB_TRY(ovis_split_gemm_pair_tn(g1_pair, 4L * mid, x_pair, x_row_bytes, slabs, p.s1, m, mid, cin, 1, 1, 0, 0, side));
  BB_TRY(ovis_slab_reduce_f32(slabs, scale1, dw1, p.s1, mid, cin, 1, side));
  BB_TRY(ovis_split_gemm_pair_rp_gated(g1_pair, 4L * mid, t1_pair, t1_row_bytes, nullptr, cin, gx_pair, 4L * cin, g3_pair,
                                       g3_row_bytes, x_pair, x_row_bytes, m, cin, mid, config, s));
  if (two) {  // the weight gradients are complete wherever `stream` goes on from here
    OVIS_HIP_TRY(hipEventRecord(ev[3], side));
    OVIS_HIP_TRY(hipStreamWaitEvent(s, ev[3], 0));
  }
#undef BB_TRY
#undef BB_AFTER_MAIN
  return OVIS_OK;
}
