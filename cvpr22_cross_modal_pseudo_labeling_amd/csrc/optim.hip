// SGD with momentum and weight decay over MANY tensors in one launch (gfx950).
//
// Reference: maskrcnn_benchmark/solver/build.py:8-37 builds one torch.optim.SGD parameter group per parameter (bias: lr x 2,
// no decay); the update of every parameter is
//     d = g + wd * p;   buf = momentum * buf + d;   p = p - lr * buf            (torch.optim.SGD, dampening 0, no nesterov)
// evaluated by `engine/solver.py::GroupFusedSGD` as six multi-tensor passes (15 tensor reads / writes per element: 2.1 GB
// and 0.44 ms per step for the teacher's 35 M parameters).  Here: one pass -- p, g, buf read, buf, p written (20 bytes per
// element) -- over all tensors that share (lr, wd, momentum), with the SAME fp32 operation sequence and roundings as the
// multi-tensor form (no contraction): bit-identical parameters.
//
// A launch = one group of tensors with common scalars.  `items[i]` = (p, g, buf, n); block b updates elements
// [chunk * kChunk, +kChunk) of item blocks[b].x, chunk = blocks[b].y.  Gradients are views into the reducer's flat buckets
// at arbitrary 4-byte offsets, so the accesses are 4 bytes per lane (a wave still covers 256 contiguous bytes).
#include "ovis_common.h"

namespace {

struct SgdItem {
  float* p;
  const float* g;
  float* buf;  // momentum buffer (ignored when momentum == 0)
  long n;
};

constexpr int kSgdThreads = 256, kSgdChunk = 4096;

__global__ __launch_bounds__(kSgdThreads) void sgd_momentum_multi_kernel(const SgdItem* __restrict__ items,
                                                                        const int2* __restrict__ blocks, float neg_lr,
                                                                        float wd, float momentum, int use_wd) {
  const int2 bc = blocks[blockIdx.x];
  const SgdItem it = items[bc.x];
  const long base = (long)bc.y * kSgdChunk;
  const long end = min(base + kSgdChunk, it.n);
#pragma unroll 4
  for (long i = base + threadIdx.x; i < end; i += kSgdThreads) {
    const float p = it.p[i];
    float d = it.g[i];
    if (use_wd) d = d + p * wd;                 // _foreach_add(grads, _foreach_mul(params, wds))
    float upd = d;
    if (momentum != 0.f) {
      upd = it.buf[i] * momentum + d;           // _foreach_mul_(bufs, momentum); _foreach_add_(bufs, d)
      it.buf[i] = upd;
    }
    it.p[i] = p + upd * neg_lr;                 // _foreach_add_(params, _foreach_mul(upd, -lr))
  }
}

}  // namespace

extern "C" int ovis_sgd_momentum_multi_f32(const void* items, const void* blocks, int num_blocks, float lr,
                                           float weight_decay, float momentum, int apply_weight_decay, void* stream) {
  if (num_blocks < 0) return OVIS_EINVAL;
  if (num_blocks == 0) return OVIS_OK;
  if (!items || !blocks) return OVIS_EINVAL;
  hipLaunchKernelGGL(sgd_momentum_multi_kernel, dim3((unsigned)num_blocks), dim3(kSgdThreads), 0, (hipStream_t)stream,
                     (const SgdItem*)items, (const int2*)blocks, -lr, weight_decay, momentum, apply_weight_decay);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_sgd_chunk_elements(void) { return kSgdChunk; }
