// Sigmoid focal loss forward / backward for gfx950 (MI355X), fp32.
//
// Formula and numerically-stable log(1-p) form follow
//   maskrcnn_benchmark/csrc/cuda/SigmoidFocalLoss_cuda.cu:21-58 (forward), :62-101 (backward).
// Pure streaming op (8 B/element forward, 12 B/element backward): each lane handles four
// consecutive logits of one row through 16-byte loads/stores when num_classes % 4 == 0
// (RetinaNet's C = 80), otherwise a scalar grid-stride path is used.
#include <float.h>

#include "ovis_common.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float focal_fwd_elem(float x, int t, int d, float gamma, float alpha) {
  const float c1 = (t == d + 1) ? 1.f : 0.f;
  const float c2 = (t >= 0 && t != d + 1) ? 1.f : 0.f;
  const float zn = 1.f - alpha, zp = alpha;
  const float p = 1.f / (1.f + expf(-x));
  const float term1 = powf(1.f - p, gamma) * logf(fmaxf(p, FLT_MIN));
  const float pos = x >= 0.f ? 1.f : 0.f;
  const float term2 = powf(p, gamma) * (-1.f * x * pos - logf(1.f + expf(x - 2.f * x * pos)));
  float loss = 0.f;
  loss += -c1 * term1 * zp;
  loss += -c2 * term2 * zn;
  return loss;
}

__device__ __forceinline__ float focal_bwd_elem(float x, int t, int d, float gamma, float alpha,
                                                float up) {
  const float c1 = (t == d + 1) ? 1.f : 0.f;
  const float c2 = (t >= 0 && t != d + 1) ? 1.f : 0.f;
  const float zn = 1.f - alpha, zp = alpha;
  const float p = 1.f / (1.f + expf(-x));
  const float term1 = powf(1.f - p, gamma) * (1.f - p - (p * gamma * logf(fmaxf(p, FLT_MIN))));
  const float pos = x >= 0.f ? 1.f : 0.f;
  const float term2 =
      powf(p, gamma) *
      ((-1.f * x * pos - logf(1.f + expf(x - 2.f * x * pos))) * (1.f - p) * gamma - p);
  float g = 0.f;
  g += -c1 * term1 * zp;
  g += -c2 * term2 * zn;
  return g * up;
}

template <bool BWD>
__global__ __launch_bounds__(kThreads) void focal_vec4_kernel(
    const float4* __restrict__ logits, const int* __restrict__ targets,
    const float4* __restrict__ up, float4* __restrict__ out, long nvec, int cvec, float gamma,
    float alpha) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < nvec;
       i += (long)gridDim.x * kThreads) {
    const long n = i / cvec;
    const int d0 = (int)(i - n * cvec) * 4;
    const int t = targets[n];
    const float4 x = logits[i];
    float4 r;
    if (BWD) {
      const float4 u = up[i];
      r.x = focal_bwd_elem(x.x, t, d0 + 0, gamma, alpha, u.x);
      r.y = focal_bwd_elem(x.y, t, d0 + 1, gamma, alpha, u.y);
      r.z = focal_bwd_elem(x.z, t, d0 + 2, gamma, alpha, u.z);
      r.w = focal_bwd_elem(x.w, t, d0 + 3, gamma, alpha, u.w);
    } else {
      r.x = focal_fwd_elem(x.x, t, d0 + 0, gamma, alpha);
      r.y = focal_fwd_elem(x.y, t, d0 + 1, gamma, alpha);
      r.z = focal_fwd_elem(x.z, t, d0 + 2, gamma, alpha);
      r.w = focal_fwd_elem(x.w, t, d0 + 3, gamma, alpha);
    }
    out[i] = r;
  }
}

template <bool BWD>
__global__ __launch_bounds__(kThreads) void focal_scalar_kernel(
    const float* __restrict__ logits, const int* __restrict__ targets,
    const float* __restrict__ up, float* __restrict__ out, long total, int C, float gamma,
    float alpha) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long)gridDim.x * kThreads) {
    const long n = i / C;
    const int d = (int)(i - n * C);
    const int t = targets[n];
    out[i] = BWD ? focal_bwd_elem(logits[i], t, d, gamma, alpha, up[i])
                 : focal_fwd_elem(logits[i], t, d, gamma, alpha);
  }
}

template <bool BWD>
int launch(const float* logits, const int* targets, const float* up, float* out, int num, int C,
           float gamma, float alpha, hipStream_t s) {
  if (num < 0 || C < 0) return OVIS_EINVAL;
  const long total = (long)num * C;
  if (total == 0) return OVIS_OK;
  if (!logits || !targets || !out || (BWD && !up)) return OVIS_EINVAL;
  const bool aligned = (((uintptr_t)logits | (uintptr_t)out | (uintptr_t)up) & 15) == 0;
  if (C % 4 == 0 && aligned) {
    const long nvec = total / 4;
    long blocks = (nvec + kThreads - 1) / kThreads;
    if (blocks > OVIS_NUM_CU * 16) blocks = OVIS_NUM_CU * 16;
    hipLaunchKernelGGL(focal_vec4_kernel<BWD>, dim3((unsigned)blocks), dim3(kThreads), 0, s,
                       (const float4*)logits, targets, (const float4*)up, (float4*)out, nvec,
                       C / 4, gamma, alpha);
  } else {
    long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > OVIS_NUM_CU * 16) blocks = OVIS_NUM_CU * 16;
    hipLaunchKernelGGL(focal_scalar_kernel<BWD>, dim3((unsigned)blocks), dim3(kThreads), 0, s,
                       logits, targets, up, out, total, C, gamma, alpha);
  }
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

}  // namespace

extern "C" int ovis_sigmoid_focal_loss_forward_f32(const float* logits, const int32_t* targets,
                                                   float* losses, int num, int num_classes,
                                                   float gamma, float alpha, void* stream) {
  return launch<false>(logits, targets, nullptr, losses, num, num_classes, gamma, alpha,
                       (hipStream_t)stream);
}

extern "C" int ovis_sigmoid_focal_loss_backward_f32(const float* logits, const int32_t* targets,
                                                    const float* d_losses, float* d_logits,
                                                    int num, int num_classes, float gamma,
                                                    float alpha, void* stream) {
  return launch<true>(logits, targets, d_losses, d_logits, num, num_classes, gamma, alpha,
                      (hipStream_t)stream);
}
