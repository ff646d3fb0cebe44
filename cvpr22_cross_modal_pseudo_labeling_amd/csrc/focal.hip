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

// Everything an element needs comes from ONE exponential, one reciprocal and one logarithm (hardware v_exp_f32 /
// v_rcp_f32 / v_log_f32, ~1 ulp): e = exp(-|x|), r = 1 / (1 + e), L = log(1 + e) give
//   p = sigmoid(x) = x >= 0 ? r : e r,   1 - p = x >= 0 ? e r : r,   log p = min(x, 0) - L,
//   log(1 - p) = -max(x, 0) - L  (the reference's stable form, SigmoidFocalLoss_cuda.cu:44-46),
// and only the term of the element's own case (positive class / negative class / ignored row) is evaluated.  The
// reference evaluates expf twice, powf twice and logf twice per element for both terms (~6 software transcendental
// sequences): this op was VALU-bound at 9-13 % of the HBM rate.
struct FocalParts {
  float p, q, logp, log1mp;
};

__device__ __forceinline__ FocalParts focal_parts(float x) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896341f);
  const float r = __builtin_amdgcn_rcpf(1.f + e);
  const float L = __builtin_amdgcn_logf(1.f + e) * 0.693147180559945309f;  // v_log_f32 is log2
  FocalParts f;
  const float er = e * r;
  f.p = x >= 0.f ? r : er;
  f.q = x >= 0.f ? er : r;
  // log(max(p, FLT_MIN)) of the reference: p underflows for x < -87.3, where the clamp takes over
  f.logp = fmaxf(fminf(x, 0.f) - L, -87.33654475055310898657f);
  f.log1mp = -fmaxf(x, 0.f) - L;
  return f;
}

__device__ __forceinline__ float focal_pow(float a, float gamma) {  // a in [0, 1]
  if (gamma == 2.f) return a * a;   // RetinaNet's gamma (config/defaults.py:433)
  if (gamma == 0.f) return 1.f;
  if (gamma == 1.f) return a;
  return __builtin_amdgcn_exp2f(gamma * __builtin_amdgcn_logf(a));  // 0 -> exp2(-inf) = 0 for gamma > 0
}

__device__ __forceinline__ float focal_fwd_elem(float x, int t, int d, float gamma, float alpha) {
  if (t < 0) return 0.f;
  const FocalParts f = focal_parts(x);
  if (t == d + 1) return -alpha * (focal_pow(f.q, gamma) * f.logp);
  return -(1.f - alpha) * (focal_pow(f.p, gamma) * f.log1mp);
}

__device__ __forceinline__ float focal_bwd_elem(float x, int t, int d, float gamma, float alpha,
                                                float up) {
  if (t < 0) return 0.f;
  const FocalParts f = focal_parts(x);
  float g;
  if (t == d + 1) g = -alpha * (focal_pow(f.q, gamma) * (f.q - f.p * gamma * f.logp));
  else g = -(1.f - alpha) * (focal_pow(f.p, gamma) * (f.log1mp * f.q * gamma - f.p));
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
