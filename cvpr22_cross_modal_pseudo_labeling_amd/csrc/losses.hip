// Fused forward+backward loss kernels of the student heads (fp32).
//
//  * background-weighted softmax cross-entropy of the embedding classifier
//      maskrcnn_benchmark/modeling/roi_heads/box_head/loss.py:172-174:
//      loss = sum_p w[label_p] * (logsumexp(x_p) - x_p[label_p]) / P,   w[0] = bg_weight, w[c>0] = 1
//  * stochastic-logit ("uncertainty") mask BCE
//      roi_heads/mask_head/roi_mask_predictors.py:41-65 + mask_head/loss.py:117-148:
//      z = mu + eps * sigma ; loss = mean over positives of BCEWithLogits(z[:, channel], target)
// Both write the loss AND the gradient w.r.t. their inputs in one pass (the logits are read once);
// the autograd wrappers only scale the stored gradient by the upstream scalar.
#include "ovis_common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// one wave per row; 4 rows per workgroup
__global__ __launch_bounds__(256) void weighted_ce_kernel(const float* __restrict__ logits,
                                                         const long long* __restrict__ labels,
                                                         float* __restrict__ dlogits,
                                                         float* __restrict__ row_loss, int P, int C,
                                                         float bg_weight) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  const float* x = logits + (long)p * C;
  const long long lab = labels[p];
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
  s = wave_sum(s);
  const bool valid = lab >= 0 && lab < C;
  const float w = !valid ? 0.f : (lab == 0 ? bg_weight : 1.f);
  const float inv_p = 1.f / (float)P;
  if (lane == 0) row_loss[p] = valid ? w * ((m + logf(s)) - x[lab]) * inv_p : 0.f;
  if (dlogits) {
    float* g = dlogits + (long)p * C;
    const float scale = w * inv_p, inv_s = 1.f / s;
    for (int c = lane; c < C; c += 64) g[c] = scale * (expf(x[c] - m) * inv_s - (c == lab ? 1.f : 0.f));
  }
}

// deterministic single-workgroup sum of n floats -> out[0]
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ v, float* __restrict__ out, int n,
                                                  float scale) {
  __shared__ float part[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += part[k];
    out[0] = t * scale;
  }
}

// one workgroup per positive RoI; MM = M*M mask pixels
__global__ __launch_bounds__(256) void mask_bce_kernel(const float* __restrict__ mu,
                                                      const float* __restrict__ sigma,
                                                      const float* __restrict__ eps,
                                                      const long long* __restrict__ pos_index,
                                                      const float* __restrict__ targets,
                                                      float* __restrict__ dmu, float* __restrict__ dsigma,
                                                      float* __restrict__ row_loss, int num_pos, int C, int MM,
                                                      int channel_fixed, const long long* __restrict__ channels) {
  __shared__ float part[4];
  const int i = blockIdx.x;
  const long p = pos_index[i];
  // per-positive logit channel (class-specific masks: labels_pos, mask_head/loss.py:131-141) or one for all (class-agnostic)
  int channel = channel_fixed;
  if (channels) {
    const long long c = channels[i];
    channel = c < 0 ? 0 : (c >= C ? C - 1 : (int)c);
  }
  const float* mu_p = mu + (p * C + channel) * MM;
  const float* ep = eps ? eps + (p * C + channel) * MM : nullptr;
  const float* sg = sigma ? sigma + p * MM : nullptr;
  const float* t = targets + (long)i * MM;
  const float inv_n = 1.f / ((float)num_pos * (float)MM);
  float acc = 0.f;
  for (int k = threadIdx.x; k < MM; k += 256) {
    const float e = ep ? ep[k] : 0.f;
    const float z = mu_p[k] + e * (sg ? (mu_p[k] * 0.0f + sg[k]) : 0.f);
    // BCEWithLogits: max(z,0) - z*t + log1p(exp(-|z|))
    acc += fmaxf(z, 0.f) - z * t[k] + log1pf(expf(-fabsf(z)));
    const float g = (1.f / (1.f + expf(-z)) - t[k]) * inv_n;
    if (dmu) dmu[(p * C + channel) * MM + k] = g;
    if (dsigma) dsigma[p * MM + k] = g * e;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) row_loss[i] = (part[0] + part[1] + part[2] + part[3]) * inv_n;
}

}  // namespace

extern "C" int ovis_weighted_ce_fwd_bwd_f32(const float* logits, const int64_t* labels, float bg_weight,
                                            float* loss, float* dlogits, float* row_scratch, int num_rows,
                                            int num_classes, void* stream) {
  if (num_rows < 0 || num_classes <= 0 || !loss) return OVIS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (num_rows == 0) {
    OVIS_HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), s));
    return OVIS_OK;
  }
  if (!logits || !labels || !row_scratch) return OVIS_EINVAL;
  hipLaunchKernelGGL(weighted_ce_kernel, dim3(ovis_ceil_div(num_rows, 4)), dim3(256), 0, s, logits,
                     (const long long*)labels, dlogits, row_scratch, num_rows, num_classes, bg_weight);
  OVIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, row_scratch, loss, num_rows, 1.f);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

static int mask_bce_launch(const float* mu, const float* sigma, const float* eps, const int64_t* pos_index,
                           const float* targets, float* loss, float* dmu, float* dsigma, float* row_scratch,
                           int num_rois, int num_pos, int num_channels, int mask_pixels, int channel,
                           const int64_t* channels, void* stream) {
  if (num_rois < 0 || num_pos < 0 || num_channels <= 0 || mask_pixels <= 0 || channel < 0 ||
      channel >= num_channels || !loss)
    return OVIS_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dmu) OVIS_HIP_TRY(hipMemsetAsync(dmu, 0, sizeof(float) * (size_t)num_rois * num_channels * mask_pixels, s));
  if (dsigma) OVIS_HIP_TRY(hipMemsetAsync(dsigma, 0, sizeof(float) * (size_t)num_rois * mask_pixels, s));
  if (num_pos == 0) {
    OVIS_HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), s));
    return OVIS_OK;
  }
  if (!mu || !pos_index || !targets || !row_scratch) return OVIS_EINVAL;
  hipLaunchKernelGGL(mask_bce_kernel, dim3(num_pos), dim3(256), 0, s, mu, sigma, eps,
                     (const long long*)pos_index, targets, dmu, dsigma, row_scratch, num_pos, num_channels,
                     mask_pixels, channel, (const long long*)channels);
  OVIS_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, row_scratch, loss, num_pos, 1.f);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}

extern "C" int ovis_mask_bce_stochastic_fwd_bwd_f32(const float* mu, const float* sigma, const float* eps,
                                                    const int64_t* pos_index, const float* targets,
                                                    float* loss, float* dmu, float* dsigma,
                                                    float* row_scratch, int num_rois, int num_pos,
                                                    int num_channels, int mask_pixels, int channel,
                                                    void* stream) {
  return mask_bce_launch(mu, sigma, eps, pos_index, targets, loss, dmu, dsigma, row_scratch, num_rois, num_pos,
                         num_channels, mask_pixels, channel, nullptr, stream);
}

extern "C" int ovis_mask_bce_stochastic_classes_fwd_bwd_f32(const float* mu, const float* sigma, const float* eps,
                                                            const int64_t* pos_index, const int64_t* channels,
                                                            const float* targets, float* loss, float* dmu,
                                                            float* dsigma, float* row_scratch, int num_rois,
                                                            int num_pos, int num_channels, int mask_pixels,
                                                            void* stream) {
  if (num_pos > 0 && !channels) return OVIS_EINVAL;
  return mask_bce_launch(mu, sigma, eps, pos_index, targets, loss, dmu, dsigma, row_scratch, num_rois, num_pos,
                         num_channels, mask_pixels, 0, channels, stream);
}
