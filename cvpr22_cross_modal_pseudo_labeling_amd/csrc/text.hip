// Text side of the cross-modal head for gfx950 (MI355X).
//
// Reference: maskrcnn_benchmark/modeling/language_backbone/transformers.py:27-68 (BERT.forward: tokenise, then
// ``self.embeddings[input_ids]`` -- a lookup in the frozen BERT word-embedding table, no transformer forward) and
// maskrcnn_benchmark/modeling/detector/st_generalized_rcnn.py:202-209 (extract_emb):
//     mask       = 1 - special_tokens_mask                                  [N, L]   ([CLS] / [SEP] / [PAD] -> 0)
//     embeddings = (input_embeddings * mask[:, :, None]).sum(1) / mask.sum(1)[:, None]
//     embeddings = F.normalize(embeddings, dim=-1)                          x / max(||x||_2, 1e-12)
// The reference materialises [N, L, D] (1203 LVIS names x ~6 tokens x 768 floats, every iteration); here one workgroup
// per word gathers its rows, averages and normalises in registers: N * D * 4 bytes written, N * (real tokens) rows read.
// The masked rows are still multiplied in (weight 0), as the reference does, so a non-finite table row under a special
// token poisons the word exactly as it would there.  A token id outside the table makes the whole word NaN (the
// reference's indexing raises; a kernel cannot).
#include "ovis_common.h"

namespace {

constexpr int kTextThreads = 256;

__global__ __launch_bounds__(kTextThreads) void text_embed_kernel(const float* __restrict__ table, long rows, int dim,
                                                                 const int* __restrict__ ids,
                                                                 const int* __restrict__ special, int L,
                                                                 float* __restrict__ out) {
  __shared__ float red[kTextThreads / 64];
  const int n = blockIdx.x;
  const int* my_ids = ids + (long)n * L;
  const int* my_sp = special + (long)n * L;
  float count = 0.f;
  bool bad = false;  // every lane walks the whole (short) token list: uniform over the workgroup
  for (int l = 0; l < L; ++l) {
    count += 1.f - (float)my_sp[l];
    bad |= my_ids[l] < 0 || my_ids[l] >= rows;
  }
  // dim / 4 float4 columns, kTextThreads lanes: 768 -> 192 float4, one per lane of the first three waves
  const int d4 = dim >> 2;
  float sq = 0.f;
  for (int c = threadIdx.x; c < d4; c += kTextThreads) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!bad) {
      for (int l = 0; l < L; ++l) {
        const float m = 1.f - (float)my_sp[l];
        const float4 v = ((const float4*)(table + (long)my_ids[l] * dim))[c];
        acc.x += v.x * m;
        acc.y += v.y * m;
        acc.z += v.z * m;
        acc.w += v.w * m;
      }
    }
    acc.x /= count;
    acc.y /= count;
    acc.z /= count;
    acc.w /= count;
    sq += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
    ((float4*)(out + (long)n * dim))[c] = acc;  // un-normalised mean; rescaled below by the same lane
  }
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  float total = 0.f;
  for (int w = 0; w < kTextThreads / 64; ++w) total += red[w];
  const float denom = fmaxf(sqrtf(total), 1e-12f);
  const float nan = __int_as_float(0x7fc00000);
  for (int c = threadIdx.x; c < d4; c += kTextThreads) {
    float4 v = ((float4*)(out + (long)n * dim))[c];
    v.x = bad ? nan : v.x / denom;
    v.y = bad ? nan : v.y / denom;
    v.z = bad ? nan : v.z / denom;
    v.w = bad ? nan : v.w / denom;
    ((float4*)(out + (long)n * dim))[c] = v;
  }
}

}  // namespace

extern "C" int ovis_text_embed_f32(const float* table, long table_rows, int dim, const int32_t* input_ids,
                                   const int32_t* special_tokens_mask, int num_words, int max_tokens, float* out,
                                   void* stream) {
  if (num_words < 0 || max_tokens < 0 || dim <= 0 || table_rows <= 0) return OVIS_EINVAL;
  if (num_words == 0) return OVIS_OK;
  if (!table || !input_ids || !special_tokens_mask || !out) return OVIS_EINVAL;
  if (dim % 4 != 0) return OVIS_ERANGE;
  if (((uintptr_t)table & 15) != 0 || ((uintptr_t)out & 15) != 0) return OVIS_EINVAL;
  hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)num_words), dim3(kTextThreads), 0, (hipStream_t)stream, table,
                     table_rows, dim, input_ids, special_tokens_mask, max_tokens, out);
  OVIS_LAUNCH_CHECK();
  return OVIS_OK;
}
