// Probe: hi/lo split product of two 16 x 16 f32 matrices on one wave, three ways (K = 16 x3, K = 32 + K = 16, K = 32 x2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ unsigned pack(float a, float b) { f2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2)); }
__device__ u4 split(f4 v) {
  unsigned h01 = pack(v.x, v.y), h23 = pack(v.z, v.w);
  float r0 = v.x - __uint_as_float(h01 << 16), r1 = v.y - __uint_as_float(h01 & 0xffff0000u);
  float r2 = v.z - __uint_as_float(h23 << 16), r3 = v.w - __uint_as_float(h23 & 0xffff0000u);
  return (u4){h01, h23, pack(r0, r1), pack(r2, r3)};
}
__device__ s4 as4(unsigned a, unsigned b) { u2 u = {a, b}; return __builtin_bit_cast(s4, u); }
// A[i][k] row-major 16x16, B[k][j] row-major; out[mode][i][j]
__global__ void probe(const float* A, const float* B, float* out) {
  const int lane = threadIdx.x, q = lane & 15, s = lane >> 4;
  f4 av, bv;
  for (int e = 0; e < 4; ++e) { av[e] = A[q * 16 + 4 * s + e]; bv[e] = B[(4 * s + e) * 16 + q]; }
  const u4 a = split(av), b = split(bv);
  f4 z = {0.f, 0.f, 0.f, 0.f};
  // mode 0: three K = 16 products
  f4 d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as4(a.x, a.y), as4(b.x, b.y), z, 0, 0, 0);
  d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as4(a.x, a.y), as4(b.z, b.w), d0, 0, 0, 0);
  d0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as4(a.z, a.w), as4(b.x, b.y), d0, 0, 0, 0);
  // mode 1: [ah | ah] . [bh ; bl] (K = 32) + al . bh (K = 16)
  const u4 ahh = {a.x, a.y, a.x, a.y}, all = {a.z, a.w, a.z, a.w};
  f4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ahh), __builtin_bit_cast(bf8, b), z, 0, 0, 0);
  d1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(as4(a.z, a.w), as4(b.x, b.y), d1, 0, 0, 0);
  // mode 2: [ah | ah] . [bh ; bl] + [al | al] . [bh ; bl]
  f4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ahh), __builtin_bit_cast(bf8, b), z, 0, 0, 0);
  d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, all), __builtin_bit_cast(bf8, b), d2, 0, 0, 0);
  // mode 3: the K = 32 product alone
  f4 d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, ahh), __builtin_bit_cast(bf8, b), z, 0, 0, 0);
  for (int e = 0; e < 4; ++e) {
    out[0 * 256 + (4 * s + e) * 16 + q] = d0[e];
    out[1 * 256 + (4 * s + e) * 16 + q] = d1[e];
    out[2 * 256 + (4 * s + e) * 16 + q] = d2[e];
    out[3 * 256 + (4 * s + e) * 16 + q] = d3[e];
  }
}
int main() {
  std::vector<float> A(256), B(256), O(1024);
  for (int i = 0; i < 256; ++i) { A[i] = sinf(0.37f * i) + 0.01f * (i % 7); B[i] = cosf(0.11f * i) - 0.02f * (i % 5); }
  float *dA, *dB, *dO;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dO, 4096);
  hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dO);
  hipMemcpy(O.data(), dO, 4096, hipMemcpyDeviceToHost);
  for (int m = 0; m < 4; ++m) {
    double worst = 0, worst_t = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double ref = 0, reft = 0;
      for (int k = 0; k < 16; ++k) { ref += (double)A[i * 16 + k] * B[k * 16 + j]; reft += (double)A[j * 16 + k] * B[k * 16 + i]; }
      worst = fmax(worst, fabs(O[m * 256 + i * 16 + j] - ref)); worst_t = fmax(worst_t, fabs(O[m * 256 + i * 16 + j] - reft));
    }
    printf("mode %d: max |D - A.B| = %.3e   (vs transposed: %.3e)\n", m, worst, worst_t);
  }
  printf("row 0 of modes 0/1/2/3:\n");
  for (int m = 0; m < 4; ++m) { for (int j = 0; j < 6; ++j) printf(" %9.5f", O[m * 256 + j]); printf("\n"); }
  return 0;
}
