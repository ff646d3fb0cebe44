// Micro-benchmark: issue cost of the small MFMA shapes the RoIAlign backward chain uses, dependent (one accumulator)
// and independent (4 accumulators), one wave per SIMD.  Prints cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  s4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
  s8 a8 = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b8 = a8;
  float af = threadIdx.x * 1e-3f, bf = 1.f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
        if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a8), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b8), acc[i], 0, 0, 0);
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int NACC> void run(const char* name, float* d, long long* c) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(256), 0, 0, d, iters, c);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-40s %.1f cycles / MFMA / SIMD (s_memtime ticks %.1f)\n", name, 0.0, (double)h / (iters * 8.0 * NACC));
}
int main() {
  float* d; long long* c; hipMalloc(&d, 1 << 20); hipMalloc(&c, 8);
  run<0, 1>("16x16x16 bf16, dependent", d, c); run<0, 4>("16x16x16 bf16, 4 accumulators", d, c);
  run<1, 1>("16x16x4 f32, dependent", d, c);   run<1, 4>("16x16x4 f32, 4 accumulators", d, c);
  run<2, 1>("16x16x32 bf16, dependent", d, c); run<2, 4>("16x16x32 bf16, 4 accumulators", d, c);
  return 0;
}
