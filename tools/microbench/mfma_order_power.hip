// Micro-benchmark: does the ORDER of the MFMAs of a 64x64 wave tile change the sustained rate under the power limit?
// 16 accumulators, 4 A and 4 B fragments per k-step, random bf16 operands, every SIMD busy.  Orders: (0) A-major: four
// consecutive instructions share the A fragment; (1) B-major: share the B fragment; (2) diagonal: neither operand repeats
// between neighbours.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

template <int ORDER>
__global__ __launch_bounds__(256) void k(const s8* __restrict__ src, float* out, int iters) {
  bf8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(bf8, src[(i * 256 + threadIdx.x) & 4095]);
    b[i] = __builtin_bit_cast(bf8, src[((i + 8) * 256 + threadIdx.x) & 4095]);
  }
  f4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int o = u & 4;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int f, g;
        if (ORDER == 0) { f = i >> 2; g = i & 3; }
        else if (ORDER == 1) { g = i >> 2; f = i & 3; }
        else { f = i & 3; g = (i + (i >> 2)) & 3; }
        acc[f * 4 + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[f + o], b[g + o], acc[f * 4 + g], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].w;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  s8* src; float* out;
  hipMalloc(&src, 4096 * sizeof(s8)); hipMalloc(&out, 4096 * 256 * 4);
  short* h = (short*)malloc(4096 * 16);
  srand(7);
  for (int i = 0; i < 4096 * 8; ++i) {
    int e = 123 + rand() % 8;
    h[i] = (short)(((rand() & 1) << 15) | (e << 7) | (rand() & 127));
  }
  hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int round = 0; round < 3; ++round)
    for (int order = 0; order < 3; ++order) {
      hipEventRecord(e0, 0);
      for (int l = 0; l < 5; ++l) {
        if (order == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, src, out, iters);
        else if (order == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(k<2>, dim3(1024), dim3(256), 0, 0, src, out, iters);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = 5.0 * 1024 * 4 * (double)iters * 2097152.0;
      printf("round %d order %d (%s)  %7.2f ms  %7.1f TFLOP/s\n", round, order,
             order == 0 ? "A shared by 4 neighbours" : (order == 1 ? "B shared by 4 neighbours" : "diagonal"), ms, flop / ms * 1e-9);
    }
  return 0;
}
