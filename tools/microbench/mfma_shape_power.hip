// Micro-benchmark: sustained rate of the two dense bf16 MFMA shapes on RANDOM operands, register-resident, every SIMD of the
// part busy (4 waves per SIMD), long enough (tens of ms per case, repeated) for the power management to settle.  Question
// behind it: the large split GEMMs run power-limited at 1.75-1.9 GHz; does the 32x32x16 shape -- half the
// operand-register reads per flop -- sustain a higher rate than the 16x16x32 the kernels use?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const s8* __restrict__ src, float* out, int iters) {
  // 8 A and 8 B fragments per lane, random bit patterns of moderate exponent (generated on the host)
  bf8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(bf8, src[(i * 256 + threadIdx.x) & 4095]);
    b[i] = __builtin_bit_cast(bf8, src[((i + 8) * 256 + threadIdx.x) & 4095]);
  }
  float s = 0.f;
  if (SHAPE == 0) {
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i)  // a 64x64 wave tile: 4 A x 4 B fragments per k-step
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i >> 2) + (u & 4)], b[(i & 3) + (u & 4)], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].w;
  } else {
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)  // a 64x64 wave tile: 2 A x 2 B fragments per k-step of 16
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i >> 1) + 2 * (u & 3)], b[(i & 1) + 2 * (u & 3)], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  s8* src; float* out;
  hipMalloc(&src, 4096 * sizeof(s8)); hipMalloc(&out, 4096 * 256 * 4);
  short* h = (short*)malloc(4096 * 16);
  srand(7);
  for (int i = 0; i < 4096 * 8; ++i) {  // bf16 with random sign / mantissa, exponent in [2^-4, 2^3]
    int e = 123 + rand() % 8;
    h[i] = (short)(((rand() & 1) << 15) | (e << 7) | (rand() & 127));
  }
  for (int zero = 0; zero < 2; ++zero) {
    if (zero) for (int i = 0; i < 4096 * 8; ++i) h[i] = 0;
    hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape) {
      const int iters = 4000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        for (int l = 0; l < 5; ++l) {
          if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, src, out, iters);
          else hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, src, out, iters);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per wave and iteration: 128 MFMAs x 16384 flop (shape 0) or 64 x 32768 (shape 1) = 2 097 152 flop
        const double flop = 5.0 * 1024 * 4 * (double)iters * 2097152.0;
        printf("%s operands  %-10s rep %d  %7.2f ms  %7.1f TFLOP/s\n", zero ? "zero  " : "random", shape ? "32x32x16" : "16x16x32", rep, ms,
               flop / ms * 1e-9);
      }
    }
  }
  return 0;
}
