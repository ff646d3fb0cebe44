// Does ds_read_b128 / ds_write_b128 work -- and at what cost -- at addresses that are only 4-byte aligned (gfx950)?
// Pattern of the RoIAlign backward's plane update if a lane owned four CONSECUTIVE cells of a row: lane (q = lane & 15,
// s = lane >> 4) touches floats [(q * W + 4 s + shift) .. + 3], W = 84, shift = 0..3; compared with the shipped pattern
// (four rows W apart: two ds_read2_b32 + two ds_write2_b32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int shift, int iters) {
  __shared__ __attribute__((aligned(16))) float plane[84 * 20 + 16];
  const int lane = threadIdx.x, q = lane & 15, s = lane >> 4;
  for (int i = lane; i < 84 * 20 + 16; i += 64) plane[i] = (float)i;
  __syncthreads();
  float acc = 0.f;
  if (MODE == 0) {  // wide: 16-byte read-modify-write of 4 consecutive cells
    float* p = plane + q * 84 + 4 * s + shift;
    for (int it = 0; it < iters; ++it) {
      f4 v;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)p) : "memory");
      v += (f4){1.f, 2.f, 3.f, 4.f};
      asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"((unsigned)(size_t)p), "v"(v) : "memory");
      acc += v.x;
    }
  } else {          // shipped: four rows W apart
    float* p = plane + (4 * s) * 84 + q + shift;
    for (int it = 0; it < iters; ++it) {
      float a = p[0], b = p[84], c = p[168], d = p[252];
      p[0] = a + 1.f; p[84] = b + 2.f; p[168] = c + 3.f; p[252] = d + 4.f;
      __builtin_amdgcn_s_waitcnt(0);
      acc += a;
    }
  }
  __syncthreads();
  float sum = 0.f;
  for (int i = lane; i < 84 * 20; i += 64) sum += plane[i];
  out[blockIdx.x * 64 + lane] = sum + acc * 0.f;
}
template <int MODE>
void run(const char* name, int shift) {
  float* d; hipMalloc(&d, 4 * 64 * 1024);
  const int iters = 2000, blocks = 1024;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, shift, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, shift, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<float> h(64); hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost);
  double total = 0; for (float v : h) total += v;
  // expected: sum of initial plane + iters * (1+2+3+4) * 64 lanes (every lane adds 10 per iteration)
  double init = 0; for (int i = 0; i < 84 * 20; ++i) init += i;
  printf("%-28s shift %d: %.3f ms  %.1f ns per RMW of a wave   sum %s (%.0f vs %.0f)\n", name, shift, ms,
         ms * 1e6 / iters / (blocks / 256.0 / 8), (fabs(total - (init + 10.0 * 64 * iters)) < 1e-3 * total) ? "ok" : "WRONG",
         total, init + 10.0 * 64 * iters);
  hipFree(d);
}
int main() {
  for (int shift = 0; shift < 4; ++shift) { run<0>("ds_read/write_b128", shift); }
  run<1>("4 x b32 rows (shipped)", 0); run<1>("4 x b32 rows (shipped)", 3);
  return 0;
}
