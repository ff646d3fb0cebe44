// Micro-benchmark: global fp32 atomic add on gfx950 by memory scope.  Every block adds 1.0 to each of N floats
// (lane-contiguous), so the exact answer is gridDim.x; "agent" is HIP's atomicAdd default, "workgroup" keeps the
// RMW in the issuing XCD's L2 (no sc1) and is only coherent when all writers of a line sit on one XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SCOPE, bool PRIVATE>
__global__ __launch_bounds__(256) void k(float* buf, int n) {
  unsigned xcc = 0;
  if (PRIVATE) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; }
  float* p = buf + (size_t)xcc * n;
  for (int i = threadIdx.x; i < n; i += 256) __hip_atomic_fetch_add(p + i, 1.0f, __ATOMIC_RELAXED, SCOPE);
}
__global__ void reduce8(const float* buf, float* out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { float s = 0; for (int x = 0; x < 8; ++x) s += buf[(size_t)x * n + i]; out[i] = s; }
}
template <int SCOPE, bool PRIVATE>
void run(const char* name, float* d, float* out, int n, int blocks) {
  hipMemset(d, 0, sizeof(float) * n * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<SCOPE, PRIVATE>), dim3(blocks), dim3(256), 0, 0, d, n);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipLaunchKernelGGL(reduce8, dim3((n + 255) / 256), dim3(256), 0, 0, d, out, n);
  std::vector<float> h(n); hipMemcpy(h.data(), out, sizeof(float) * n, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < n; ++i) bad += (h[i] != (float)blocks);
  double atom = (double)blocks * n;
  printf("%-34s %.3f ms  %.1f G lane-atomics/s  wrong=%d/%d (e.g. %.0f vs %d)\n", name, ms, atom / ms / 1e6, bad, n, h[1], blocks);
}
int main() {
  const int n = 1 << 16, blocks = 4096;   // 256 KB of floats, 268M lane-atomics
  float *d, *out; hipMalloc(&d, sizeof(float) * n * 8); hipMalloc(&out, sizeof(float) * n);
  run<__HIP_MEMORY_SCOPE_AGENT, false>("agent scope, one buffer", d, out, n, blocks);
  run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("workgroup scope, one buffer", d, out, n, blocks);
  run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, per-XCC buffers", d, out, n, blocks);
  run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, per-XCC buffers", d, out, n, blocks);
  return 0;
}
