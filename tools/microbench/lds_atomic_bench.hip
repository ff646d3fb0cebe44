// Micro-benchmark: LDS throughput of ds_read_b32 / ds_write_b32 / ds_add_f32 / ds_add_u32 / ds_max_i32 on gfx950,
// conflict-free (lane-linear) and 2-way same-address patterns. Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int PAT>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[4096];
  int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = 0.f;
  __syncthreads();
  int lane = tid & 63, wave = tid >> 6;
  int idx = wave * 1024 + (PAT == 0 ? lane : (PAT == 1 ? (lane >> 1) : (lane & 7)));
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int a = idx + u * 64;
      if (MODE == 0) acc += lds[a];
      if (MODE == 1) lds[a] = acc + u;
      if (MODE == 2) atomicAdd(&lds[a], 1.0f);
      if (MODE == 3) atomicAdd((unsigned*)&lds[a], 1u);
      if (MODE == 4) atomicMax((int*)&lds[a], it);
    }
    if (MODE == 0) asm volatile("" ::"v"(acc));
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = acc + lds[5];
}
template <int MODE, int PAT>
void run(const char* name, float* d) {
  const int iters = 2000, blocks = 256 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(256), 0, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr = (double)blocks * 4 * iters * 16;  // wave-instructions
  // per CU: 256 CUs; cycles at 2.4 GHz
  double cyc_per_instr_per_cu = (ms * 1e-3 * 2.4e9) / (instr / 256.0);
  printf("%-28s pat=%d  %.3f ms  %.2f LDS-cycles per wave-instr per CU\n", name, PAT, ms, cyc_per_instr_per_cu);
}
int main() {
  float* d; hipMalloc(&d, 1 << 20);
  run<0, 0>("ds_read_b32", d); run<1, 0>("ds_write_b32", d);
  run<2, 0>("ds_add_f32", d); run<2, 1>("ds_add_f32", d); run<2, 2>("ds_add_f32", d);
  run<3, 0>("ds_add_u32", d); run<3, 1>("ds_add_u32", d); run<3, 2>("ds_add_u32", d);
  run<4, 0>("ds_max_i32", d);
  return 0;
}
