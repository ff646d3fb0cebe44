// LDS bank conflicts of the split GEMM's fragment-read pattern (ds_read_b128: lane (frow = lane & 15, fq = lane >> 4) reads
// the 16-byte chunk fq of row 16 + frow + s) as a function of the row shift s of the halo-tile 3x3 and of the chunk
// swizzle swz(row): physical chunk = fq ^ swz(row).  Run under
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- tools/microbench/lds_shift_probe
// one dispatch per (swizzle id, shift): conflict-free reads show SQ_LDS_BANK_CONFLICT = 0.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz(int id, int r) {
  switch (id) {
    case 0: return (r >> 1) & 7;                    // shipped: pairs of rows share a chunk rotation
    case 1: return r & 7;
    case 2: return (r >> 2) & 7;
    case 3: return ((r >> 1) ^ (r >> 4)) & 7;
    case 4: return ((r & 1) << 2) | ((r >> 1) & 3);
    case 5: return (r >> 1) & 3;
    default: return ((r >> 1) & 3) | ((r >> 2) & 4);
  }
}

__global__ __launch_bounds__(256) void probe(int id, int shift, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[40960];
  for (int i = threadIdx.x; i < 40960 / 4; i += 256) ((float*)lds)[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int row = 16 + (wave & 1) * 64 + frow + shift;
  const int addr = row * 128 + ((fq ^ swz(id, row)) << 4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < 1024; ++it) {
#pragma unroll
    for (int f = 0; f < 4; ++f) acc += *(const f32x4*)(lds + addr + f * 2048);
    asm volatile("" ::: "memory");
  }
  sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  float* sink;
  hipMalloc(&sink, sizeof(float) * 256);
  for (int id = 0; id < 7; ++id)
    for (int s = -8; s <= 8; ++s) hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, id, s, sink);
  hipDeviceSynchronize();
  printf("launched 7 x 17 probes\n");
  return 0;
}
