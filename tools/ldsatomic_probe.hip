// Diagnostic: cost of ds_add_f32 vs private ds_read/add/ds_write, by active lanes and waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int MODE>
__global__ __launch_bounds__(512) void probe(long long* out, int active, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < 8192; k += blockDim.x) lds[k] = 0.f;
  __syncthreads();
  float v = (float)lane;
  // MODE 0: shared block, atomics; MODE 1: wave-private block, read-add-write; MODE 2: shared block atomics, stride-17 addresses
  float* base = (MODE == 1) ? lds + w * 1024 : lds;
  long long t0 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)");
  t0 = wall_clock64(); (void)t0;
  unsigned long long c0, c1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c0));
  if (lane < active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int idx = (MODE == 2) ? ((lane * 17 + j * 64) & 1023) : (j * 64 + lane);
        if (MODE == 1) base[idx] += v; else atomicAdd(base + idx, v);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(c1));
  if (lane == 0) out[blockIdx.x * 8 + w] = (long long)(c1 - c0);
  if (base[lane] == 12345.f) out[0] = 0;
}
int main() {
  long long* d; hipMalloc(&d, 256 * 8 * 8);
  std::vector<long long> h(256 * 8);
  const int iters = 8;
  for (int mode = 0; mode < 3; ++mode)
    for (int waves : {1, 4, 8})
      for (int active : {64, 32, 16, 4}) {
        for (int rep = 0; rep < 2; ++rep) {
          if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(64 * waves), 32768, 0, d, active, iters);
          if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(64 * waves), 32768, 0, d, active, iters);
          if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(64 * waves), 32768, 0, d, active, iters);
          hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, 256 * 8 * 8, hipMemcpyDeviceToHost);
        std::vector<long long> v;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) v.push_back(h[b * 8 + w]);
        std::sort(v.begin(), v.end());
        printf("mode %d waves/CU %d active %2d : median %7.1f memtime-ticks per wave-instruction (x%d instr)\n", mode, waves, active,
               (double)v[v.size() / 2] / (16.0 * iters), 16 * iters);
      }
  return 0;
}
