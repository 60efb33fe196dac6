// Diagnostic: effective shader clock under light vs sustained load
// (s_memtime = shader cycles, s_memrealtime = 100 MHz).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
__global__ void spin(long iters, unsigned long long* out) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float x = threadIdx.x;
  for (long i = 0; i < iters; ++i) x = fmaf(x, 1.000001f, 0.5f);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
int main() {
  unsigned long long *d, h[3];
  hipMalloc(&d, 24);
  auto run = [&](int blocks, long iters, const char* tag) {
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, iters, d);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%-34s shader %8llu cyc  real %7.2f us  => %.0f MHz\n", tag, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
  };
  run(1, 2000, "cold, 1 block, short");
  run(1, 2000, "1 block, short (again)");
  for (int i = 0; i < 3; ++i) { usleep(2000); run(1, 2000, "after 2 ms idle, 1 block short"); }
  run(1024, 2000, "1024 blocks, short");
  run(1024, 2000000, "1024 blocks, long (sustained)");
  run(1024, 2000, "1024 blocks short, right after long");
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(spin, dim3(480), dim3(256), 0, 0, 3000L, d);
  run(480, 3000, "after 50 back-to-back ~10us kernels");
  return 0;
}
