// scratch: apply<G, 1, true> (packed bf16 fragments) vs apply<G, 1, false> on random data, one wave
#include "gvp_quad.h"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace gq;
typedef Gemm<P1, 16, 24, Segs<Seg<P2, 0, 17>, Seg<P2, 17, 4>>> G;    // QNode-like ws: 5 + 1 steps
__global__ void k(const float* W, const float* x, float* out) {
  __shared__ __attribute__((aligned(16))) float f32[G::NFRAG * 64], fpk[G::NFRAG * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < G::NFRAG * 64; i += 64) { f32[i] = G::element(W, i); fpk[i] = packed_element<G>(W, i); }
  __syncthreads();
  float b[1][G::NSTEPS];
  for (int s = 0; s < G::NSTEPS; ++s) b[0][s] = x[s * 64 + lane];
  f4 a0[1] = {f4{0, 0, 0, 0}}, a1[1] = {f4{0, 0, 0, 0}};
  apply<G, 1, false>(f32, 0, b, a0, lane);
  apply<G, 1, true>(fpk, 0, b, a1, lane);
  for (int r = 0; r < 4; ++r) { out[lane * 4 + r] = a0[0][r]; out[256 + lane * 4 + r] = a1[0][r]; }
}
int main() {
  std::vector<float> W(16 * 24), x(G::NSTEPS * 64), o(512);
  for (size_t i = 0; i < W.size(); ++i) W[i] = sinf(0.37f * i);
  for (size_t i = 0; i < x.size(); ++i) x[i] = cosf(0.11f * i);
  float *dW, *dx, *dout;
  hipMalloc(&dW, W.size() * 4); hipMalloc(&dx, x.size() * 4); hipMalloc(&dout, 512 * 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dW, dx, dout);
  hipMemcpy(o.data(), dout, 512 * 4, hipMemcpyDeviceToHost);
  float md = 0, mx = 0;
  for (int i = 0; i < 256; ++i) { md = fmaxf(md, fabsf(o[i] - o[256 + i])); mx = fmaxf(mx, fabsf(o[i])); }
  printf("NSTEPS %d  max |fp32| %f  max diff %f\n", G::NSTEPS, mx, md);
  for (int i = 0; i < 8; ++i) printf("%f %f\n", o[i], o[256 + i]);
}
