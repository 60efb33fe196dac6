// norm_kernels.hip -- row-wise nn.LayerNorm of the joint head (joint_gnn.py:376-389: preattn_norm{1,2}, ff_norm{1,2} on the
// compact residue / atom rows), forward and backward.
//
// The head normalises [19,200 x 128] residue rows four times per training step (two forward, two backward).  The stock
// kernels take 15 us forward and 28 + 12 + 5 us backward (grad-input, partial gamma / beta, final gamma / beta) for what
// is 20 MB / 30 MB of HBM traffic (profiles/r03/kernel_stats_davis_b64_joint.csv: 195 us per step over the four norms).
// Here: one wave per row, the row in registers (dim / 64 values per lane, coalesced), two-pass mean / variance like
// torch's, and in the backward ONE pass that writes d x and accumulates d gamma / d beta per wave in registers over a
// grid-stride walk of the rows -- per-workgroup partials go to a slab row, quad::reduce_segments adds them in a fixed
// order (deterministic, like every weight gradient of this library).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvp_internal.h"

namespace {

constexpr int WAVE = 64, WPB = 8, TPB = WAVE * WPB;
constexpr int MAX_WGS = 512;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

template <int VPL>
__global__ __launch_bounds__(TPB) void layer_norm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int64_t rows, float eps,
                                                             float* __restrict__ y, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out) {
  constexpr int D = VPL * WAVE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float g[VPL], b[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) { g[k] = gamma ? gamma[k * WAVE + lane] : 1.f; b[k] = beta ? beta[k * WAVE + lane] : 0.f; }
  for (int64_t r = (int64_t)blockIdx.x * WPB + w; r < rows; r += (int64_t)gridDim.x * WPB) {
    float v[VPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) { v[k] = x[r * D + k * WAVE + lane]; s += v[k]; }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
    for (int k = 0; k < VPL; ++k) y[r * D + k * WAVE + lane] = fmaf((v[k] - mean) * rstd, g[k], b[k]);
    if (lane == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
  }
}

// d x = rstd (a - mean(a) - xhat mean(a xhat)),  a = d y * gamma;  d gamma = sum_rows d y * xhat;  d beta = sum_rows d y
template <int VPL>
__global__ __launch_bounds__(TPB) void layer_norm_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                             const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                             const float* __restrict__ gamma, int64_t rows,
                                                             float* __restrict__ gx, float* __restrict__ slab) {
  constexpr int D = VPL * WAVE;
  __shared__ float part[WPB][2 * D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float g[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) { g[k] = gamma ? gamma[k * WAVE + lane] : 1.f; dg[k] = 0.f; db[k] = 0.f; }
  for (int64_t r = (int64_t)blockIdx.x * WPB + w; r < rows; r += (int64_t)gridDim.x * WPB) {
    const float mean = mean_in[r], rstd = rstd_in[r];
    float xh[VPL], a[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const float dy = gy[r * D + k * WAVE + lane];
      xh[k] = (x[r * D + k * WAVE + lane] - mean) * rstd;
      dg[k] = fmaf(dy, xh[k], dg[k]);
      db[k] += dy;
      a[k] = dy * g[k];
      s1 += a[k];
      s2 = fmaf(a[k], xh[k], s2);
    }
    s1 = wave_sum(s1) * (1.0f / D);
    s2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int k = 0; k < VPL; ++k) gx[r * D + k * WAVE + lane] = rstd * (a[k] - s1 - xh[k] * s2);
  }
#pragma unroll
  for (int k = 0; k < VPL; ++k) { part[w][k * WAVE + lane] = dg[k]; part[w][D + k * WAVE + lane] = db[k]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += TPB) {
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < WPB; ++ww) s += part[ww][c];
    slab[(size_t)blockIdx.x * 2 * D + c] = s;
  }
}

int grid_for(int64_t rows) {
  int64_t wgs = (rows + WPB * 4 - 1) / (WPB * 4);          // >= 4 rows per wave before the grid is capped
  return (int)(wgs < 1 ? 1 : (wgs > MAX_WGS ? MAX_WGS : wgs));
}
bool dim_ok(int dim) { return dim == 64 || dim == 128 || dim == 256 || dim == 512; }

}  // namespace

extern "C" {

int cgvp_layer_norm_fwd(const float* x, const float* gamma, const float* beta, int64_t rows, int32_t dim, float eps,
                        float* y, float* mean, float* rstd, void* stream) {
  if (!dim_ok(dim)) return CGVP_ERR_UNSUPPORTED_DIMS;
  if (rows < 0 || !(eps >= 0.f)) return CGVP_ERR_BAD_ARG;
  if (rows == 0) return 0;
  if (!x || !y || !mean || !rstd) return CGVP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((rows + WPB - 1) / WPB < 4096 ? (rows + WPB - 1) / WPB : 4096));
#define LAUNCH(V) hipLaunchKernelGGL(layer_norm_fwd_kernel<V>, grid, dim3(TPB), 0, st, x, gamma, beta, rows, eps, y, mean, rstd)
  switch (dim) { case 64: LAUNCH(1); break; case 128: LAUNCH(2); break; case 256: LAUNCH(4); break; default: LAUNCH(8); }
#undef LAUNCH
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

int64_t cgvp_layer_norm_bwd_workspace_floats(int64_t rows, int32_t dim) {
  if (!dim_ok(dim)) return CGVP_ERR_UNSUPPORTED_DIMS;
  if (rows < 0) return CGVP_ERR_BAD_ARG;
  return (int64_t)grid_for(rows) * 2 * dim;
}

int cgvp_layer_norm_bwd(const float* gy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        int64_t rows, int32_t dim, float* gx, float* workspace, float* g_gamma_beta, void* stream) {
  if (!dim_ok(dim)) return CGVP_ERR_UNSUPPORTED_DIMS;
  if (rows < 0 || !g_gamma_beta) return CGVP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    quad::zero_words(g_gamma_beta, (size_t)2 * dim, st);
    return 0;
  }
  if (!gy || !x || !mean || !rstd || !gx || !workspace) return CGVP_ERR_BAD_ARG;
  const int G = grid_for(rows);
#define LAUNCH(V) hipLaunchKernelGGL(layer_norm_bwd_kernel<V>, dim3(G), dim3(TPB), 0, st, gy, x, mean, rstd, gamma, rows, gx, workspace)
  switch (dim) { case 64: LAUNCH(1); break; case 128: LAUNCH(2); break; case 256: LAUNCH(4); break; default: LAUNCH(8); }
#undef LAUNCH
  cgvp_segment sg[1] = {{workspace, G, 2 * dim, 0, 2 * dim, 0}};
  quad::reduce_segments(sg, 1, g_gamma_beta, st, 1);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
