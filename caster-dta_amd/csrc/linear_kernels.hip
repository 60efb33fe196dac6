// linear_kernels.hip -- weight / bias gradients of the row-wise nn.Linear layers of the joint head (joint_gnn.py:188-198
// residue / atom stacks, :376-389 attention in / out projections and feed-forward):
//
//     dW[o][i] = sum_r dY[r][o] X[r][i]        db[o] = sum_r dY[r][o]          r over ALL residues / atoms of the batch
//
// i.e. GEMMs with a tiny output (128..384 x 64..256) and a reduction over tens of thousands of rows.  The library
// GEMM PyTorch picks for that shape tiles only the output (16 workgroups for 128 x 128) and walks the 19,200 rows
// serially: 44-60 us per layer, 12 layers per training step, plus a separate column-sum kernel per bias
// (profiles/r03/kernel_stats_davis_b64_joint_before.csv).  Here the ROWS are split over the chip:
//
//   grid (row splits S, output-row groups O / 64); 8 waves per workgroup; a workgroup stages 64-row chunks of X and of
//   its 64 columns of dY in LDS (row pitch = 16 mod 32 floats: the operand reads below are bank-conflict free) and
//   runs v_mfma_f32_16x16x4_f32 with the batch rows on K: lane (n, g) of a k-step reads row 4 kk + g -- the memory
//   layout IS the operand layout, nothing is transposed.  Wave w owns output tile row (w & 3) and every second
//   16-column tile of X, plus a constant [1 0 .. 0] tile whose first column accumulates the bias gradient.
//   Partial sums go to a slab row per split; quad::reduce_segments adds the S rows in a fixed order (deterministic).
//
// Exact fp32 (the matrix cores' f32 path is an fmaf chain); only the summation order differs from the library's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvp_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int WAVE = 64, WPB = 8, TPB = WAVE * WPB, KC = 64, OG = 64;      // rows per LDS chunk; output rows per workgroup
constexpr int MAX_NT = 17;                                                  // I <= 256: 16 column tiles + the bias tile
constexpr int MAX_SPLITS = 128;

struct WgArgs {
  const float* x; const float* gy; int64_t R; int I, O; int rows_per_split; float* slab; int slab_stride;
};

template <int NTW>      // column tiles per wave (compile-time so that the accumulators stay in registers)
__device__ __forceinline__ void wgrad_body(const WgArgs& a, float* lds) {
  const int I = a.I, ldx = I + 16, ldy = OG + 16;
  float* xs = lds;                      // [KC][ldx]
  float* ys = lds + KC * ldx;           // [KC][ldy]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
  const int mt = w & 3, par = w >> 2;   // output tile row; parity of the column tiles this wave owns
  const int o0 = blockIdx.y * OG;
  const int nt_all = I / 16 + 1;        // + the bias tile (index I / 16)
  const int64_t r_lo = (int64_t)blockIdx.x * a.rows_per_split;
  const int64_t r_hi = r_lo + a.rows_per_split < a.R ? r_lo + a.rows_per_split : a.R;
  f4 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
  // Register-staged double buffering: the NEXT 64-row chunk's global loads are issued before the current chunk's MFMAs
  // and land in LDS after them (one exposed load latency per workgroup instead of one per chunk).
  const int i4 = I / 4;
  constexpr int XV = KC * (256 / 4) / TPB, YV = KC * (OG / 4) / TPB;       // float4 per thread: X (I <= 256), dY
  f4 xr[XV], yr[YV];
  auto fetch = [&](int64_t rc) {
    const int rows = (int)(r_hi - rc < KC ? r_hi - rc : KC);
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int q = tid + k * TPB;
      const int r = q / i4, c = (q - r * i4) * 4;
      xr[k] = f4{0.f, 0.f, 0.f, 0.f};
      if (q < KC * i4 && r < rows) xr[k] = *reinterpret_cast<const f4*>(a.x + (rc + r) * I + c);
    }
#pragma unroll
    for (int k = 0; k < YV; ++k) {
      const int q = tid + k * TPB;
      const int r = q / (OG / 4), c = (q - r * (OG / 4)) * 4;
      yr[k] = f4{0.f, 0.f, 0.f, 0.f};
      if (r < rows) yr[k] = *reinterpret_cast<const f4*>(a.gy + (rc + r) * a.O + o0 + c);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int k = 0; k < XV; ++k) {
      const int q = tid + k * TPB;
      const int r = q / i4, c = (q - r * i4) * 4;
      if (q < KC * i4) *reinterpret_cast<f4*>(xs + r * ldx + c) = xr[k];
    }
#pragma unroll
    for (int k = 0; k < YV; ++k) {
      const int q = tid + k * TPB;
      const int r = q / (OG / 4), c = (q - r * (OG / 4)) * 4;
      *reinterpret_cast<f4*>(ys + r * ldy + c) = yr[k];
    }
  };
  if (r_lo < r_hi) fetch(r_lo);
  for (int64_t rc = r_lo; rc < r_hi; rc += KC) {
    __syncthreads();                    // the previous chunk's operand reads are done
    stash();
    __syncthreads();
    if (rc + KC < r_hi) fetch(rc + KC);
#pragma unroll 4
    for (int kk = 0; kk < KC / 4; ++kk) {
      const float av = ys[(4 * kk + g) * ldy + 16 * mt + n];                 // A[m = o][k = row]
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int nt = 2 * t + par;
        float bv = 0.f;
        if (nt < nt_all - 1) bv = xs[(4 * kk + g) * ldx + 16 * nt + n];      // B[k = row][n = i]
        else if (nt == nt_all - 1) bv = n == 0 ? 1.f : 0.f;                  // bias tile: column 0 sums dY over the rows
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }
  // D layout: lane (n, g) register r = element (row 4 g + r, column n) of the tile
  float* out = a.slab + (int64_t)blockIdx.x * a.slab_stride;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int nt = 2 * t + par;
    if (nt < nt_all - 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)(o0 + 16 * mt + 4 * g + r) * I + 16 * nt + n] = acc[t][r];
    } else if (nt == nt_all - 1 && n == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(int64_t)a.O * I + o0 + 16 * mt + 4 * g + r] = acc[t][r];
    }
  }
}

template <int NTW>
__global__ __launch_bounds__(TPB) void linear_wgrad_kernel(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  wgrad_body<NTW>(a, lds);
}

int splits_for(int64_t R, int O) {
  const int groups = O / OG;
  int64_t s = 256 / groups;                               // ~one workgroup per CU
  if (s > MAX_SPLITS) s = MAX_SPLITS;
  const int64_t most = (R + KC - 1) / KC;                 // at least one LDS chunk of rows per split
  if (s > most) s = most;
  return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" {

int64_t cgvp_linear_wgrad_workspace_floats(int64_t R, int32_t I, int32_t O) {
  if (R < 0 || I < 16 || I > 256 || (I & 15) || O < OG || (O % OG)) return CGVP_ERR_UNSUPPORTED_DIMS;
  return (int64_t)splits_for(R, O) * ((int64_t)O * I + O);
}

int cgvp_linear_wgrad(const float* x, const float* gy, int64_t R, int32_t I, int32_t O, float* workspace, float* out,
                      void* stream) {
  const int64_t wsz = cgvp_linear_wgrad_workspace_floats(R, I, O);
  if (wsz < 0) return (int)wsz;
  if (!out || !workspace || (R > 0 && (!x || !gy))) return CGVP_ERR_BAD_ARG;
  if (((uintptr_t)x & 15) || ((uintptr_t)gy & 15) || ((uintptr_t)workspace & 15)) return CGVP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int len = O * I + O;
  if (R == 0) {
    quad::zero_words(out, (size_t)len, st);
    return 0;
  }
  const int S = splits_for(R, O);
  int64_t rps = (R + S - 1) / S;
  rps = (rps + 3) / 4 * 4;                                 // whole k-steps; row r of a split stays 16-B aligned (I % 4 == 0)
  WgArgs a{x, gy, R, I, O, (int)rps, workspace, len};
  const dim3 grid((unsigned)((R + rps - 1) / rps), (unsigned)(O / OG));
  const size_t lds = (size_t)(KC * (I + 16) + KC * (OG + 16)) * sizeof(float);
  const int ntw = (I / 16 + 1 + 1) / 2;                    // column tiles of the busier parity
  // The dynamic-LDS attribute is set ONCE per instantiation and device, so it must cover the LARGEST I that instantiation
  // serves (I <= 80 / 144 / 256 for N_ = 3 / 5 / 9), not the I of whichever call happened to come first.
#define LAUNCH(N_, IMAX_)                                                                                       \
  do {                                                                                                          \
    constexpr size_t lds_max_ = (size_t)(KC * ((IMAX_) + 16) + KC * (OG + 16)) * sizeof(float);                  \
    static_assert(((IMAX_) / 16 + 2) / 2 <= (N_) && lds_max_ <= 160 * 1024, "instantiation bound");             \
    CGVP_SET_DYN_LDS_ONCE(linear_wgrad_kernel<N_>, lds_max_);                                                   \
    hipLaunchKernelGGL(linear_wgrad_kernel<N_>, grid, dim3(TPB), lds, st, a);                                   \
  } while (0)
  if (ntw <= 3) LAUNCH(3, 80); else if (ntw <= 5) LAUNCH(5, 144); else LAUNCH(9, 256);
#undef LAUNCH
  cgvp_segment sg[2] = {{workspace, (int)grid.x, len, 0, O * I, 0}, {workspace, (int)grid.x, len, O * I, O, O * I}};
  quad::reduce_segments(sg, 2, out, st, 1);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
