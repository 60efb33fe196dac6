// elementwise_kernels.hip -- the dropout sites of the joint head fused with what surrounds them (joint_gnn.py:188-198
// per-row stacks `dropout(act(lin(x)))`, :376-389 `x + dropout(attn)`, `x + dropout(ff(...))`, ff = Linear, ReLU, Dropout,
// Linear), on compact [rows][dim] fp32 activations:
//
//     dropout_add   y = x + a * f          (x optional)            backward: d a = g * f, d x = g (no kernel)
//     act_dropout   y = act(t) * f         act = LeakyReLU(slope)  backward: d t = g * f * (y > 0 ? 1 : slope)
//
// f = the inverted-dropout factor (0 or 1 / (1 - p)) of element (row, column): a pure function of the step's
// {seed, offset} pair, the call site's stream id, the row and the 8-column block (gvp_rng.h, the generator of the
// encoders' in-kernel dropout) -- the backward regenerates it, no mask tensor exists and the pair's memory-resident
// offset advances once per step (cgvp_rng_next), so HIP-graph replays draw fresh masks.  The stock sequence is two
// launches forward (activation / dropout, or dropout / add) and two or three backward per site; here it is one each way.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvp_internal.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;

struct Factors { f4 lo, hi; };
__device__ __forceinline__ Factors factors(const gvp::RngArgs& rng, int64_t row, int blk) {
  Factors f{f4{1.f, 1.f, 1.f, 1.f}, f4{1.f, 1.f, 1.f, 1.f}};
  if (rng.seed) {
    float v[8];
    gvp::dropout8_raw(rng.seed[0], rng.seed[1], rng.stream, row, blk, rng.p, v);
    f.lo = f4{v[0], v[1], v[2], v[3]};
    f.hi = f4{v[4], v[5], v[6], v[7]};
  }
  return f;
}

// MODE 0: y = x + a f (x may be NULL)      1: out = g f
//      2: y = act(t) f                      3: gt = g f act'(y)
template <int MODE>
__global__ __launch_bounds__(TPB) void drop_kernel(const float* __restrict__ a, const float* __restrict__ b, gvp::RngArgs rng,
                                                   float slope, int64_t rows, int blocks_per_row, float* __restrict__ out) {
  const int64_t total = rows * blocks_per_row;
  for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TPB) {
    const int64_t row = t / blocks_per_row;
    const int blk = (int)(t - row * blocks_per_row);
    const int64_t at = t * 8;                                  // = row * dim + 8 * blk
    const Factors f = factors(rng, row, blk);
    f4 v0 = *reinterpret_cast<const f4*>(a + at), v1 = *reinterpret_cast<const f4*>(a + at + 4);
    if (MODE == 0) {
      v0 *= f.lo; v1 *= f.hi;
      if (b) { v0 += *reinterpret_cast<const f4*>(b + at); v1 += *reinterpret_cast<const f4*>(b + at + 4); }
    } else if (MODE == 1) {
      v0 *= f.lo; v1 *= f.hi;
    } else if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v0[k] = (v0[k] > 0.f ? v0[k] : v0[k] * slope) * f.lo[k];
        v1[k] = (v1[k] > 0.f ? v1[k] : v1[k] * slope) * f.hi[k];
      }
    } else {
      const f4 y0 = *reinterpret_cast<const f4*>(b + at), y1 = *reinterpret_cast<const f4*>(b + at + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v0[k] *= f.lo[k] * (y0[k] > 0.f ? 1.f : slope);
        v1[k] *= f.hi[k] * (y1[k] > 0.f ? 1.f : slope);
      }
    }
    *reinterpret_cast<f4*>(out + at) = v0;
    *reinterpret_cast<f4*>(out + at + 4) = v1;
  }
}

int check(const float* a, const cgvp_rng* rng, int64_t rows, int32_t dim, const float* out) {
  if (rows < 0 || dim <= 0 || (dim & 7)) return dim > 0 && (dim & 7) ? CGVP_ERR_UNSUPPORTED_DIMS : CGVP_ERR_BAD_ARG;
  if (rows > 0 && (!a || !out)) return CGVP_ERR_BAD_ARG;
  if (((uintptr_t)a & 15) || ((uintptr_t)out & 15)) return CGVP_ERR_BAD_ARG;
  if (rng && rng->seed && !(rng->p >= 0.f && rng->p < 1.f)) return CGVP_ERR_BAD_ARG;
  if (rng && ((uintptr_t)rng->seed & 7)) return CGVP_ERR_BAD_ARG;
  return 0;
}
gvp::RngArgs args_of(const cgvp_rng* rng) {
  if (!rng || !rng->seed || rng->p <= 0.f) return gvp::RngArgs{nullptr, 0.f, 0};
  return gvp::RngArgs{reinterpret_cast<const unsigned long long*>(rng->seed), rng->p, rng->stream};
}
template <int MODE>
int launch(const float* a, const float* b, const cgvp_rng* rng, float slope, int64_t rows, int32_t dim, float* out, void* stream) {
  if (rows == 0) return 0;
  const int64_t total = rows * (dim / 8);
  int64_t g = (total + TPB - 1) / TPB;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(drop_kernel<MODE>, dim3((unsigned)g), dim3(TPB), 0, (hipStream_t)stream, a, b, args_of(rng), slope, rows,
                     dim / 8, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

int cgvp_dropout_add(const float* a, const float* x, const cgvp_rng* rng, int64_t rows, int32_t dim, float* y, void* stream) {
  if (int rc = check(a, rng, rows, dim, y)) return rc;
  if ((uintptr_t)x & 15) return CGVP_ERR_BAD_ARG;
  return launch<0>(a, x, rng, 0.f, rows, dim, y, stream);
}

int cgvp_dropout_scale(const float* g, const cgvp_rng* rng, int64_t rows, int32_t dim, float* out, void* stream) {
  if (int rc = check(g, rng, rows, dim, out)) return rc;
  return launch<1>(g, nullptr, rng, 0.f, rows, dim, out, stream);
}

int cgvp_act_dropout_fwd(const float* t, const cgvp_rng* rng, float slope, int64_t rows, int32_t dim, float* y, void* stream) {
  if (int rc = check(t, rng, rows, dim, y)) return rc;
  return launch<2>(t, nullptr, rng, slope, rows, dim, y, stream);
}

int cgvp_act_dropout_bwd(const float* g, const float* y, const cgvp_rng* rng, float slope, int64_t rows, int32_t dim, float* gt,
                         void* stream) {
  if (int rc = check(g, rng, rows, dim, gt)) return rc;
  if (rows > 0 && (!y || ((uintptr_t)y & 15))) return CGVP_ERR_BAD_ARG;
  return launch<3>(g, y, rng, slope, rows, dim, gt, stream);
}

}  // extern "C"
