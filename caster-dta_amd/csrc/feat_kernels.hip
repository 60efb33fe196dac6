// feat_kernels.hip -- protein edge featurisation on the device (opt-in).
//
// The reference computes every residue pair's edge features once per protein on the CPU
// (utils/create_protein_features.py:225-273, calc_pos_encoding :368-386) and stores them with the graph:
// 32 scalars + one unit vector = 140 B per edge that the encoder then reads from HBM.  All of it is a pure function of
// the two C-alpha positions and the two sequence indices (16 B per edge of indices, 12 B per residue of coordinates):
//
//   e_s[ 0:16] = exp(-((d - mu_k) / 1.25)^2),  d = |CA_i - CA_j|, mu_k = linspace(0, 20, 16)        (:233-237)
//   e_s[16:24] = cos((j - i) f_k),  e_s[24:32] = sin((j - i) f_k),  f_k = exp(-2 k ln(10000) / 8)    (:258, :368-386)
//   e_v        = (CA_i - CA_j) / d, zero when d == 0 (self loops, duplicate coordinates)             (:244, :360-365)
//
// for the edge i -> j (row 0 / row 1 of edge_index).  This kernel writes the same tensors in ORIGINAL edge order, so a
// dataset may keep coordinates + indices only and materialise the features per batch.  Eight lanes per edge, one
// float4 of the 32 scalars each (coalesced 128-B rows).  The positional-encoding argument is formed and reduced mod
// 2 pi in fp64 (|j - i| reaches the thousands: an fp32 product would already be off by 1e-4 rad), the rest is fp32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/caster_gvp.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int RBF = 16, POS = 16, ES = RBF + POS;
constexpr double kTwoPi = 6.283185307179586476925286766559;

struct FeatArgs {
  const float* ca; const int64_t* seq; const int64_t* ei; int64_t N; int64_t E; float* e_s; float* e_v;
};

__global__ __launch_bounds__(256) void edge_feat_kernel(FeatArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e = t >> 3;
  const int q = (int)(t & 7);
  if (e >= a.E) return;
  int64_t i = a.ei[e], j = a.ei[a.E + e];
  i = i < 0 ? 0 : (i >= a.N ? a.N - 1 : i);                      // malformed ids never fault
  j = j < 0 ? 0 : (j >= a.N ? a.N - 1 : j);
  f4 out;
  if (q < 4) {
    const float dx = a.ca[3 * i] - a.ca[3 * j], dy = a.ca[3 * i + 1] - a.ca[3 * j + 1], dz = a.ca[3 * i + 2] - a.ca[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mu = (float)(20.0 * (4 * q + r) / 15.0);       // linspace(0, 20, 16)
      const float z = (d - mu) * (1.0f / 1.25f);
      out[r] = expf(-z * z);
    }
    if (q == 0) {
      const float inv = d > 0.f ? 1.0f / d : 0.f;
      a.e_v[3 * e] = dx * inv; a.e_v[3 * e + 1] = dy * inv; a.e_v[3 * e + 2] = dz * inv;
    }
  } else {
    const double diff = (double)(a.seq[j] - a.seq[i]);           // destination index - source index
    const int k0 = 4 * (q & 1);                                  // q = 4,5: cos of freqs 0-3 / 4-7; q = 6,7: sin
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double f = exp(-2.0 * (k0 + r) * (9.210340371976182736 / 8.0));      // ln(10000) / 8 per step
      double x = diff * f;
      x -= kTwoPi * rint(x * (1.0 / kTwoPi));
      out[r] = q < 6 ? cosf((float)x) : sinf((float)x);
    }
  }
  *reinterpret_cast<f4*>(a.e_s + e * ES + 4 * q) = out;
}

}  // namespace

extern "C" int cgvp_edge_featurise(const float* ca_xyz, const int64_t* seq_index, const int64_t* edge_index,
                                   int64_t num_nodes, int64_t num_edges, float* e_s, float* e_v, void* stream) {
  if (num_nodes < 0 || num_edges < 0) return CGVP_ERR_BAD_ARG;
  if (num_edges == 0) return 0;
  if (num_nodes == 0 || !ca_xyz || !seq_index || !edge_index || !e_s || !e_v || ((uintptr_t)e_s & 15)) return CGVP_ERR_BAD_ARG;
  FeatArgs a{ca_xyz, seq_index, edge_index, num_nodes, num_edges, e_s, e_v};
  const int64_t threads = num_edges * 8;
  hipLaunchKernelGGL(edge_feat_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  const hipError_t err = hipGetLastError();
  return err == hipSuccess ? 0 : (int)err;
}
