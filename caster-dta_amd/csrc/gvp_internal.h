// Internal declarations shared by the translation units of libcaster_gvp.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/caster_gvp.h"
#include "gvp_math.h"

// Float offsets of the per-kernel slices inside the fragment image.
struct QuadOffsets {
  int emb, conv0, node0, layer_stride, head, total;
};

namespace quad {
int offsets(int nt_node, int nt_edge, int num_convs, QuadOffsets* o);
int prepare(const gvp::EncLayout& L, int num_convs, const float* params, float* image, hipStream_t st);
int node_embed(int nt_node, const float* img, const float* x_s, const float* x_v, const int64_t* ntypes,
               int64_t N, float* h, hipStream_t st);
int conv(int nt_edge, const float* img, const float* h, const float* e_s, const float* e_v,
         const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm, const int32_t* esrc,
         const int32_t* edst, int64_t N, int64_t E, int mean, float* dh, hipStream_t st);
int node_update(const float* img_node, const float* img_head, const float* h, const float* dh, int64_t N,
                int with_head, float* h_out, float* out, hipStream_t st);
}  // namespace quad
