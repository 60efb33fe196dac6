// Internal declarations shared by the translation units of libcaster_gvp.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/caster_gvp.h"
#include "gvp_math.h"
#include "gvp_rng.h"

// Tile policy index the launchers in namespace quad take (their `bf16` parameter): activation storage type x layer kind
// (cgvp_dims.storage / cgvp_dims.layer_kind; gvp_quad.h LayerKind).  bf16 storage exists for CASTER-DTA's layers only.
constexpr int POLICY_F32 = 0, POLICY_BF16 = 1, POLICY_GVPDEF = 2, POLICY_LINEAR = 3;

// Float offsets of the per-kernel slices inside the fragment image.
struct QuadOffsets {
  int emb, conv0, node0, layer_stride, head, total;
  int embT, convT0, nodeT0, layerT_stride, headT;     // transposed slices (backward)
};

#ifndef CGVP_BWD_MAX_GRID
#define CGVP_BWD_MAX_GRID 240       // of 256 CUs: the rest is left to the drug-side backward running beside it
#endif
constexpr int kBwdMaxGrid = CGVP_BWD_MAX_GRID;    // persistent workgroups of the backward kernels (one per CU) = slab rows per stage

// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, set ONCE per kernel and device instead of before every launch
// (hipFuncSetAttribute is a host call of tens of microseconds; a backward pass made 13 of them).  One static flag word per
// expansion site, one bit per device: idempotent, benign process state (the attribute itself is process state of the runtime).
#include <atomic>
#define CGVP_SET_DYN_LDS_ONCE(FN, BYTES)                                                                                      \
  do {                                                                                                                        \
    static std::atomic<uint64_t> cgvp_done_{0};                                                                               \
    int cgvp_dev_ = 0;                                                                                                        \
    (void)hipGetDevice(&cgvp_dev_);                                                                                           \
    const uint64_t cgvp_bit_ = 1ull << (cgvp_dev_ & 63);                                                                      \
    if (!(cgvp_done_.load(std::memory_order_acquire) & cgvp_bit_)) {                                                          \
      if (hipError_t cgvp_err_ = hipFuncSetAttribute(reinterpret_cast<const void*>(FN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES))) \
        return (int)cgvp_err_;                                                                                                \
      cgvp_done_.fetch_or(cgvp_bit_, std::memory_order_release);                                                              \
    }                                                                                                                         \
  } while (0)

namespace quad {
int offsets(int nt_node, int nt_edge, int num_convs, QuadOffsets* o);
int prepare(const gvp::EncLayout& L, int num_convs, int packed_bf16, const float* params, float* image, hipStream_t st);
int node_embed(int nt_node, const float* img, const float* x_s, const float* x_v, const int64_t* ntypes,
               int64_t N, float* h, unsigned long long* rng_state, unsigned long long* rng_out, int bf16, hipStream_t st);
int pass_begin(const gvp::EncLayout& L, int num_convs, int bf16, const float* params, float* image, const float* x_s,
               const float* x_v, const int64_t* ntypes, int64_t N, float* h, unsigned long long* rng_state,
               unsigned long long* rng_out, const int64_t* edge_index, int64_t E, int32_t* counters, hipStream_t st);
int conv(int nt_edge, const float* img, const float* h, const float* e_s, const float* e_v,
         const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm, const int32_t* esrc,
         const int32_t* edst, int64_t N, int64_t E, int mean, float* dh, int fuse, const float* img_node,
         const float* img_head, float* h_out, float* out, const float* mask0, const float* mask1, gvp::RngArgs rng,
         const float* e_in, float* e_out, int bf16, hipStream_t st);
int node_update(const float* img_node, const float* img_head, const float* h, const float* dh, int64_t N,
                int with_head, float* h_out, float* out, const float* mask0, const float* mask1, gvp::RngArgs rng,
                int bf16, hipStream_t st);

// ---- backward (gvp_quad_bwd_kernels.hip).  Weight gradients are written as one
// partial block per workgroup into `slab` ([grid][block floats]); `grid` returns
// the number of rows to reduce.
int node_update_bwd(const float* img_node, const float* imgT_node, const float* h, const float* dh,
                    const float* mask0, const float* mask1, gvp::RngArgs rng, const float* g_up0, const float* g_up1,
                    const float* g_up2, int64_t N, float* g_dh, float* g_h, float* zero_rows, float* slab, int* grid,
                    int bf16, hipStream_t st);
int head_bwd(const float* img_head, const float* imgT_head, const float* h_out, const float* g_out, int64_t N,
             float* g_h_out, float* slab, int* grid, int bf16, hipStream_t st);
// head_bwd + node_update_bwd of the last layer as ONE launch (g_h_out doubles as the node stage's upstream gradient)
int node_head_bwd(const float* img_head, const float* imgT_head, const float* h_out, const float* g_out, float* g_h_out,
                  float* head_slab, const float* img_node, const float* imgT_node, const float* h, const float* dh,
                  const float* mask0, const float* mask1, gvp::RngArgs rng, int64_t N, float* g_dh, float* g_h,
                  float* zero_rows, float* slab, int* grid, int bf16, hipStream_t st);
int conv_bwd(int nt_edge, const float* img, const float* imgT, const float* h, const float* e_emb,
             const int32_t* rowptr, const int32_t* esrc, const int32_t* edst, int64_t N, int64_t E, int mean,
             const float* g_dh, float* g_src, float* g_dst, float* g_e, float* slab, int* grid, int bf16, hipStream_t st);
// backward of gvp_edge + LayerNorm from the summed d(edge embedding) of the conv layers (weight gradients only)
int edge_embed_bwd(int nt_edge, const float* img, const float* imgT, const float* e_s, const float* e_v,
                   const int64_t* etypes, const int32_t* eperm, int64_t E, const float* const* g_e, int n_g,
                   float* g_e_s, float* g_e_v, float* slab, int* grid, int bf16, hipStream_t st);
constexpr int kEdgeRow = 36;              // floats per edge of the stored edge embedding: [e_s 32 | e_v 3 | pad]
int node_embed_bwd(int nt_node, const float* img, const float* imgT, const float* x_s, const float* x_v,
                   const int64_t* ntypes, int64_t N, const float* g_up0, const float* g_up1, const float* g_up2,
                   float* g_x_s, float* g_x_v, float* slab, int* grid, int bf16, hipStream_t st);
// GINE layer backward on 16-atom MFMA tiles (gine_quad_kernels.hip): one slab row per workgroup
// (`rows` x `row_len` floats, state_dict order of the layer).
constexpr int kGineBwdMaxGrid = 256;      // upper bound (workspace sizing)
constexpr int kGineBwdDefaultGrid = 16;   // the CUs the protein backward (kBwdMaxGrid workgroups, one per CU) leaves free
int gine_bwd(int cin, int chid, int cout, int nt, int net, int ed, const float* x, const int64_t* ntypes,
             const float* eattr, const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm,
             const int32_t* esrc, const int32_t* edst, int64_t N, const cgvp_gine_w* w, float slope,
             const float* mask, gvp::RngArgs rng, const float* g_out, float* g_x, float* slab, int max_workgroups, int* rows,
             int* row_len, hipStream_t st, const float* agg_in = nullptr, const uint16_t* pos_in = nullptr);
int gine_fwd(int cin, int chid, int cout, int nt, int net, int ed, const float* x, const int64_t* ntypes,
             const float* eattr, const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm,
             const int32_t* esrc, const int32_t* edst, int64_t N, const cgvp_gine_w* w, float slope,
             const float* mask, gvp::RngArgs rng, float* out, float* agg_out, uint16_t* pos_out, hipStream_t st);
int reduce_slab(const float* slab, int rows, int stride, int col0, int len, float* dst, hipStream_t st);
int reduce_segments(const cgvp_segment* segs, int nsegs, float* grad_params, hipStream_t st, int overwrite = 0);
int bwd_block_sizes(int nt_node, int nt_edge, int* emb, int* conv_edge, int* conv_total, int* node, int* head);
// gvp_kernels.hip
void zero_words(void* p, size_t words, hipStream_t s);     // the library's own zero fill (never hipMemsetAsync: see gvp_kernels.hip)
int csr_build(const int64_t* edge_index, int64_t N, int64_t E, int32_t* rowptr, int32_t* eperm, int32_t* esrc,
              int32_t* edst, int32_t* work, int work_is_zero, int32_t* ids_scratch, unsigned long long* rng_state,
              unsigned long long* rng_out, hipStream_t stream);
}  // namespace quad
