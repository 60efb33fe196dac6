// gvp_quad_kernels.hip -- the MFMA ("quad layout") kernels of the LBA protein
// encoder: 16 items per wave tile, 4 lanes per item, every channel GEMM on
// v_mfma_f32_16x16x4_f32 with weight fragments staged in LDS (see gvp_quad.h).
//
//   lba_prepare_kernel   arena -> fragment image (once per parameter update)
//   embed_quad_kernel    gvp_node: GVP + LayerNorm per residue
//   conv_quad_kernel     per edge: gvp_edge + LayerNorm + 3 message GVPs, then a
//                        segmented sum over the dst-sorted edges of the wave's own
//                        target nodes (LDS, no atomics) -> dh
//   node_quad_kernel     residual + LN, 2-GVP feed-forward, residual + LN (+ head)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "gvp_internal.h"
#include "gvp_quad.h"

using namespace gq;

namespace {

// Diagnostic build only (-DCGVP_STAMPS): per-wave s_memtime stamps into a debug
// buffer that nothing else reads; the production library contains none of it.
#ifdef CGVP_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
#define STAMP(slot)                                                                          \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (g_stamp_buf && (threadIdx.x & 63) == 0)                                              \
      g_stamp_buf[((size_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * 16 + (slot)] = t_;        \
  } while (0)
// wall-clock stamp (s_memrealtime: 100 MHz, one counter for the whole device -- s_memtime counters are per XCD)
#define STAMP_REAL(slot)                                                                     \
  do {                                                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
    if (g_stamp_buf && (threadIdx.x & 63) == 0)                                              \
      g_stamp_buf[((size_t)blockIdx.x * WPB + (threadIdx.x >> 6)) * 16 + (slot)] = t_;        \
  } while (0)
#else
#define STAMP(slot) do {} while (0)
#define STAMP_REAL(slot) do {} while (0)
#endif

constexpr int WAVE = 64;
constexpr int WPB = 4;                 // waves per workgroup
constexpr int TPB = WAVE * WPB;
constexpr int gq_fwd_tpb = TPB;
constexpr int TILE = 16;

// Workgroup-cooperative copy of an image slice (NFLOATS, multiple of 4) into LDS:
// every global load is issued before the first LDS store, so the slice costs one
// memory latency instead of one per iteration.
template <int NFLOATS>
__device__ __forceinline__ void stage_slice(float* lds, const float* __restrict__ src, int tid) {
  static_assert(NFLOATS % 4 == 0, "image slices are whole float4s");
  constexpr int NF4 = NFLOATS / 4, IT = (NF4 + TPB - 1) / TPB;
  const f4* s = reinterpret_cast<const f4*>(src);
  f4* d = reinterpret_cast<f4*>(lds);
  f4 v[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int idx = tid + k * TPB;
    if (idx < NF4) v[k] = s[idx];
  }
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int idx = tid + k * TPB;
    if (idx < NF4) d[idx] = v[k];
  }
}

// Up to three slices (from three places of the image) into consecutive LDS: ALL global loads are issued before the
// first LDS store, so the copies cost one memory latency together instead of one each (the fused conv + node update
// (+ head) kernel stages 68-74 KB per workgroup).  A size of 0 skips a slice.
template <int N0, int N1, int N2, int TPB = gq_fwd_tpb>
__device__ __forceinline__ void stage_slices(float* lds, const float* __restrict__ s0, const float* __restrict__ s1,
                                             const float* __restrict__ s2, int tid) {
  static_assert(N0 % 4 == 0 && N1 % 4 == 0 && N2 % 4 == 0, "image slices are whole float4s");
  constexpr int F0 = N0 / 4, F1 = N1 / 4, F2 = N2 / 4;
  constexpr int I0 = (F0 + TPB - 1) / TPB, I1 = (F1 + TPB - 1) / TPB, I2 = (F2 + TPB - 1) / TPB;
  f4* d = reinterpret_cast<f4*>(lds);
  f4 v0[I0 > 0 ? I0 : 1], v1[I1 > 0 ? I1 : 1], v2[I2 > 0 ? I2 : 1];
#pragma unroll
  for (int k = 0; k < I0; ++k) { const int idx = tid + k * TPB; if (idx < F0) v0[k] = reinterpret_cast<const f4*>(s0)[idx]; }
#pragma unroll
  for (int k = 0; k < I1; ++k) { const int idx = tid + k * TPB; if (idx < F1) v1[k] = reinterpret_cast<const f4*>(s1)[idx]; }
#pragma unroll
  for (int k = 0; k < I2; ++k) { const int idx = tid + k * TPB; if (idx < F2) v2[k] = reinterpret_cast<const f4*>(s2)[idx]; }
#pragma unroll
  for (int k = 0; k < I0; ++k) { const int idx = tid + k * TPB; if (idx < F0) d[idx] = v0[k]; }
#pragma unroll
  for (int k = 0; k < I1; ++k) { const int idx = tid + k * TPB; if (idx < F1) d[F0 + idx] = v1[k]; }
#pragma unroll
  for (int k = 0; k < I2; ++k) { const int idx = tid + k * TPB; if (idx < F2) d[F0 + F1 + idx] = v2[k]; }
}

// ------------------------------------------------------------------ prepare
template <int NTN, int NTE>
__device__ __forceinline__ float image_element(const float* __restrict__ P, const EncLayout& L, int num_convs, int packed, int idx) {
  typedef Image<NTN, NTE> IM;

  float v;
  if (idx < IM::EMB_SIZE) {
    v = idx < IM::EMB_LN ? QNode<NTN>::element(P + L.node_gvp, idx, packed != 0) : P[L.node_ln + (idx - IM::EMB_LN)];
  } else if (idx < IM::head(num_convs)) {
    const int j = idx - IM::EMB_SIZE;
    const int l = j / (IM::CV_SIZE + IM::ND_SIZE);
    int k = j - l * (IM::CV_SIZE + IM::ND_SIZE);
    const float* C = P + L.conv0 + l * L.conv_stride;
    if (k < IM::CV_SIZE) {
      if (k < IM::CV_ELN) v = QEdge<NTE>::element(P + L.edge_gvp, k, packed != 0);
      else if (k < IM::CV_M0) v = P[L.edge_ln + (k - IM::CV_ELN)];
      else if (k < IM::CV_M1) v = QMsg0::element(C + CONV_M0, k - IM::CV_M0, packed != 0);
      else if (k < IM::CV_M2) v = QMsg1::element(C + conv_m1(), k - IM::CV_M1, packed != 0);
      else v = QMsg2::element(C + conv_m2(), k - IM::CV_M2, packed != 0);
    } else {
      k -= IM::CV_SIZE;
      if (k < IM::ND_FF0) v = C[conv_ln0() + k];
      else if (k < IM::ND_FF1) v = QFf0::element(C + conv_ff0(), k - IM::ND_FF0, packed != 0);
      else if (k < IM::ND_LN1) v = QFf1::element(C + conv_ff1(), k - IM::ND_FF1, packed != 0);
      else v = C[conv_ln1() + (k - IM::ND_LN1)];
    }
  } else if (idx < IM::fwd_total(num_convs)) {
    const int k = idx - IM::head(num_convs);
    v = k < IM::HD_GVP ? P[L.ln_out + k] : QHead::element(P + L.head, k - IM::HD_GVP, packed != 0);
  } else if (idx < IM::convT(num_convs, 0)) {
    v = QNode<NTN>::element_t(P + L.node_gvp, idx - IM::embT(num_convs), packed != 0);
  } else if (idx < IM::headT(num_convs)) {
    const int j = idx - IM::convT(num_convs, 0);
    const int l = j / (IM::TC_SIZE + IM::TN_SIZE);
    int k = j - l * (IM::TC_SIZE + IM::TN_SIZE);
    const float* C = P + L.conv0 + l * L.conv_stride;
    if (k < IM::TC_M0) v = QEdge<NTE>::element_t(P + L.edge_gvp, k, packed != 0);
    else if (k < IM::TC_M1) v = QMsg0::element_t(C + CONV_M0, k - IM::TC_M0, packed != 0);
    else if (k < IM::TC_M2) v = QMsg1::element_t(C + conv_m1(), k - IM::TC_M1, packed != 0);
    else if (k < IM::TC_SIZE) v = QMsg2::element_t(C + conv_m2(), k - IM::TC_M2, packed != 0);
    else if (k < IM::TC_SIZE + IM::TN_FF1) v = QFf0::element_t(C + conv_ff0(), k - IM::TC_SIZE, packed != 0);
    else v = QFf1::element_t(C + conv_ff1(), k - IM::TC_SIZE - IM::TN_FF1, packed != 0);
  } else {
    v = QHead::element_t(P + L.head, idx - IM::headT(num_convs), packed != 0);
  }
    return v;
}

template <int NTN, int NTE>
__global__ void lba_prepare_kernel(const float* __restrict__ P, EncLayout L, int num_convs, int packed, float* __restrict__ img) {
  const int total = Image<NTN, NTE>::total(num_convs);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x)
    img[idx] = image_element<NTN, NTE>(P, L, num_convs, packed, idx);
}

// ------------------------------------------------------------------ embed
struct EmbedQArgs {
  const float* img; const float* x_s; const float* x_v; const int64_t* ntypes; int64_t N; float* h;
  // dropout generator hand-off (optional): the FIRST kernel of an encoder pass advances the persistent {seed, offset}
  // and leaves this pass's pair in rng_out, which the later kernels of the pass (and its backward) read
  unsigned long long* rng_state; unsigned long long* rng_out;
};

// the embedding of the 16 residues of this wave's tile; `lds` = the embed slice of the image, staged by the caller
template <int NTN, typename ST>
__device__ __forceinline__ void embed_tile(const EmbedQArgs& a, const float* lds, int64_t block) {
  typedef Image<NTN, 0> IM;
  typedef QNode<NTN> Q;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int64_t n = (block * WPB + w) * TILE + i;
  const bool active = n < a.N;
  float bs[1][Q::SSTEPS], bv[1][3][1];
  int type[1] = {0};
#pragma unroll
  for (int s = 0; s < Q::SSTEPS; ++s) {
    const int c = 4 * s + g;
    bs[0][s] = (active && c < NODE_IN_S) ? Io<ST>::ld(a.x_s, n * NODE_IN_S + c) : 0.f;
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) bv[0][p][0] = (active && g < NODE_IN_V) ? Io<ST>::ld(a.x_v, n * 3 * NODE_IN_V + 3 * g + p) : 0.f;
  if (NTN > 0 && active) {
    type[0] = (int)a.ntypes[n];
    type[0] = type[0] < 0 ? 0 : (type[0] >= NTN ? NTN - 1 : type[0]);
  }
  f4 s[1][1];
  float v[1][3][1];
  typename Q::Cache c[1];
  Q::template forward<1, Io<ST>::BF>(lds + IM::EMB_GVP, lane, type, bs, bv, s, v, c);
  ln_quad<NS, NV>(lds + IM::EMB_LN, lane, s[0], v[0]);
  if (active) {
    Io<ST>::st4(a.h, n * ROW + 4 * g, s[0][0]);
#pragma unroll
    for (int p = 0; p < 3; ++p) Io<ST>::st(a.h, n * ROW + NS + 3 * g + p, v[0][p][0]);
  }
}
__device__ __forceinline__ void rng_handoff(const EmbedQArgs& a) {
  if (a.rng_state && blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned long long off = a.rng_state[1] + 1;
    a.rng_state[1] = off;
    a.rng_out[0] = a.rng_state[0];
    a.rng_out[1] = off;
  }
}

template <int NTN, typename ST>
__global__ __launch_bounds__(TPB) void embed_quad_kernel(EmbedQArgs a) {
  typedef Image<NTN, 0> IM;
  __shared__ __attribute__((aligned(16))) float lds[IM::EMB_SIZE];
  stage_slice<IM::EMB_SIZE>(lds, a.img, threadIdx.x);
  rng_handoff(a);
  __syncthreads();
  embed_tile<NTN, ST>(a, lds, blockIdx.x);
}

// First launch of a protein TRAINING pass (cgvp_lba_pass_begin): three independent pieces of work share one grid --
//   blocks [0, embed_blocks)            gvp_node on 64 residues each; they build their 1.2k-float slice of the fragment
//                                       image straight from the arena, so they do not wait for ...
//   next prep_blocks blocks             ... the fragment image of the current weights (cgvp_lba_prepare's work), and
//   the remaining count_blocks blocks   the per-target edge counts of the CSR build (cgvp_csr_from_coo's first launch).
// Two launches less at the head of every step (~5 us each at 64 x 300 residues: these kernels are launch latency).
struct PassBeginArgs {
  EmbedQArgs e;
  const float* params; EncLayout L; int num_convs; int packed; float* image;
  const int64_t* ei; int64_t E; int64_t Ncount; int32_t* counters;
  int embed_blocks, prep_blocks;
};
template <int NTN, int NTE, typename ST>
__global__ __launch_bounds__(TPB) void pass_begin_kernel(PassBeginArgs a) {
  typedef Image<NTN, NTE> IM;
  __shared__ __attribute__((aligned(16))) float lds[IM::EMB_SIZE];
  const int b = blockIdx.x;
  if (b >= a.embed_blocks + a.prep_blocks) {                       // CSR count (malformed edges are dropped, never fault)
    const int64_t e = (int64_t)(b - a.embed_blocks - a.prep_blocks) * TPB + threadIdx.x;
    if (e < a.E) {
      const int64_t s = a.ei[e], d = a.ei[a.E + e];
      if (s >= 0 && s < a.Ncount && d >= 0 && d < a.Ncount) atomicAdd(&a.counters[d], 1);
    }
    return;
  }
  if (b >= a.embed_blocks) {                                        // fragment image
    const int total = IM::total(a.num_convs);
    for (int idx = (b - a.embed_blocks) * TPB + threadIdx.x; idx < total; idx += a.prep_blocks * TPB)
      a.image[idx] = image_element<NTN, NTE>(a.params, a.L, a.num_convs, a.packed, idx);
    return;
  }
  for (int k = threadIdx.x; k < IM::EMB_SIZE; k += TPB) lds[k] = image_element<NTN, NTE>(a.params, a.L, a.num_convs, a.packed, k);
  rng_handoff(a.e);
  __syncthreads();
  embed_tile<NTN, ST>(a.e, lds, b);
}

// ------------------------------------------------------------------ node update
// mask0 / mask1: optional dropout masks [N][20] = 16 scalar-channel + 4 vector-channel
// factors (0 or 1/(1-p); a vector channel's xyz share one factor, gvp_layers.py:187-198),
// applied to dh before the first residual and to the feed-forward output before the second.
struct NodeQArgs {
  const float* img_node; const float* img_head;
  const float* h; const float* dh; int64_t N; float* h_out; float* out;
  const float* mask0; const float* mask1;
  gvp::RngArgs rng;            // in-kernel dropout (used where the mask pointers are NULL); stream = 2 * layer
};
constexpr int MROW = NS + NV;

// Dropout factors of node n for lane (i, g): scalar channels 4g..4g+3 and vector channel g of mask `which`
// (0 = dropout[0] on dh, 1 = dropout[1] on the feed-forward output): read from the given mask, or generated.
__device__ __forceinline__ void node_dropout(const float* mask, const gvp::RngArgs& rng, int which, int64_t n, int g,
                                             f4& ms, float& mv) {
  if (mask) {
    ms = *reinterpret_cast<const f4*>(mask + n * MROW + 4 * g);
    mv = mask[n * MROW + NS + g];
  } else if (rng.seed) {
    const unsigned long long seed = rng.seed[0], off = rng.seed[1];
    float fs[4];
    gvp::dropout_row20(seed, off, rng.stream + which, n, g, rng.p, fs, mv);
    ms = f4{fs[0], fs[1], fs[2], fs[3]};
  }
}

// Rest of GVPConvLayer.forward for one tile of 16 residues, from s/v = h + mask0 * dh (lane (i, g)
// holds scalars 4g..4g+3 and vector channel g): LN0, feed-forward GVPs, residual, LN1 -> h_out
// (optional with the head), then gvp_norm_before_scalar + gvp_to_scalar -> out.
template <bool HEAD, typename ST>
__device__ __forceinline__ void node_tile(const float* nd, const float* hd, int lane, bool active, int64_t n,
                                          f4 (&s)[1], float (&v)[3][1], f4 m1s, float m1v, float* h_out,
                                          float* out) {
  typedef Image<0, 0> IM;
  const int g = lane >> 4;
  const int zt[1] = {0};
  ln_quad<NS, NV>(nd + IM::ND_LN0, lane, s, v);
  {
    f4 hs[1][4], s2[1][1];
    float hv[1][3][2], v2[1][3][1];
    {
      float bs[1][4], bv[1][3][1];
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[0][r] = s[0][r];
#pragma unroll
      for (int p = 0; p < 3; ++p) bv[0][p][0] = v[p][0];
      typename Ff0<ST>::Cache c[1];
      Ff0<ST>::template forward<1, Io<ST>::BF>(nd + IM::ND_FF0, lane, zt, bs, bv, hs, hv, c);
    }
    {
      float bs[1][16], bv[1][3][2];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[0][4 * t + r] = hs[0][t][r];
#pragma unroll
      for (int p = 0; p < 3; ++p) { bv[0][p][0] = hv[0][p][0]; bv[0][p][1] = hv[0][p][1]; }
      typename Ff1<ST>::Cache c[1];
      Ff1<ST>::template forward<1, Io<ST>::BF>(nd + IM::ND_FF1, lane, zt, bs, bv, s2, v2, c);
    }
    s[0] += s2[0][0] * m1s;
#pragma unroll
    for (int p = 0; p < 3; ++p) v[p][0] += v2[0][p][0] * m1v;
  }
  ln_quad<NS, NV>(nd + IM::ND_LN1, lane, s, v);
  if (active && (!HEAD || h_out)) {          // with the head h_out is optional (training saves it for the backward)
    Io<ST>::st4(h_out, n * ROW + 4 * g, s[0]);
#pragma unroll
    for (int p = 0; p < 3; ++p) Io<ST>::st(h_out, n * ROW + NS + 3 * g + p, v[p][0]);
  }
  if (!HEAD) return;
  ln_quad<NS, NV>(hd + IM::HD_LN, lane, s, v);
  float bs[1][4], bv[1][3][1], dummy[1][3][1];
#pragma unroll
  for (int r = 0; r < 4; ++r) bs[0][r] = s[0][r];
#pragma unroll
  for (int p = 0; p < 3; ++p) bv[0][p][0] = v[p][0];
  f4 o[1][4];
  QHead::Cache c[1];
  QHead::template forward<1, Io<ST>::BF>(hd + IM::HD_GVP, lane, zt, bs, bv, o, dummy, c);
  if (active) {
#pragma unroll
    for (int t = 0; t < 4; ++t) Io<ST>::st4(out, n * OUT + 16 * t + 4 * g, o[0][t]);
  }
}

// s/v = h[n] + mask0 * dh_row for lane (i, g); `dr` + `dr_off` is the node's [28] aggregated message: an fp32 row of
// the wave's LDS accumulator (DT = float, fused layer kernel) or a row of the dh buffer in HBM (DT = ST)
template <typename ST, typename DT>
__device__ __forceinline__ void node_inputs(const NodeQArgs& a, int lane, bool active, int64_t n, const float* dr,
                                            int64_t dr_off, f4 (&s)[1], float (&v)[3][1], f4& m1s, float& m1v) {
  const int g = lane >> 4;
  s[0] = f4{0.f, 0.f, 0.f, 0.f};
  v[0][0] = v[1][0] = v[2][0] = 0.f;
  m1s = f4{1.f, 1.f, 1.f, 1.f};
  m1v = 1.f;
  if (!active) return;
  f4 ds = Io<DT>::ld4(dr, dr_off + 4 * g);
  float dv[3] = {Io<DT>::ld(dr, dr_off + NS + 3 * g), Io<DT>::ld(dr, dr_off + NS + 3 * g + 1), Io<DT>::ld(dr, dr_off + NS + 3 * g + 2)};
  {
    f4 m0s = {1.f, 1.f, 1.f, 1.f};
    float m0v = 1.f;
    node_dropout(a.mask0, a.rng, 0, n, g, m0s, m0v);
    node_dropout(a.mask1, a.rng, 1, n, g, m1s, m1v);
    ds *= m0s;
#pragma unroll
    for (int p = 0; p < 3; ++p) dv[p] *= m0v;
  }
  s[0] = Io<ST>::ld4(a.h, n * ROW + 4 * g) + ds;
#pragma unroll
  for (int p = 0; p < 3; ++p) v[p][0] = Io<ST>::ld(a.h, n * ROW + NS + 3 * g + p) + dv[p];
}

template <bool HEAD, typename ST>
__global__ __launch_bounds__(TPB) void node_quad_kernel(NodeQArgs a) {
  typedef Image<0, 0> IM;
  __shared__ __attribute__((aligned(16))) float lds[IM::ND_SIZE + (HEAD ? IM::HD_SIZE : 0)];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15;
  const int64_t n = ((int64_t)blockIdx.x * WPB + w) * TILE + i;
  const bool active = n < a.N;
  f4 s[1], m1s;
  float v[3][1], m1v;
  node_inputs<ST, ST>(a, lane, active, n, a.dh, n * ROW, s, v, m1s, m1v);     // row loads fly while the image is staged
  stage_slices<IM::ND_SIZE, (HEAD ? IM::HD_SIZE : 0), 0>(lds, a.img_node, a.img_head, nullptr, threadIdx.x);
  __syncthreads();
  node_tile<HEAD, ST>(lds, lds + IM::ND_SIZE, lane, active, n, s, v, m1s, m1v, a.h_out, a.out);
}

// ------------------------------------------------------------------ conv
struct ConvQArgs {
  const float* img;   // this layer's conv slice
  const float* h; const float* e_s; const float* e_v; const int64_t* etypes;
  const int32_t* rowptr; const int32_t* eperm; const int32_t* esrc; const int32_t* edst;
  int64_t N; int npw; int mean; float* dh;
  NodeQArgs node;     // FUSE > 0: the node update of the same layer runs on the wave's own targets (h = this->h)
  // EDGE EMBEDDING STORE (sorted-edge order, EROW floats per edge = [e_s 32 | e_v 3 | pad]): gvp_edge + LayerNorm
  // (protein_gnn.py:376) does not depend on the layer, so the first conv layer may write it (e_out) and later layers
  // and the backward kernels read it (e_in) instead of re-deriving it from the raw features
  const float* e_in; float* e_out; int64_t E;      // e_out holds E + 1 rows: row E is a spare that masked-off lanes write
};
constexpr int EROW = 36;
// EMODE of the conv kernel: 0 = derive the edge embedding from the raw features (and keep it in registers),
// 1 = derive it and also store it to e_out, 2 = read it from e_in (raw features untouched)

constexpr int CTN = 2;                 // edge tiles processed in lockstep per pass (32 edges)

// Per-lane inputs of CTN tiles of 16 sorted edges (lane = (edge i, group g)).
struct ConvIn {
  f4 es0[CTN], es1[CTN], sj[CTN], si[CTN];
  float ev[CTN][3], vj[CTN][3], vi[CTN][3];
  int et[CTN];
  int32_t dst[CTN];
  bool active[CTN];
};

// Gather the inputs of the pass starting at sorted-edge position `base`:
// CSR tables -> raw edge features (original edge order) + source / target rows.
template <int NTE, int EMODE, typename ST>
__device__ __forceinline__ void conv_gather(const ConvQArgs& a, int32_t base, int32_t e1, int lane, ConvIn& in) {
  const int i = lane & 15, g = lane >> 4;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  int32_t eid[CTN], src[CTN];
#pragma unroll
  for (int j = 0; j < CTN; ++j) {                    // hop 1: all index loads in flight together
    const int32_t p = base + j * TILE + i;
    in.active[j] = p < e1;
    in.dst[j] = in.active[j] ? a.edst[p] : -1;
    eid[j] = (EMODE != 2 && in.active[j]) ? a.eperm[p] : 0;
    src[j] = in.active[j] ? a.esrc[p] : 0;
  }
#pragma unroll
  for (int j = 0; j < CTN; ++j) {                    // hop 2: rows
    in.es0[j] = in.es1[j] = in.sj[j] = in.si[j] = zero;
    in.et[j] = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) in.ev[j][d] = in.vj[j][d] = in.vi[j][d] = 0.f;
    if (in.active[j]) {
      if (EMODE == 2) {                                // stored embedding, sequential in sorted-edge order
        const int64_t er = (int64_t)(base + j * TILE + i) * EROW;
        in.es0[j] = Io<ST>::ld4(a.e_in, er + 4 * g);
        in.es1[j] = Io<ST>::ld4(a.e_in, er + 16 + 4 * g);
        if (g == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) in.ev[j][d] = Io<ST>::ld(a.e_in, er + ES + d);
        }
      } else {
        const int64_t er = (int64_t)eid[j] * EDGE_IN_S;
        in.es0[j] = Io<ST>::ld4(a.e_s, er + 4 * g);
        in.es1[j] = Io<ST>::ld4(a.e_s, er + 16 + 4 * g);
        if (g == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) in.ev[j][d] = Io<ST>::ld(a.e_v, (int64_t)eid[j] * 3 + d);
        }
        if (NTE > 0) {
          in.et[j] = (int)a.etypes[eid[j]];
          in.et[j] = in.et[j] < 0 ? 0 : (in.et[j] >= NTE ? NTE - 1 : in.et[j]);
        }
      }
      const int64_t hj = (int64_t)src[j] * ROW, hi = (int64_t)in.dst[j] * ROW;
      in.sj[j] = Io<ST>::ld4(a.h, hj + 4 * g);
      in.si[j] = Io<ST>::ld4(a.h, hi + 4 * g);
#pragma unroll
      for (int d = 0; d < 3; ++d) { in.vj[j][d] = Io<ST>::ld(a.h, hj + NS + 3 * g + d); in.vi[j][d] = Io<ST>::ld(a.h, hi + NS + 3 * g + d); }
    }
  }
}

// Message of CTN tiles: raw edge features -> gvp_edge + LayerNorm (in registers)
// -> cat with source / target node rows -> 3 message GVPs.
template <int NTE, int EMODE, typename ST>
__device__ __forceinline__ void conv_tiles(const float* img, const ConvIn& in, int lane, float* e_out, int64_t e_spare,
                                           int32_t base, f4 (&m_s)[CTN], float (&m_v)[CTN][3]) {
  typedef Image<0, NTE> IM;
  const f4 (&es0)[CTN] = in.es0; const f4 (&es1)[CTN] = in.es1; const f4 (&sj)[CTN] = in.sj; const f4 (&si)[CTN] = in.si;
  const float (&ev)[CTN][3] = in.ev; const float (&vj)[CTN][3] = in.vj; const float (&vi)[CTN][3] = in.vi;
  const int (&et)[CTN] = in.et;
  const int zero_t[CTN] = {};
  // gvp_edge + LayerNorm, in registers (protein_gnn.py:376)
  f4 e_s[CTN][2];
  float e_v[CTN][3][1];
  if (EMODE == 2) {
#pragma unroll
    for (int j = 0; j < CTN; ++j) {
      e_s[j][0] = es0[j]; e_s[j][1] = es1[j];
#pragma unroll
      for (int d = 0; d < 3; ++d) e_v[j][d][0] = ev[j][d];
    }
  } else {
    float bs[CTN][8], bv[CTN][3][1];
#pragma unroll
    for (int j = 0; j < CTN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { bs[j][r] = es0[j][r]; bs[j][4 + r] = es1[j][r]; }
#pragma unroll
      for (int d = 0; d < 3; ++d) bv[j][d][0] = ev[j][d];
    }
    typename QEdge<NTE>::Cache c[CTN];
    QEdge<NTE>::template forward<CTN, Io<ST>::BF>(img + IM::CV_EDGE, lane, et, bs, bv, e_s, e_v, c);
#pragma unroll
    for (int j = 0; j < CTN; ++j) ln_quad<ES, EV>(img + IM::CV_ELN, lane, e_s[j], e_v[j]);
    if (EMODE == 1) {
      // continue with the values the store will hold (bf16 storage): later layers and the backward read exactly these
#pragma unroll
      for (int j = 0; j < CTN; ++j) {
        e_s[j][0] = Io<ST>::rt4(e_s[j][0]); e_s[j][1] = Io<ST>::rt4(e_s[j][1]);
#pragma unroll
        for (int d = 0; d < 3; ++d) e_v[j][d][0] = Io<ST>::rt(e_v[j][d][0]);
      }
      const int i = lane & 15, g = lane >> 4;
#pragma unroll
      for (int j = 0; j < CTN; ++j) {
        // branch-free: lanes past the wave's last edge write the store's spare row (row `e_spare` = E).  With the
        // stores under `if (active)` hipcc duplicated the whole message body of this kernel (412 instead of 206 MFMAs).
        // (every write to the spare row is a zero, so the row stays deterministic)
        const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const bool act = in.active[j], actv = act && g == 0;
        const int64_t p = base + j * TILE + i;
        const int64_t er = (act ? p : e_spare) * EROW, erv = (actv ? p : e_spare) * EROW;
        Io<ST>::st4(e_out, er + 4 * g, act ? e_s[j][0] : zero4);
        Io<ST>::st4(e_out, er + 16 + 4 * g, act ? e_s[j][1] : zero4);
        Io<ST>::st4(e_out, erv + ES, actv ? f4{e_v[j][0][0], e_v[j][1][0], e_v[j][2][0], 0.f} : zero4);   // + zero pad
      }
    }
  }
  STAMP(5);
  // message_func.0 on cat((s_j, V_j), edge, (s_i, V_i))   (gvp_layers.py:306)
  f4 s1[CTN][1], s2[CTN][1];
  float v1[CTN][3][1], v2[CTN][3][1];
  {
    float bs[CTN][16], bv[CTN][3][3];
#pragma unroll
    for (int j = 0; j < CTN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bs[j][r] = sj[j][r]; bs[j][4 + r] = e_s[j][0][r]; bs[j][8 + r] = e_s[j][1][r]; bs[j][12 + r] = si[j][r];
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) { bv[j][d][0] = vj[j][d]; bv[j][d][1] = vi[j][d]; bv[j][d][2] = e_v[j][d][0]; }
    }
    typename Msg0<ST>::Cache c[CTN];
    Msg0<ST>::template forward<CTN, Io<ST>::BF>(img + IM::CV_M0, lane, zero_t, bs, bv, s1, v1, c);
  }
  {
    float bs[CTN][4], bv[CTN][3][1];
#pragma unroll
    for (int j = 0; j < CTN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[j][r] = s1[j][0][r];
#pragma unroll
      for (int d = 0; d < 3; ++d) bv[j][d][0] = v1[j][d][0];
    }
    typename Msg1<ST>::Cache c[CTN];
    Msg1<ST>::template forward<CTN, Io<ST>::BF>(img + IM::CV_M1, lane, zero_t, bs, bv, s2, v2, c);
  }
  {
    float bs[CTN][4], bv[CTN][3][1];
#pragma unroll
    for (int j = 0; j < CTN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[j][r] = s2[j][0][r];
#pragma unroll
      for (int d = 0; d < 3; ++d) bv[j][d][0] = v2[j][d][0];
    }
    typename Msg2<ST>::Cache c[CTN];
    Msg2<ST>::template forward<CTN, Io<ST>::BF>(img + IM::CV_M2, lane, zero_t, bs, bv, s1, v1, c);
  }
#pragma unroll
  for (int j = 0; j < CTN; ++j) {
    m_s[j] = s1[j][0];
#pragma unroll
    for (int d = 0; d < 3; ++d) m_v[j][d] = v1[j][d][0];
  }
}

// Waves are independent after the image is staged: each owns `npw` consecutive
// target nodes, walks their dst-sorted edges 32 at a time, reduces every tile
// with an in-register segmented scan across the 16 edge lanes (DPP row shifts;
// no LDS traffic, no barrier) and lets the last lane of each segment add the
// segment total into the wave's private LDS accumulator.  Ownership makes the
// result independent of scheduling: no atomics on HBM, bitwise reproducible.
// FUSE: 0 = conv only; 1 = + node update; 2 = + node update with the output head
// CW = waves per workgroup: 8 = ONE workgroup per CU, so a launch of <= 240 workgroups leaves whole CUs free for the
// drug encoder's kernels on the side stream (with 4-wave workgroups the dispatcher spread 480 of them over all 256 CUs
// and the drug forward found no CU to start on until both conv launches had drained: 54 us late in the round-4 step
// trace); 4 = two workgroups per CU, the shape of larger launches (conv_fwd_waves below picks).
template <int NTE, int FUSE, int EMODE, typename ST, int CW = 4>
__global__ __launch_bounds__(WAVE * CW) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_quad_kernel(ConvQArgs a) {
  constexpr int WPB = CW, TPB = WAVE * CW;
  typedef Image<0, NTE> IM;
  typedef Image<0, 0> IMN;
  constexpr int ACC = WAVE * ROW;
  constexpr int ND_FLOATS = FUSE == 0 ? 0 : IMN::ND_SIZE + (FUSE == 2 ? IMN::HD_SIZE : 0);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* nd_img = lds + IM::CV_SIZE;
  float* acc = nd_img + ND_FLOATS + w * ACC;
  STAMP(0);
  STAMP_REAL(14);
  const int64_t n0 = ((int64_t)blockIdx.x * WPB + w) * a.npw;
  const int nn = n0 < a.N ? (int)((a.N - n0 < a.npw) ? (a.N - n0) : a.npw) : 0;
  const int32_t e0 = nn > 0 ? a.rowptr[n0] : 0, e1 = nn > 0 ? a.rowptr[n0 + nn] : 0;
  // The first pass's gathers (3 dependent hops) are issued BEFORE the image is
  // staged so that their latency hides behind the staging traffic and barrier.
  ConvIn in;
  conv_gather<NTE, EMODE, ST>(a, e0, e1, lane, in);
  if (EMODE == 1 && blockIdx.x == 0 && threadIdx.x < EROW / 4)        // the store's spare row is always all zeros
    Io<ST>::st4(a.e_out, a.E * EROW + 4 * threadIdx.x, f4{0.f, 0.f, 0.f, 0.f});
  stage_slices<IM::CV_SIZE, (FUSE > 0 ? IMN::ND_SIZE : 0), (FUSE == 2 ? IMN::HD_SIZE : 0), TPB>(
      img, a.img, a.node.img_node, a.node.img_head, threadIdx.x);      // nd_img = img + CV_SIZE, the head slice behind it
  for (int k = lane; k < nn * ROW; k += WAVE) acc[k] = 0.f;
  STAMP(1);
  __syncthreads();                                  // image staged, accumulators cleared
  STAMP(2);

  const int i = lane & 15, g = lane >> 4;
  for (int32_t base = e0; base < e1; base += CTN * TILE) {
    if (base != e0) conv_gather<NTE, EMODE, ST>(a, base, e1, lane, in);
    const int32_t (&dst)[CTN] = in.dst;
    const bool (&active)[CTN] = in.active;
    f4 m_s[CTN];
    float m_v[CTN][3];
    STAMP(3);
    conv_tiles<NTE, EMODE, ST>(img, in, lane, a.e_out, a.E, base, m_s, m_v);
    STAMP(6);
#pragma unroll
    for (int j = 0; j < CTN; ++j) {
      float x[7] = {m_s[j][0], m_s[j][1], m_s[j][2], m_s[j][3], m_v[j][0], m_v[j][1], m_v[j][2]};
      seg_scan16<7>(dst[j], x);
      // lane i closes a segment when its right neighbour has another target (or is past the end)
      const int nxt = __builtin_amdgcn_update_dpp(-1, dst[j], 0x100 | 1 /* row_shl:1 */, 0xf, 0xf, false);
      if (active[j] && (i == TILE - 1 || nxt != dst[j])) {
        // wave-private rows, and within a tile every (row, channel) has exactly one closing lane: plain
        // read-add-write (LDS float atomics retire ~1 lane per 3 cycles per CU)
        float* row = acc + (dst[j] - (int)n0) * ROW;
        f4* qs = reinterpret_cast<f4*>(row + 4 * g);
        *qs = *qs + f4{x[0], x[1], x[2], x[3]};
#pragma unroll
        for (int d = 0; d < 3; ++d) row[NS + 3 * g + d] += x[4 + d];
      }
      if (CTN > 1) {                     // the next lockstep tile may continue this tile's last target
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
    }
  }
  STAMP(7);
  __syncthreads();
  STAMP(8);
  if (nn > 0) {
    for (int k = lane; k < nn * ROW; k += WAVE) {
      float v = acc[k];
      if (a.mean) {
        const int nd = k / ROW;
        const int deg = a.rowptr[n0 + nd + 1] - a.rowptr[n0 + nd];
        v = v / (float)(deg > 1 ? deg : 1);
        if (FUSE > 0) acc[k] = v;
      }
      if (a.dh) {                                             // kept for the backward pass; inference passes no dh buffer when fused
        Io<ST>::st(a.dh, n0 * ROW + k, v);
        if (FUSE > 0) acc[k] = Io<ST>::rt(v);                 // the fused node update continues with what the backward will re-read
      }
    }
  }
  STAMP(9);
  if (FUSE > 0) {
    // the node update of the wave's own targets, 16 at a time, straight from the LDS rows
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int t0 = 0; t0 < nn; t0 += TILE) {
      const int k = t0 + i;
      const bool act = k < nn;
      const int64_t n = n0 + k;
      f4 s[1], m1s;
      float v[3][1], m1v;
      node_inputs<ST, float>(a.node, lane, act, n, acc, (int64_t)(act ? k : 0) * ROW, s, v, m1s, m1v);
      node_tile<FUSE == 2, ST>(nd_img, nd_img + IMN::ND_SIZE, lane, act, n, s, v, m1s, m1v, a.node.h_out, a.node.out);
    }
  }
  STAMP(10);
  STAMP_REAL(15);
}

template <int NTN, int NTE>
int prepare_impl(const EncLayout& L, int num_convs, int packed, const float* params, float* image, hipStream_t st) {
  const int total = Image<NTN, NTE>::total(num_convs);
  hipLaunchKernelGGL((lba_prepare_kernel<NTN, NTE>), dim3((total + 255) / 256), dim3(256), 0, st, params, L,
                     num_convs, packed, image);
  return 0;
}

template <int NTN, int NTE>
void offsets_impl(int num_convs, QuadOffsets* o) {
  typedef Image<NTN, NTE> IM;
  o->emb = IM::emb(); o->conv0 = IM::conv(0); o->node0 = IM::node(0); o->layer_stride = IM::CV_SIZE + IM::ND_SIZE;
  o->head = IM::head(num_convs); o->total = IM::total(num_convs);
  o->embT = IM::embT(num_convs); o->convT0 = IM::convT(num_convs, 0); o->nodeT0 = IM::nodeT(num_convs, 0);
  o->layerT_stride = IM::TC_SIZE + IM::TN_SIZE; o->headT = IM::headT(num_convs);
}

#define DISPATCH_NT(NTN_, NTE_, CALL)                       \
  if (NTN_ == 0 && NTE_ == 0) { CALL(0, 0); }               \
  else if (NTN_ == 20 && NTE_ == 1) { CALL(20, 1); }        \
  else if (NTN_ == 21 && NTE_ == 1) { CALL(21, 1); }        \
  else if (NTN_ == 20 && NTE_ == 0) { CALL(20, 0); }        \
  else if (NTN_ == 0 && NTE_ == 1) { CALL(0, 1); }          \
  else return CGVP_ERR_UNSUPPORTED_DIMS;

}  // namespace

namespace quad {

#ifdef CGVP_STAMPS
extern "C" int cgvp_debug_set_stamp_buffer(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
}
#endif

int offsets(int nt_node, int nt_edge, int num_convs, QuadOffsets* o) {
#define CALL(A, B) offsets_impl<A, B>(num_convs, o)
  DISPATCH_NT(nt_node, nt_edge, CALL)
#undef CALL
  return 0;
}

int prepare(const EncLayout& L, int num_convs, int packed, const float* params, float* image, hipStream_t st) {
#define CALL(A, B) prepare_impl<A, B>(L, num_convs, packed, params, image, st)
  DISPATCH_NT(L.nt_node, L.nt_edge, CALL)
#undef CALL
  return 0;
}

int node_embed(int nt_node, const float* img, const float* x_s, const float* x_v, const int64_t* ntypes,
               int64_t N, float* h, unsigned long long* rng_state, unsigned long long* rng_out, int bf16, hipStream_t st) {
  EmbedQArgs a{img, x_s, x_v, ntypes, N, h, rng_state, rng_out};
  const dim3 grid((unsigned)((N + WPB * TILE - 1) / (WPB * TILE)));
#define EMB_LAUNCH(NT) { if (bf16) hipLaunchKernelGGL((embed_quad_kernel<NT, bf16s>), grid, dim3(TPB), 0, st, a); \
                         else hipLaunchKernelGGL((embed_quad_kernel<NT, float>), grid, dim3(TPB), 0, st, a); }
  switch (nt_node) {
    case 0: EMB_LAUNCH(0) break;
    case 20: EMB_LAUNCH(20) break;
    case 21: EMB_LAUNCH(21) break;
    default: return CGVP_ERR_UNSUPPORTED_DIMS;
  }
#undef EMB_LAUNCH
  return 0;
}

template <int NTN, int NTE>
int pass_begin_impl(const PassBeginArgs& a, int blocks, int bf16, hipStream_t st) {
  if (bf16) hipLaunchKernelGGL((pass_begin_kernel<NTN, NTE, bf16s>), dim3(blocks), dim3(TPB), 0, st, a);
  else hipLaunchKernelGGL((pass_begin_kernel<NTN, NTE, float>), dim3(blocks), dim3(TPB), 0, st, a);
  return 0;
}
int pass_begin(const EncLayout& L, int num_convs, int bf16, const float* params, float* image, const float* x_s,
               const float* x_v, const int64_t* ntypes, int64_t N, float* h, unsigned long long* rng_state,
               unsigned long long* rng_out, const int64_t* edge_index, int64_t E, int32_t* counters, hipStream_t st) {
  QuadOffsets o;
  if (int rc = offsets(L.nt_node, L.nt_edge, num_convs, &o)) return rc;
  PassBeginArgs a{};
  a.e = EmbedQArgs{nullptr, x_s, x_v, ntypes, N, h, rng_state, rng_out};
  a.params = params; a.L = L; a.num_convs = num_convs; a.packed = bf16 ? 1 : 0; a.image = image;
  a.ei = edge_index; a.E = (edge_index && counters) ? E : 0; a.Ncount = N; a.counters = counters;
  a.embed_blocks = (int)((N + WPB * TILE - 1) / (WPB * TILE));
  a.prep_blocks = (o.total + TPB - 1) / TPB;
  const int blocks = a.embed_blocks + a.prep_blocks + (int)((a.E + TPB - 1) / TPB);
#define CALL(A, B) pass_begin_impl<A, B>(a, blocks, bf16, st)
  DISPATCH_NT(L.nt_node, L.nt_edge, CALL)
#undef CALL
  return 0;
}

// 8-wave workgroups only while they all fit one-per-CU in a single round (their 100 KB of LDS allows no second
// workgroup on a CU: kiba_b32's 288 of them ran as 256 + a 32-workgroup second round, 0.275 -> 0.288 ms per step);
// CGVP_CONV_FWD_WAVES=4|8 in the environment (read once) forces a shape: A/B knob.
inline int conv_fwd_waves(int64_t groups) {
  static const int forced = [] { const char* e = getenv("CGVP_CONV_FWD_WAVES"); return (e && e[0] == '4') ? 4 : ((e && e[0] == '8') ? 8 : 0); }();
  if (forced) return forced;
  return groups <= 8 * (int64_t)kBwdMaxGrid ? 8 : 4;
}
template <int NTE, int FUSE, int EMODE, typename ST, int CW>
int conv_launch_w(const ConvQArgs& a, int64_t groups, hipStream_t st) {
  const size_t lds = (size_t)(Image<0, NTE>::CV_SIZE + (FUSE == 0 ? 0 : Image<0, 0>::ND_SIZE + (FUSE == 2 ? Image<0, 0>::HD_SIZE : 0)) +
                              CW * WAVE * ROW) * sizeof(float);
  if (lds > 64 * 1024) CGVP_SET_DYN_LDS_ONCE((conv_quad_kernel<NTE, FUSE, EMODE, ST, CW>), lds);      // once per kernel and device; an error is returned
  hipLaunchKernelGGL((conv_quad_kernel<NTE, FUSE, EMODE, ST, CW>), dim3((unsigned)((groups + CW - 1) / CW)), dim3(WAVE * CW), lds, st, a);
  return 0;
}
template <int NTE, int FUSE, int EMODE, typename ST>
int conv_launch_e(const ConvQArgs& a, int64_t groups, hipStream_t st) {
  if (conv_fwd_waves(groups) == 4) return conv_launch_w<NTE, FUSE, EMODE, ST, 4>(a, groups, st);
  return conv_launch_w<NTE, FUSE, EMODE, ST, 8>(a, groups, st);
}
template <int NTE, int FUSE, typename ST>
int conv_launch_s(const ConvQArgs& a, int64_t grid, hipStream_t st) {
  if (a.e_in) return conv_launch_e<NTE, FUSE, 2, ST>(a, grid, st);
  if (a.e_out) return conv_launch_e<NTE, FUSE, 1, ST>(a, grid, st);
  return conv_launch_e<NTE, FUSE, 0, ST>(a, grid, st);
}
// `bf16` below is the tile policy index (gvp_internal.h, POLICY_*): 0 float, 1 bf16s, 2 f32_gvpdef, 3 f32_linear
template <int NTE, int FUSE>
int conv_launch(const ConvQArgs& a, int64_t grid, int bf16, hipStream_t st) {
  if (bf16 >= POLICY_GVPDEF) {      // the other layer kinds: stand-alone layers (stored edge embedding, no type columns, two launches)
    if constexpr (NTE == 0 && FUSE == 0) {
      if (!a.e_in) return CGVP_ERR_UNSUPPORTED_DIMS;
      if (bf16 == POLICY_GVPDEF) return conv_launch_e<0, 0, 2, f32_gvpdef>(a, grid, st);
      return conv_launch_e<0, 0, 2, f32_linear>(a, grid, st);
    }
    return CGVP_ERR_UNSUPPORTED_DIMS;
  }
  if (bf16) return conv_launch_s<NTE, FUSE, bf16s>(a, grid, st);
  return conv_launch_s<NTE, FUSE, float>(a, grid, st);
}

// fuse: 0 = conv only (dh required); 1 / 2 = the layer's node update (2: with the output head) in the same launch
int conv(int nt_edge, const float* img, const float* h, const float* e_s, const float* e_v,
         const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm, const int32_t* esrc,
         const int32_t* edst, int64_t N, int64_t E, int mean, float* dh, int fuse, const float* img_node,
         const float* img_head, float* h_out, float* out, const float* mask0, const float* mask1, gvp::RngArgs rng,
         const float* e_in, float* e_out, int bf16, hipStream_t st) {
  // target nodes per wave: one pass of CTN lockstep 16-edge tiles (~30 edges) per wave
  int64_t deg = N > 0 ? (E + N - 1) / N : 1;
  if (deg < 1) deg = 1;
  int npw = (int)((CTN * TILE - 2) / deg);
  npw = npw < 1 ? 1 : (npw > WAVE ? WAVE : npw);
  // Dense graphs (conv-only launches): one ~20-edge target per wave left every pass 62 % full and made 16k workgroups
  // stage the 37 KB image for 4 targets each (long_graph_x64).  A wave walks its targets' edges 32 at a time whatever the
  // target boundaries, so give it more targets once there are more than a few rounds of resident waves: the last pass of
  // a wave is the only partial one.
  if (fuse == 0) {
    const int64_t max_waves = 4 * 2048;             // four rounds of 2 workgroups x 4 waves on 256 CUs
    if ((N + npw - 1) / npw > max_waves) {
      int64_t m = (N + max_waves - 1) / max_waves;
      npw = (int)(m > WAVE ? WAVE : m);
    }
  }
  ConvQArgs a{img, h, e_s, e_v, etypes, rowptr, eperm, esrc, edst, N, npw, mean, dh,
              NodeQArgs{img_node, img_head, h, nullptr, N, h_out, out, mask0, mask1, rng}, e_in, e_out, E};
  const int64_t groups = (N + npw - 1) / npw;
  const int64_t grid = groups;                 // waves; conv_launch_w turns them into workgroups
  if (nt_edge != 0 && nt_edge != 1) return CGVP_ERR_UNSUPPORTED_DIMS;
  if (fuse < 0 || fuse > 2) return CGVP_ERR_BAD_ARG;
  if (nt_edge == 0) {
    if (fuse == 0) return conv_launch<0, 0>(a, grid, bf16, st);
    if (fuse == 1) return conv_launch<0, 1>(a, grid, bf16, st);
    return conv_launch<0, 2>(a, grid, bf16, st);
  }
  if (fuse == 0) return conv_launch<1, 0>(a, grid, bf16, st);
  if (fuse == 1) return conv_launch<1, 1>(a, grid, bf16, st);
  return conv_launch<1, 2>(a, grid, bf16, st);
}

int node_update(const float* img_node, const float* img_head, const float* h, const float* dh, int64_t N,
                int with_head, float* h_out, float* out, const float* mask0, const float* mask1, gvp::RngArgs rng,
                int bf16, hipStream_t st) {
  NodeQArgs a{img_node, img_head, h, dh, N, h_out, out, mask0, mask1, rng};
  const dim3 grid((unsigned)((N + WPB * TILE - 1) / (WPB * TILE)));
  if (bf16 >= POLICY_GVPDEF) {
    if (with_head) return CGVP_ERR_UNSUPPORTED_DIMS;
    if (bf16 == POLICY_GVPDEF) hipLaunchKernelGGL((node_quad_kernel<false, f32_gvpdef>), grid, dim3(TPB), 0, st, a);
    else hipLaunchKernelGGL((node_quad_kernel<false, f32_linear>), grid, dim3(TPB), 0, st, a);
  } else if (bf16) {
    if (with_head) hipLaunchKernelGGL((node_quad_kernel<true, bf16s>), grid, dim3(TPB), 0, st, a);
    else hipLaunchKernelGGL((node_quad_kernel<false, bf16s>), grid, dim3(TPB), 0, st, a);
  } else {
    if (with_head) hipLaunchKernelGGL((node_quad_kernel<true, float>), grid, dim3(TPB), 0, st, a);
    else hipLaunchKernelGGL((node_quad_kernel<false, float>), grid, dim3(TPB), 0, st, a);
  }
  return 0;
}

}  // namespace quad
