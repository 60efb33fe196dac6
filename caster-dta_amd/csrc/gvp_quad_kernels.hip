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
#include <stdint.h>

#include "gvp_internal.h"
#include "gvp_quad.h"

using namespace gq;

namespace {

constexpr int WAVE = 64;
constexpr int WPB = 4;                 // waves per workgroup
constexpr int TPB = WAVE * WPB;
constexpr int TILE = 16;

// Workgroup-cooperative copy of an image slice (multiple of 4 floats) into LDS.
__device__ __forceinline__ void stage_slice(float* lds, const float* __restrict__ src, int nfloats, int tid) {
  const f4* s = reinterpret_cast<const f4*>(src);
  f4* d = reinterpret_cast<f4*>(lds);
  for (int i = tid; i < nfloats / 4; i += TPB) d[i] = s[i];
}

// ------------------------------------------------------------------ prepare
template <int NTN, int NTE>
__global__ void lba_prepare_kernel(const float* __restrict__ P, EncLayout L, int num_convs, float* __restrict__ img) {
  typedef Image<NTN, NTE> IM;
  const int total = IM::total(num_convs);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    float v;
    if (idx < IM::EMB_SIZE) {
      v = idx < IM::EMB_LN ? QNode<NTN>::element(P + L.node_gvp, idx) : P[L.node_ln + (idx - IM::EMB_LN)];
    } else if (idx < IM::head(num_convs)) {
      const int j = idx - IM::EMB_SIZE;
      const int l = j / (IM::CV_SIZE + IM::ND_SIZE);
      int k = j - l * (IM::CV_SIZE + IM::ND_SIZE);
      const float* C = P + L.conv0 + l * L.conv_stride;
      if (k < IM::CV_SIZE) {
        if (k < IM::CV_ELN) v = QEdge<NTE>::element(P + L.edge_gvp, k);
        else if (k < IM::CV_M0) v = P[L.edge_ln + (k - IM::CV_ELN)];
        else if (k < IM::CV_M1) v = QMsg0::element(C + CONV_M0, k - IM::CV_M0);
        else if (k < IM::CV_M2) v = QMsg1::element(C + conv_m1(), k - IM::CV_M1);
        else v = QMsg2::element(C + conv_m2(), k - IM::CV_M2);
      } else {
        k -= IM::CV_SIZE;
        if (k < IM::ND_FF0) v = C[conv_ln0() + k];
        else if (k < IM::ND_FF1) v = QFf0::element(C + conv_ff0(), k - IM::ND_FF0);
        else if (k < IM::ND_LN1) v = QFf1::element(C + conv_ff1(), k - IM::ND_FF1);
        else v = C[conv_ln1() + (k - IM::ND_LN1)];
      }
    } else {
      const int k = idx - IM::head(num_convs);
      v = k < IM::HD_GVP ? P[L.ln_out + k] : QHead::element(P + L.head, k - IM::HD_GVP);
    }
    img[idx] = v;
  }
}

// ------------------------------------------------------------------ embed
struct EmbedQArgs {
  const float* img; const float* x_s; const float* x_v; const int64_t* ntypes; int64_t N; float* h;
};

template <int NTN>
__global__ __launch_bounds__(TPB) void embed_quad_kernel(EmbedQArgs a) {
  typedef Image<NTN, 0> IM;
  typedef QNode<NTN> Q;
  __shared__ __attribute__((aligned(16))) float lds[IM::EMB_SIZE];
  stage_slice(lds, a.img, IM::EMB_SIZE, threadIdx.x);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int64_t n = ((int64_t)blockIdx.x * WPB + w) * TILE + i;
  const bool active = n < a.N;
  float bs[Q::SSTEPS], bv[3][1];
  int type = 0;
#pragma unroll
  for (int s = 0; s < Q::SSTEPS; ++s) {
    const int c = 4 * s + g;
    bs[s] = (active && c < NODE_IN_S) ? a.x_s[n * NODE_IN_S + c] : 0.f;
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) bv[p][0] = (active && g < NODE_IN_V) ? a.x_v[n * 3 * NODE_IN_V + 3 * g + p] : 0.f;
  if (NTN > 0 && active) {
    type = (int)a.ntypes[n];
    type = type < 0 ? 0 : (type >= NTN ? NTN - 1 : type);
  }
  f4 s[1];
  float v[3][1];
  typename Q::Cache c;
  Q::forward(lds + IM::EMB_GVP, lane, type, bs, bv, s, v, c);
  ln_quad<NS, NV>(lds + IM::EMB_LN, lane, s, v);
  if (active) {
    float* row = a.h + n * ROW;
    *reinterpret_cast<f4*>(row + 4 * g) = s[0];
#pragma unroll
    for (int p = 0; p < 3; ++p) row[NS + 3 * g + p] = v[p][0];
  }
}

// ------------------------------------------------------------------ conv
struct ConvQArgs {
  const float* img;   // this layer's conv slice
  const float* h; const float* e_s; const float* e_v; const int64_t* etypes;
  const int32_t* rowptr; const int32_t* eperm; const int32_t* esrc; const int32_t* edst;
  int64_t N; int npw; int mean; float* dh;
};

template <int NTE>
__device__ __forceinline__ void conv_tile(const float* img, const ConvQArgs& a, int32_t p, bool active, int lane,
                                          f4& m_s, float (&m_v)[3]) {
  typedef Image<0, NTE> IM;
  const int g = lane >> 4;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  f4 es0 = zero, es1 = zero, sj = zero, si = zero;
  float ev[3] = {0.f, 0.f, 0.f}, vj[3] = {0.f, 0.f, 0.f}, vi[3] = {0.f, 0.f, 0.f};
  int et = 0;
  if (active) {
    const int32_t eid = a.eperm[p];
    const float* er = a.e_s + (int64_t)eid * EDGE_IN_S;
    es0 = *reinterpret_cast<const f4*>(er + 4 * g);
    es1 = *reinterpret_cast<const f4*>(er + 16 + 4 * g);
    if (g == 0) {
#pragma unroll
      for (int d = 0; d < 3; ++d) ev[d] = a.e_v[(int64_t)eid * 3 + d];
    }
    if (NTE > 0) {
      et = (int)a.etypes[eid];
      et = et < 0 ? 0 : (et >= NTE ? NTE - 1 : et);
    }
    const float* hj = a.h + (int64_t)a.esrc[p] * ROW;
    const float* hi = a.h + (int64_t)a.edst[p] * ROW;
    sj = *reinterpret_cast<const f4*>(hj + 4 * g);
    si = *reinterpret_cast<const f4*>(hi + 4 * g);
#pragma unroll
    for (int d = 0; d < 3; ++d) { vj[d] = hj[NS + 3 * g + d]; vi[d] = hi[NS + 3 * g + d]; }
  }
  // gvp_edge + LayerNorm, in registers (protein_gnn.py:376)
  f4 e_s[2];
  float e_v[3][1];
  {
    float bs[8], bv[3][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bs[r] = es0[r]; bs[4 + r] = es1[r]; }
#pragma unroll
    for (int d = 0; d < 3; ++d) bv[d][0] = ev[d];
    typename QEdge<NTE>::Cache c;
    QEdge<NTE>::forward(img + IM::CV_EDGE, lane, et, bs, bv, e_s, e_v, c);
    ln_quad<ES, EV>(img + IM::CV_ELN, lane, e_s, e_v);
  }
  // message_func.0 on cat((s_j, V_j), edge, (s_i, V_i))   (gvp_layers.py:306)
  f4 s1[1], s2[1];
  float v1[3][1], v2[3][1];
  {
    float bs[16], bv[3][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bs[r] = sj[r]; bs[4 + r] = e_s[0][r]; bs[8 + r] = e_s[1][r]; bs[12 + r] = si[r]; }
#pragma unroll
    for (int d = 0; d < 3; ++d) { bv[d][0] = vj[d]; bv[d][1] = vi[d]; bv[d][2] = e_v[d][0]; }
    QMsg0::Cache c;
    QMsg0::forward(img + IM::CV_M0, lane, 0, bs, bv, s1, v1, c);
  }
  {
    float bs[4], bv[3][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) bs[r] = s1[0][r];
#pragma unroll
    for (int d = 0; d < 3; ++d) bv[d][0] = v1[d][0];
    QMsg1::Cache c;
    QMsg1::forward(img + IM::CV_M1, lane, 0, bs, bv, s2, v2, c);
  }
  {
    float bs[4], bv[3][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) bs[r] = s2[0][r];
#pragma unroll
    for (int d = 0; d < 3; ++d) bv[d][0] = v2[d][0];
    QMsg2::Cache c;
    QMsg2::forward(img + IM::CV_M2, lane, 0, bs, bv, s1, v1, c);
  }
  m_s = s1[0];
#pragma unroll
  for (int d = 0; d < 3; ++d) m_v[d] = v1[d][0];
}

template <int NTE>
__global__ __launch_bounds__(TPB) void conv_quad_kernel(ConvQArgs a) {
  typedef Image<0, NTE> IM;
  constexpr int MSG = TILE * ROW, ACC = WAVE * ROW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* msg = lds + IM::CV_SIZE + w * (MSG + ACC + TILE);
  float* acc = msg + MSG;
  int* dloc = reinterpret_cast<int*>(acc + ACC);
  int* s_tiles = reinterpret_cast<int*>(lds + IM::CV_SIZE + WPB * (MSG + ACC + TILE));   // one dynamic LDS object only
  stage_slice(img, a.img, IM::CV_SIZE, threadIdx.x);

  // this wave's target nodes [n0, n0 + nn) and their (sorted) edges [e0, e1)
  const int64_t n0 = ((int64_t)blockIdx.x * WPB + w) * a.npw;
  const int nn = n0 < a.N ? (int)((a.N - n0 < a.npw) ? (a.N - n0) : a.npw) : 0;
  const int32_t e0 = nn > 0 ? a.rowptr[n0] : 0, e1 = nn > 0 ? a.rowptr[n0 + nn] : 0;
  const int ntiles = (e1 - e0 + TILE - 1) / TILE;
  if (lane == 0) s_tiles[w] = ntiles;
  for (int i = lane; i < nn * ROW; i += WAVE) acc[i] = 0.f;
  __syncthreads();
  int max_tiles = 0;
#pragma unroll
  for (int k = 0; k < WPB; ++k) max_tiles = s_tiles[k] > max_tiles ? s_tiles[k] : max_tiles;

  const int i = lane & 15, g = lane >> 4;
  for (int t = 0; t < max_tiles; ++t) {
    const bool live = t < ntiles;                 // wave-uniform
    const int32_t p = e0 + t * TILE + i;
    const bool active = live && p < e1;
    f4 m_s = {0.f, 0.f, 0.f, 0.f};
    float m_v[3] = {0.f, 0.f, 0.f};
    if (live) conv_tile<NTE>(img, a, p, active, lane, m_s, m_v);
    __syncthreads();                              // previous tile's reduction has drained msg[]
    if (live) {
      *reinterpret_cast<f4*>(msg + i * ROW + 4 * g) = m_s;
#pragma unroll
      for (int d = 0; d < 3; ++d) msg[i * ROW + NS + 3 * g + d] = m_v[d];
      if (g == 0) dloc[i] = active ? (a.edst[p] - (int)n0) : -1;
    }
    __syncthreads();
    // segmented sum over the sorted targets: lane c < 28 owns channel c of this wave's rows
    if (live && lane < ROW) {
      const int cnt = (e1 - (e0 + t * TILE) < TILE) ? (e1 - (e0 + t * TILE)) : TILE;
      int cur = dloc[0];
      float run = 0.f;
      for (int r = 0; r < cnt; ++r) {
        const int d = dloc[r];
        if (d != cur) { acc[cur * ROW + lane] += run; run = 0.f; cur = d; }
        run += msg[r * ROW + lane];
      }
      acc[cur * ROW + lane] += run;
    }
  }
  __syncthreads();
  if (nn > 0) {
    float* out = a.dh + n0 * ROW;
    for (int k = lane; k < nn * ROW; k += WAVE) {
      float v = acc[k];
      if (a.mean) {
        const int nd = k / ROW;
        const int deg = a.rowptr[n0 + nd + 1] - a.rowptr[n0 + nd];
        v = v / (float)(deg > 1 ? deg : 1);
      }
      out[k] = v;
    }
  }
}

// ------------------------------------------------------------------ node update
struct NodeQArgs {
  const float* img_node; const float* img_head;
  const float* h; const float* dh; int64_t N; float* h_out; float* out;
};

template <bool HEAD>
__global__ __launch_bounds__(TPB) void node_quad_kernel(NodeQArgs a) {
  typedef Image<0, 0> IM;
  __shared__ __attribute__((aligned(16))) float lds[IM::ND_SIZE + (HEAD ? IM::HD_SIZE : 0)];
  stage_slice(lds, a.img_node, IM::ND_SIZE, threadIdx.x);
  if (HEAD) stage_slice(lds + IM::ND_SIZE, a.img_head, IM::HD_SIZE, threadIdx.x);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int64_t n = ((int64_t)blockIdx.x * WPB + w) * TILE + i;
  const bool active = n < a.N;
  f4 s[1] = {{0.f, 0.f, 0.f, 0.f}};
  float v[3][1] = {{0.f}, {0.f}, {0.f}};
  if (active) {
    const float* hr = a.h + n * ROW;
    const float* dr = a.dh + n * ROW;
    s[0] = *reinterpret_cast<const f4*>(hr + 4 * g) + *reinterpret_cast<const f4*>(dr + 4 * g);
#pragma unroll
    for (int p = 0; p < 3; ++p) v[p][0] = hr[NS + 3 * g + p] + dr[NS + 3 * g + p];
  }
  ln_quad<NS, NV>(lds + IM::ND_LN0, lane, s, v);
  {
    f4 hs[4], s2[1];
    float hv[3][2], v2[3][1];
    {
      float bs[4], bv[3][1];
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[r] = s[0][r];
#pragma unroll
      for (int p = 0; p < 3; ++p) bv[p][0] = v[p][0];
      QFf0::Cache c;
      QFf0::forward(lds + IM::ND_FF0, lane, 0, bs, bv, hs, hv, c);
    }
    {
      float bs[16], bv[3][2];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[4 * t + r] = hs[t][r];
#pragma unroll
      for (int p = 0; p < 3; ++p) { bv[p][0] = hv[p][0]; bv[p][1] = hv[p][1]; }
      QFf1::Cache c;
      QFf1::forward(lds + IM::ND_FF1, lane, 0, bs, bv, s2, v2, c);
    }
    s[0] += s2[0];
#pragma unroll
    for (int p = 0; p < 3; ++p) v[p][0] += v2[p][0];
  }
  ln_quad<NS, NV>(lds + IM::ND_LN1, lane, s, v);
  if (!HEAD) {
    if (active) {
      float* row = a.h_out + n * ROW;
      *reinterpret_cast<f4*>(row + 4 * g) = s[0];
#pragma unroll
      for (int p = 0; p < 3; ++p) row[NS + 3 * g + p] = v[p][0];
    }
    return;
  }
  const float* hd = lds + IM::ND_SIZE;
  ln_quad<NS, NV>(hd + IM::HD_LN, lane, s, v);
  float bs[4], bv[3][1], dummy[3][1];
#pragma unroll
  for (int r = 0; r < 4; ++r) bs[r] = s[0][r];
#pragma unroll
  for (int p = 0; p < 3; ++p) bv[p][0] = v[p][0];
  f4 o[4];
  QHead::Cache c;
  QHead::forward(hd + IM::HD_GVP, lane, 0, bs, bv, o, dummy, c);
  if (active) {
    float* row = a.out + n * OUT;
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f4*>(row + 16 * t + 4 * g) = o[t];
  }
}

template <int NTN, int NTE>
int prepare_impl(const EncLayout& L, int num_convs, const float* params, float* image, hipStream_t st) {
  const int total = Image<NTN, NTE>::total(num_convs);
  hipLaunchKernelGGL((lba_prepare_kernel<NTN, NTE>), dim3((total + 255) / 256), dim3(256), 0, st, params, L,
                     num_convs, image);
  return 0;
}

template <int NTN, int NTE>
void offsets_impl(int num_convs, QuadOffsets* o) {
  typedef Image<NTN, NTE> IM;
  o->emb = IM::emb(); o->conv0 = IM::conv(0); o->node0 = IM::node(0); o->layer_stride = IM::CV_SIZE + IM::ND_SIZE;
  o->head = IM::head(num_convs); o->total = IM::total(num_convs);
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

int offsets(int nt_node, int nt_edge, int num_convs, QuadOffsets* o) {
#define CALL(A, B) offsets_impl<A, B>(num_convs, o)
  DISPATCH_NT(nt_node, nt_edge, CALL)
#undef CALL
  return 0;
}

int prepare(const EncLayout& L, int num_convs, const float* params, float* image, hipStream_t st) {
#define CALL(A, B) prepare_impl<A, B>(L, num_convs, params, image, st)
  DISPATCH_NT(L.nt_node, L.nt_edge, CALL)
#undef CALL
  return 0;
}

int node_embed(int nt_node, const float* img, const float* x_s, const float* x_v, const int64_t* ntypes,
               int64_t N, float* h, hipStream_t st) {
  EmbedQArgs a{img, x_s, x_v, ntypes, N, h};
  const dim3 grid((unsigned)((N + WPB * TILE - 1) / (WPB * TILE)));
  switch (nt_node) {
    case 0: hipLaunchKernelGGL(embed_quad_kernel<0>, grid, dim3(TPB), 0, st, a); break;
    case 20: hipLaunchKernelGGL(embed_quad_kernel<20>, grid, dim3(TPB), 0, st, a); break;
    case 21: hipLaunchKernelGGL(embed_quad_kernel<21>, grid, dim3(TPB), 0, st, a); break;
    default: return CGVP_ERR_UNSUPPORTED_DIMS;
  }
  return 0;
}

int conv(int nt_edge, const float* img, const float* h, const float* e_s, const float* e_v,
         const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm, const int32_t* esrc,
         const int32_t* edst, int64_t N, int64_t E, int mean, float* dh, hipStream_t st) {
  // target nodes per wave: fill whole 16-edge MFMA tiles, ~48 edges per wave
  int64_t deg = N > 0 ? (E + N - 1) / N : 1;
  if (deg < 1) deg = 1;
  int npw = (int)(48 / deg);
  npw = npw < 1 ? 1 : (npw > WAVE ? WAVE : npw);
  ConvQArgs a{img, h, e_s, e_v, etypes, rowptr, eperm, esrc, edst, N, npw, mean, dh};
  const int64_t groups = (N + npw - 1) / npw;
  const dim3 grid((unsigned)((groups + WPB - 1) / WPB));
  const size_t per_wave = (size_t)(TILE * ROW + WAVE * ROW + TILE) * sizeof(float);
  if (nt_edge == 0) {
    const size_t lds = Image<0, 0>::CV_SIZE * sizeof(float) + WPB * per_wave + WPB * sizeof(int);
    hipLaunchKernelGGL(conv_quad_kernel<0>, grid, dim3(TPB), lds, st, a);
  } else if (nt_edge == 1) {
    const size_t lds = Image<0, 1>::CV_SIZE * sizeof(float) + WPB * per_wave + WPB * sizeof(int);
    hipLaunchKernelGGL(conv_quad_kernel<1>, grid, dim3(TPB), lds, st, a);
  } else {
    return CGVP_ERR_UNSUPPORTED_DIMS;
  }
  return 0;
}

int node_update(const float* img_node, const float* img_head, const float* h, const float* dh, int64_t N,
                int with_head, float* h_out, float* out, hipStream_t st) {
  NodeQArgs a{img_node, img_head, h, dh, N, h_out, out};
  const dim3 grid((unsigned)((N + WPB * TILE - 1) / (WPB * TILE)));
  if (with_head) hipLaunchKernelGGL(node_quad_kernel<true>, grid, dim3(TPB), 0, st, a);
  else hipLaunchKernelGGL(node_quad_kernel<false>, grid, dim3(TPB), 0, st, a);
  return 0;
}

}  // namespace quad
