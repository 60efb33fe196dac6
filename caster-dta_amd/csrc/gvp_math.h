// Per-item GVP arithmetic shared by every kernel in gvp_kernels.hip.
//
// One "item" is one residue or one residue-residue edge; a GPU lane owns one
// item and keeps all of its channels in registers (every loop below has
// compile-time bounds and is fully unrolled).  Weights are read through
// wave-uniform pointers with compile-time offsets, so on gfx950 they become
// scalar loads (s_load_dword*) and feed v_fma_f32 as SGPR operands.
//
// The same templates compile with a host C++ compiler (tests/host_math builds
// them with g++ to unit-test this exact arithmetic on CPU, where no GPU
// exists).  That build is test scaffolding only; the product has no CPU path.
//
// Reference semantics restated here (file:line in /root/reference):
//   _norm_no_nan      models/gvp_layers.py:79-86
//   GVP.forward       models/gvp_layers.py:142-175  (vector_gate=True, vector_act=None)
//   LayerNorm.forward models/gvp_layers.py:231-242
//   type one-hot cat  models/protein_gnn.py:139-152 (one-hot columns come FIRST)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define GVP_HD __host__ __device__ __forceinline__
#else
#define GVP_HD inline
#endif

namespace gvp {

constexpr int nz(int x) { return x > 0 ? x : 1; }

// Weight pointer.  The argument structs carry plain `const float*`; device code
// re-types them into the CONSTANT address space (addrspace 4: ordinary global
// memory, promised read-only for the launch) right before reading, which is what
// makes hipcc lower every wave-uniform weight read to s_load_dword* feeding
// v_fmac as an SGPR operand, instead of a 64-lane global_load per weight.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) float* wptr;
#else
typedef const float* wptr;
#endif
GVP_HD wptr cw(const float* p) { return (wptr)p; }

// Parameters live in ONE contiguous fp32 arena laid out in the reference's
// state_dict order (zero-size dummy_params skipped), so a kernel needs a single
// base pointer and every weight sits at a compile-time offset from its block.
// One GVP block (gvp_layers.py:129-137 registration order), row-major [out][in]:
//   wh.weight [H][VI] | ws.weight [SO][nt+SI+H] | ws.bias [SO] |
//   wv.weight [VO][H] | wsv.weight [VO][SO] | wsv.bias [VO]      (last three iff VO > 0)
// ws has nt one-hot type columns FIRST, then SI scalar features, then H norms.
template <int SI, int VI, int SO, int VO, int H>
struct GvpLayout {
  static constexpr int ws(int) { return H * VI; }
  static constexpr int bs(int nt) { return H * VI + SO * (nt + SI + H); }
  static constexpr int wv(int nt) { return bs(nt) + SO; }
  static constexpr int wsv(int nt) { return wv(nt) + VO * H; }
  static constexpr int bsv(int nt) { return wsv(nt) + VO * SO; }
  static constexpr int size(int nt) { return bsv(nt) + VO; }
};
// One tuple-LayerNorm block: scalar_norm.weight [S] | scalar_norm.bias [S].
template <int S>
struct LnLayout {
  static constexpr int size() { return 2 * S; }
};

GVP_HD float f_exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __expf(x);
#else
  return expf(x);
#endif
}
GVP_HD float f_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
GVP_HD float f_sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}
GVP_HD float f_rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}
GVP_HD float f_sigmoid(float x) { return f_rcp(1.0f + f_exp(-x)); }
GVP_HD float f_max(float a, float b) { return a > b ? a : b; }

constexpr float kNormEps = 1e-8f;  // clamp on the SQUARED norm (gvp_layers.py:85)
constexpr float kLnEps = 1e-5f;    // nn.LayerNorm default eps

// Intermediates of one GVP evaluation, kept for the backward pass.
template <int SO, int VO, int H>
struct GvpCache {
  float vh[nz(H)][3];   // wh . V
  float vn[nz(H)];      // clamped norms of vh
  float sp[SO];         // pre-activation scalars (the gate reads these)
  float vp[nz(VO)][3];  // wv . vh before gating
  float sg[nz(VO)];     // sigmoid(gate)
};

// GVP forward for one item.  NT one-hot type columns precede the scalar inputs
// in ws; `type` selects the column (ignored when NT == 0).  RELU applies F.relu to the scalar output; the vector output is
// gated by sigmoid(wsv . s_pre + b) (vector_gate=True, vector_act=None).
template <int NT, int SI, int VI, int SO, int VO, int H, bool RELU>
GVP_HD void gvp_forward(const float* block, int type, const float (&s)[SI],
                        const float (&v)[nz(VI)][3], float (&so)[SO], float (&vo)[nz(VO)][3],
                        GvpCache<SO, VO, H>& c) {
  constexpr int nt = NT;
  constexpr int K = nt + SI + H;
  typedef GvpLayout<SI, VI, SO, VO, H> L;
  const wptr Wh = cw(block), Ws = Wh + L::ws(nt), Bs = Wh + L::bs(nt), Wv = Wh + L::wv(nt),
             Wsv = Wh + L::wsv(nt), Bsv = Wh + L::bsv(nt);
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const float w = Wh[h * VI + i];
      a0 = fmaf(w, v[i][0], a0);
      a1 = fmaf(w, v[i][1], a1);
      a2 = fmaf(w, v[i][2], a2);
    }
    c.vh[h][0] = a0; c.vh[h][1] = a1; c.vh[h][2] = a2;
    c.vn[h] = f_sqrt(f_max(fmaf(a0, a0, fmaf(a1, a1, a2 * a2)), kNormEps));
  }
#pragma unroll
  for (int o = 0; o < SO; ++o) {
    const wptr row = Ws + o * K;
    float a = Bs[o];
    if (NT > 0) a += row[type];
#pragma unroll
    for (int k = 0; k < SI; ++k) a = fmaf(row[nt + k], s[k], a);
#pragma unroll
    for (int h = 0; h < H; ++h) a = fmaf(row[nt + SI + h], c.vn[h], a);
    c.sp[o] = a;
  }
#pragma unroll
  for (int o = 0; o < VO; ++o) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float w = Wv[o * H + h];
      a0 = fmaf(w, c.vh[h][0], a0);
      a1 = fmaf(w, c.vh[h][1], a1);
      a2 = fmaf(w, c.vh[h][2], a2);
    }
    c.vp[o][0] = a0; c.vp[o][1] = a1; c.vp[o][2] = a2;
    float g = Bsv[o];
#pragma unroll
    for (int k = 0; k < SO; ++k) g = fmaf(Wsv[o * SO + k], c.sp[k], g);
    const float sg = f_sigmoid(g);
    c.sg[o] = sg;
    vo[o][0] = a0 * sg; vo[o][1] = a1 * sg; vo[o][2] = a2 * sg;
  }
#pragma unroll
  for (int o = 0; o < SO; ++o) so[o] = RELU ? f_max(c.sp[o], 0.f) : c.sp[o];
}

// Statistics of one tuple LayerNorm, kept for the backward pass.
struct LnCache {
  float rstd;   // 1/sqrt(var + eps) of the scalar channels
  float rvn;    // 1/sqrt(mean_c clamp(|v_c|^2))
};

// Tuple LayerNorm for one item, in place: nn.LayerNorm on the S scalars and
// V / sqrt(mean over channels of clamp(|V_c|^2, 1e-8)) on the vectors.
template <int S, int NV>
GVP_HD void ln_forward(const float* block, float (&s)[S], float (&v)[nz(NV)][3], LnCache& c) {
  const wptr G = cw(block), B = G + S;
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < S; ++k) mean += s[k];
  mean *= (1.0f / S);
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < S; ++k) { const float d = s[k] - mean; var = fmaf(d, d, var); }
  var *= (1.0f / S);
  const float rstd = f_rsqrt(var + kLnEps);
  c.rstd = rstd;
#pragma unroll
  for (int k = 0; k < S; ++k) s[k] = fmaf((s[k] - mean) * rstd, G[k], B[k]);
  if (NV > 0) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      acc += f_max(fmaf(v[i][0], v[i][0], fmaf(v[i][1], v[i][1], v[i][2] * v[i][2])), kNormEps);
    const float rvn = f_rsqrt(acc * (1.0f / nz(NV)));
    c.rvn = rvn;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i][0] *= rvn; v[i][1] *= rvn; v[i][2] *= rvn; }
  }
}

// ---------------------------------------------------------------------------
// CASTER-DTA(s,v) dimensions of the LBA protein encoder
// (train_model.py:276-293, pretrained_model_downstream/model_kwargs.json).
constexpr int NODE_IN_S = 17, NODE_IN_V = 3;   // residue features
constexpr int EDGE_IN_S = 32, EDGE_IN_V = 1;   // RBF + positional, direction
constexpr int NS = 16, NV = 4;                 // hidden node dims
constexpr int ES = 32, EV = 1;                 // hidden edge dims
constexpr int ROW = NS + 3 * NV;               // merged node row: [s(16) | v(4x3)] = 28 floats
constexpr int FS = 4 * NS, FV = 2 * NV;        // feed-forward hidden dims (gvp_layers.py:359)
constexpr int OUT = 64;                        // residue embedding width
constexpr int MS = 2 * NS + ES, MV = 2 * NV + EV;  // message input dims (64, 9)

typedef GvpLayout<NODE_IN_S, NODE_IN_V, NS, NV, NV> LNodeGvp;   // gvp_node.0
typedef GvpLayout<EDGE_IN_S, EDGE_IN_V, ES, EV, EV> LEdgeGvp;   // gvp_edge.0
typedef GvpLayout<MS, MV, NS, NV, MV> LMsg0;                    // message_func.0
typedef GvpLayout<NS, NV, NS, NV, NV> LMsg;                     // message_func.1/2
typedef GvpLayout<NS, NV, FS, FV, FV> LFf0;                     // ff_func.0
typedef GvpLayout<FS, FV, NS, NV, FV> LFf1;                     // ff_func.1
typedef GvpLayout<NS, NV, OUT, 0, NV> LHead;                    // gvp_to_scalar

// Arena offsets (in floats) of the blocks of VectorProteinGNN_LBAModel, in
// state_dict order: gvp_node.{0,1}, gvp_edge.{0,1}, then per conv layer
// conv.message_func.{0,1,2}, norm.{0,1}, ff_func.{0,1}, and finally
// gvp_norm_before_scalar, gvp_to_scalar.
struct EncLayout {
  int nt_node, nt_edge;      // one-hot widths in front of the scalar inputs
  int node_gvp, node_ln, edge_gvp, edge_ln;
  int conv0;                 // first conv layer block
  int conv_stride;           // floats per conv layer block
  int ln_out, head;
  int total;
};
GVP_HD EncLayout make_layout(int nt_node, int nt_edge, int num_convs) {
  EncLayout L;
  L.nt_node = nt_node; L.nt_edge = nt_edge;
  int o = 0;
  L.node_gvp = o; o += LNodeGvp::size(nt_node);
  L.node_ln = o;  o += LnLayout<NS>::size();
  L.edge_gvp = o; o += LEdgeGvp::size(nt_edge);
  L.edge_ln = o;  o += LnLayout<ES>::size();
  L.conv0 = o;
  L.conv_stride = LMsg0::size(0) + 2 * LMsg::size(0) + 2 * LnLayout<NS>::size() + LFf0::size(0) + LFf1::size(0);
  o += num_convs * L.conv_stride;
  L.ln_out = o; o += LnLayout<NS>::size();
  L.head = o;   o += LHead::size(0);
  L.total = o;
  return L;
}
// Offsets inside one conv layer block.
constexpr int CONV_M0 = 0;
constexpr int conv_m1() { return LMsg0::size(0); }
constexpr int conv_m2() { return conv_m1() + LMsg::size(0); }
constexpr int conv_ln0() { return conv_m2() + LMsg::size(0); }
constexpr int conv_ln1() { return conv_ln0() + LnLayout<NS>::size(); }
constexpr int conv_ff0() { return conv_ln1() + LnLayout<NS>::size(); }
constexpr int conv_ff1() { return conv_ff0() + LFf0::size(0); }

// Residue embedding: GVP (17+types, 3)->(16,4) without activations + LayerNorm
// (protein_gnn.py:325-329, :375).  Output is the merged 28-float node row.
template <int NTN>
GVP_HD void node_embed_item(const float* P, const EncLayout& L, int type,
                            const float (&xs)[NODE_IN_S], const float (&xv)[NODE_IN_V][3],
                            float (&row)[ROW]) {
  float s[NS], v[NV][3];
  GvpCache<NS, NV, NV> c;
  LnCache lc;
  gvp_forward<NTN, NODE_IN_S, NODE_IN_V, NS, NV, NV, false>(P + L.node_gvp, type, xs, xv, s, v, c);
  ln_forward<NS, NV>(P + L.node_ln, s, v, lc);
#pragma unroll
  for (int k = 0; k < NS; ++k) row[k] = s[k];
#pragma unroll
  for (int i = 0; i < NV; ++i) { row[NS + 3 * i] = v[i][0]; row[NS + 3 * i + 1] = v[i][1]; row[NS + 3 * i + 2] = v[i][2]; }
}

// One edge of GVPConv (gvp_layers.py:303-308) with the edge embedding
// (protein_gnn.py:331-335, :376) recomputed in registers instead of being
// materialised in HBM: raw edge features -> GVP -> LayerNorm -> concat with the
// source (j) and target (i) node rows -> three message GVPs -> merged message.
template <int NTE>
GVP_HD void conv_message_item(const float* P, const EncLayout& L, int layer, int etype,
                              const float (&es_raw)[EDGE_IN_S], const float (&ev_raw)[EDGE_IN_V][3],
                              const float (&xj)[ROW], const float (&xi)[ROW], float (&msg)[ROW]) {
  float ms[MS], mv[MV][3];
  {
    float es[ES], ev[EV][3];
    GvpCache<ES, EV, EV> c;
    LnCache lc;
    gvp_forward<NTE, EDGE_IN_S, EDGE_IN_V, ES, EV, EV, false>(P + L.edge_gvp, etype, es_raw, ev_raw, es, ev, c);
    ln_forward<ES, EV>(P + L.edge_ln, es, ev, lc);
    // tuple_cat((s_j, v_j), edge_attr, (s_i, v_i))  -- gvp_layers.py:306
#pragma unroll
    for (int k = 0; k < NS; ++k) { ms[k] = xj[k]; ms[NS + ES + k] = xi[k]; }
#pragma unroll
    for (int k = 0; k < ES; ++k) ms[NS + k] = es[k];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int d = 0; d < 3; ++d) { mv[i][d] = xj[NS + 3 * i + d]; mv[NV + EV + i][d] = xi[NS + 3 * i + d]; }
#pragma unroll
    for (int d = 0; d < 3; ++d) mv[NV][d] = ev[0][d];
  }
  const float* C = P + L.conv0 + layer * L.conv_stride;
  float s1[NS], v1[NV][3], s2[NS], v2[NV][3];
  {
    GvpCache<NS, NV, MV> c0;
    gvp_forward<0, MS, MV, NS, NV, MV, true>(C + CONV_M0, 0, ms, mv, s1, v1, c0);
  }
  GvpCache<NS, NV, NV> c1;
  gvp_forward<0, NS, NV, NS, NV, NV, true>(C + conv_m1(), 0, s1, v1, s2, v2, c1);
  gvp_forward<0, NS, NV, NS, NV, NV, false>(C + conv_m2(), 0, s2, v2, s1, v1, c1);
#pragma unroll
  for (int k = 0; k < NS; ++k) msg[k] = s1[k];
#pragma unroll
  for (int i = 0; i < NV; ++i) { msg[NS + 3 * i] = v1[i][0]; msg[NS + 3 * i + 1] = v1[i][1]; msg[NS + 3 * i + 2] = v1[i][2]; }
}

// Node side of GVPConvLayer (gvp_layers.py:407-410, eval mode): x = LN0(x + dh);
// x = LN1(x + FF(x)).  When HEAD, also gvp_norm_before_scalar + gvp_to_scalar
// (protein_gnn.py:385-386) producing the 64-wide residue embedding.
template <bool HEAD>
GVP_HD void node_update_item(const float* P, const EncLayout& L, int layer, const float (&x)[ROW], const float (&dh)[ROW],
                             float (&row)[ROW], float (&out)[OUT]) {
  float s[NS], v[NV][3];
#pragma unroll
  for (int k = 0; k < NS; ++k) s[k] = x[k] + dh[k];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int d = 0; d < 3; ++d) v[i][d] = x[NS + 3 * i + d] + dh[NS + 3 * i + d];
  const float* C = P + L.conv0 + layer * L.conv_stride;
  LnCache lc;
  ln_forward<NS, NV>(C + conv_ln0(), s, v, lc);
  {
    float hs[FS], hv[FV][3], s2[NS], v2[NV][3];
    {
      GvpCache<FS, FV, FV> c;
      gvp_forward<0, NS, NV, FS, FV, FV, true>(C + conv_ff0(), 0, s, v, hs, hv, c);
    }
    GvpCache<NS, NV, FV> c2;
    gvp_forward<0, FS, FV, NS, NV, FV, false>(C + conv_ff1(), 0, hs, hv, s2, v2, c2);
#pragma unroll
    for (int k = 0; k < NS; ++k) s[k] += s2[k];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int d = 0; d < 3; ++d) v[i][d] += v2[i][d];
  }
  ln_forward<NS, NV>(C + conv_ln1(), s, v, lc);
#pragma unroll
  for (int k = 0; k < NS; ++k) row[k] = s[k];
#pragma unroll
  for (int i = 0; i < NV; ++i) { row[NS + 3 * i] = v[i][0]; row[NS + 3 * i + 1] = v[i][1]; row[NS + 3 * i + 2] = v[i][2]; }
  if (HEAD) {
    ln_forward<NS, NV>(P + L.ln_out, s, v, lc);
    float dummy[1][3];
    GvpCache<OUT, 0, NV> c;
    gvp_forward<0, NS, NV, OUT, 0, NV, true>(P + L.head, 0, s, v, out, dummy, c);
  }
}

}  // namespace gvp
