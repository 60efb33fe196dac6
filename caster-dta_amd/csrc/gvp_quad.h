// gvp_quad.h -- MFMA ("quad layout") building blocks of the GVP kernels.
//
// Work unit: a TILE of 16 items (edges or residues) per wave.  Lane l of the
// wave is (item i = l & 15, group g = l >> 4): four lanes share one item and
// each keeps a quarter of its channels, so nothing per-item is ever replicated.
//
// Every channel GEMM of a GVP runs on v_mfma_f32_16x16x4_f32 with the items on
// the N (column) side and the nn.Linear weight as the A operand:
//
//     D[m][n=item] += sum_k W[m][k] * X[k][item]
//     A: lane (m=l&15, g) supplies W[m][k-slot g]      (from an LDS fragment image)
//     B: lane (i=l&15, g) supplies X[k-slot g][item i] (a register it already holds)
//     D: lane (i, g) receives rows m = 4g + r, r = 0..3
//
// Because the k index of a GEMM is only summed over, its order is free: the
// weight fragments are PRE-PERMUTED so that whatever distribution of channels
// the lanes already hold is directly the B operand of the next GEMM.  Two
// distributions are used ("patterns"):
//     P1: channel c = 16 t + 4 g + r   (what a float4 load of a row gives; MFMA D rows)
//     P2: channel c = 4 r + g          (compact for few channels: vectors, gates)
// A D tile is P1 by construction; P2 outputs are obtained by permuting the ROWS
// of the A fragments (row m = 4g + r carries channel 4r + g).  Accumulator tiles
// therefore chain GEMM -> elementwise -> GEMM with no LDS round trip and no
// cross-lane traffic; only LayerNorm statistics cross the 4 lanes of an item.
//
// Weight fragments are built once per parameter update by a tiny prep kernel
// (cgvp_lba_prepare) into an "image" in global memory; each workgroup copies its
// slice into LDS with straight float4 loads and then every MFMA's A operand is
// one conflict-free ds_read_b32 (address = fragment base + lane).
#pragma once
#include <hip/hip_runtime.h>

#include "gvp_math.h"

namespace gq {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int P1 = 0, P2 = 1;
constexpr int ceil4(int x) { return (x + 3) / 4; }

// One run of k-slots of a GEMM: which source column of W feeds slot (step s, group g).
template <int KIND, int BASE, int WIDTH>
struct Seg {
  static_assert(KIND == P2 || WIDTH % 16 == 0, "P1 segments are whole 16-channel tiles");
  static constexpr int steps = KIND == P1 ? WIDTH / 4 : ceil4(WIDTH);
  static __host__ __device__ int col(int s, int g) {
    if (KIND == P1) return BASE + 16 * (s >> 2) + 4 * g + (s & 3);
    const int c = 4 * s + g;
    return c < WIDTH ? BASE + c : -1;
  }
};

template <class... S>
struct Segs {
  static constexpr int steps = (S::steps + ... + 0);
  static __host__ __device__ int col(int s, int g) {
    int res = -1, s0 = 0;
    ((res = (s >= s0 && s < s0 + S::steps) ? S::col(s - s0, g) : res, s0 += S::steps), ...);
    return res;
  }
};

// A weight matrix W[O][LD] (row-major nn.Linear) as MFMA A fragments.
// Fragment (mt, step) holds, for lane (m, g): W[row(mt, m)][col(step, g)] or 0.
template <int ROWKIND, int O, int LD, class KSegs>
struct Gemm {
  static_assert(ROWKIND == P1 || O <= 16, "P2 outputs fit one tile");
  static constexpr int MT = (O + 15) / 16;
  static constexpr int NSTEPS = KSegs::steps;
  static constexpr int NFRAG = MT * NSTEPS;
  static __host__ __device__ int row(int mt, int m) {
    if (ROWKIND == P1) { const int r = 16 * mt + m; return r < O ? r : -1; }
    const int c = 4 * (m & 3) + (m >> 2);
    return c < O ? c : -1;
  }
  // element `idx` (= frag * 64 + lane) of this GEMM's fragment image
  static __host__ __device__ float element(const float* W, int idx) {
    const int lane = idx & 63, f = idx >> 6;
    const int mt = f / NSTEPS, s = f - mt * NSTEPS;
    const int r = row(mt, lane & 15), c = KSegs::col(s, lane >> 4);
    return (r >= 0 && c >= 0) ? W[r * LD + c] : 0.f;
  }
};

__device__ __forceinline__ f4 mfma(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[j] += sum over this GEMM's k-steps of A-fragment(mt, s) x b[j][s] for TN
// tiles in lockstep: one LDS fragment read feeds TN independent MFMA chains.
// With a single tile, long chains are split over two accumulators instead (a
// dependent f32 MFMA chain issues every 40 cycles, independent ones every 32).
template <class G, int TN>
__device__ __forceinline__ void apply(const float* frag, int mt, const float (&b)[TN][G::NSTEPS], f4 (&acc)[TN], int lane) {
  const float* f = frag + mt * G::NSTEPS * 64 + lane;
  if (TN == 1 && G::NSTEPS >= 8) {
    f4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s + 1 < G::NSTEPS; s += 2) {
      acc[0] = mfma(f[s * 64], b[0][s], acc[0]);
      acc2 = mfma(f[(s + 1) * 64], b[0][s + 1], acc2);
    }
    if (G::NSTEPS & 1) acc[0] = mfma(f[(G::NSTEPS - 1) * 64], b[0][G::NSTEPS - 1], acc[0]);
    acc[0] += acc2;
    return;
  }
#pragma unroll
  for (int s = 0; s < G::NSTEPS; ++s) {
    const float a = f[s * 64];
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = mfma(a, b[j][s], acc[j]);
  }
}

// Shift within the 16-lane row of a group g (= the 16 items of the tile):
// lane i receives lane i-K's value, lanes i < K receive `fill`.
template <int K>
__device__ __forceinline__ int row_shr_i(int x, int fill) {
  return __builtin_amdgcn_update_dpp(fill, x, 0x110 | K, 0xf, 0xf, false);
}
template <int K>
__device__ __forceinline__ float row_shr_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x110 | K, 0xf, 0xf, false));
}
// One Hillis-Steele step of a segmented inclusive scan over the items of a tile;
// `id` is the (sorted) segment key of each item.
template <int K, int NVAL>
__device__ __forceinline__ void seg_scan_step(int id, float (&x)[NVAL]) {
  const bool same = row_shr_i<K>(id, -2) == id;
#pragma unroll
  for (int k = 0; k < NVAL; ++k) {
    const float y = row_shr_f<K>(x[k]);
    x[k] += same ? y : 0.f;
  }
}
template <int NVAL>
__device__ __forceinline__ void seg_scan16(int id, float (&x)[NVAL]) {
  seg_scan_step<1, NVAL>(id, x);
  seg_scan_step<2, NVAL>(id, x);
  seg_scan_step<4, NVAL>(id, x);
  seg_scan_step<8, NVAL>(id, x);
}

__device__ __forceinline__ float quad_sum(float x) {   // over the 4 lanes of an item
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

// ------------------------------------------------------------------ one GVP
// Image slice of a GVP: [Wh frags | Ws frags | Wv frags | Wsv frags] then a
// vector region [bs (SO) | type table (NT x SO, transposed one-hot columns) |
// bsv (16, P2 order is resolved at read time)].
//   SSegs: k-slots of the scalar inputs, columns relative to the ws row (after
//          the NT type columns are skipped by BASE offsets chosen by the caller)
//   VSegs: k-slots of the vector-channel inputs (columns of wh)
template <int NT, int SI, int VI, int SO, int VO, int H, bool RELU_, class SSegs, class VSegs>
struct GvpQ {
  static_assert(SO % 16 == 0, "scalar outputs are whole tiles");
  static constexpr bool RELU = RELU_;
  static constexpr int K = NT + SI + H;
  static constexpr int HR = ceil4(H), VOR = VO > 0 ? ceil4(VO) : 1, OT = SO / 16;
  static constexpr int SSTEPS = SSegs::steps, VSTEPS = VSegs::steps;
  typedef gvp::GvpLayout<SI, VI, SO, VO, H> A;                       // arena block layout
  typedef Gemm<P2, H, VI, VSegs> GWh;
  typedef Gemm<P1, SO, K, Segs<SSegs, Seg<P2, NT + SI, H>>> GWs;
  typedef Gemm<P2, (VO > 0 ? VO : 1), H, Segs<Seg<P2, 0, H>>> GWv;
  typedef Gemm<P2, (VO > 0 ? VO : 1), SO, Segs<Seg<P1, 0, SO>>> GWsv;
  static constexpr int F_WH = 0;
  static constexpr int F_WS = F_WH + GWh::NFRAG;
  static constexpr int F_WV = F_WS + GWs::NFRAG;
  static constexpr int F_WSV = F_WV + (VO > 0 ? GWv::NFRAG : 0);
  static constexpr int NFRAG = F_WSV + (VO > 0 ? GWsv::NFRAG : 0);
  static constexpr int V_BS = NFRAG * 64;
  static constexpr int V_WT = V_BS + SO;
  static constexpr int V_BSV = V_WT + NT * SO;
  static constexpr int SIZE = V_BSV + 16;        // floats; multiple of 4

  // element `idx` of the image slice, from the GVP's arena block `P`
  static __host__ __device__ float element(const float* P, int idx) {
    if (idx < F_WS * 64) return GWh::element(P, idx);
    if (idx < F_WV * 64) return GWs::element(P + A::ws(NT), idx - F_WS * 64);
    if (VO > 0 && idx < F_WSV * 64) return GWv::element(P + A::wv(NT), idx - F_WV * 64);
    if (VO > 0 && idx < V_BS) return GWsv::element(P + A::wsv(NT), idx - F_WSV * 64);
    if (idx < V_WT) return P[A::bs(NT) + (idx - V_BS)];
    if (idx < V_BSV) { const int j = idx - V_WT; return P[A::ws(NT) + (j % SO) * K + (j / SO)]; }
    const int o = idx - V_BSV;
    return (VO > 0 && o < VO) ? P[A::bsv(NT) + o] : 0.f;
  }

  // Forward for the 16 items of a tile.  `img` = this GVP's slice in LDS.
  //   bs : the lane's scalar k-slot values in SSegs order
  //   bv : per xyz plane, the lane's vector k-slot values in VSegs order
  //   so : OT accumulator tiles (P1: channel 16t + 4g + r)
  //   vo : per plane VOR values (P2: channel 4r + g)
  struct Cache {                 // what the backward pass needs again
    f4 vh[3];                    // wh.V per plane (P2 rows h)
    float vn[HR];
    f4 sp[OT];                   // pre-activation scalars
    f4 vp[3];                    // wv.vh per plane before gating
    f4 sg;                       // sigmoid(gate) (P2 rows o)
  };
  template <int TN>
  static __device__ __forceinline__ void forward(const float* img, int lane, const int (&type)[TN],
                                                 const float (&bs)[TN][SSTEPS], const float (&bv)[TN][3][VSTEPS],
                                                 f4 (&so)[TN][OT], float (&vo)[TN][3][VOR], Cache (&c)[TN]) {
    const int g = lane >> 4;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      float b[TN][VSTEPS];
      f4 acc[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[j] = zero;
#pragma unroll
        for (int s = 0; s < VSTEPS; ++s) b[j][s] = bv[j][p][s];
      }
      apply<GWh, TN>(img + F_WH * 64, 0, b, acc, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) c[j].vh[p] = acc[j];
    }
    float bfull[TN][SSTEPS + HR];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int s = 0; s < SSTEPS; ++s) bfull[j][s] = bs[j][s];
#pragma unroll
      for (int r = 0; r < HR; ++r) {
        const float n2 = c[j].vh[0][r] * c[j].vh[0][r] + c[j].vh[1][r] * c[j].vh[1][r] + c[j].vh[2][r] * c[j].vh[2][r];
        c[j].vn[r] = gvp::f_sqrt(gvp::f_max(n2, gvp::kNormEps));
        bfull[j][SSTEPS + r] = c[j].vn[r];
      }
    }
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      f4 acc[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[j] = *reinterpret_cast<const f4*>(img + V_BS + 16 * t + 4 * g);
        if (NT > 0) acc[j] += *reinterpret_cast<const f4*>(img + V_WT + type[j] * SO + 16 * t + 4 * g);
      }
      apply<GWs, TN>(img + F_WS * 64, t, bfull, acc, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) c[j].sp[t] = acc[j];
    }
    if (VO > 0) {
      float bsp[TN][4 * OT];
      f4 gate[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) bsp[j][4 * t + r] = c[j].sp[t][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) gate[j][r] = img[V_BSV + ((4 * r + g) & 15)];
      }
      apply<GWsv, TN>(img + F_WSV * 64, 0, bsp, gate, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[j].sg[r] = gvp::f_sigmoid(gate[j][r]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float bh[TN][HR];
        f4 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[j] = zero;
#pragma unroll
          for (int r = 0; r < HR; ++r) bh[j][r] = c[j].vh[p][r];
        }
        apply<GWv, TN>(img + F_WV * 64, 0, bh, acc, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          c[j].vp[p] = acc[j];
#pragma unroll
          for (int r = 0; r < VOR; ++r) vo[j][p][r] = acc[j][r] * c[j].sg[r];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) so[j][t][r] = RELU ? gvp::f_max(c[j].sp[t][r], 0.f) : c[j].sp[t][r];
  }
};

// Tuple LayerNorm on a tile (gvp_layers.py:231-242): S scalars in P1 (S/16
// tiles), NV vector channels in P2 (ceil(NV/4) regs per plane).  `ln` points at
// [gamma (S) | beta (S)] in LDS.  In place.
template <int S, int NV>
__device__ __forceinline__ void ln_quad(const float* ln, int lane, f4 (&s)[S / 16], float (&v)[3][ceil4(NV)]) {
  const int g = lane >> 4;
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < S / 16; ++t) sum += s[t][0] + s[t][1] + s[t][2] + s[t][3];
  const float mean = quad_sum(sum) * (1.0f / S);
  float var = 0.f;
#pragma unroll
  for (int t = 0; t < S / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = s[t][r] - mean; var = fmaf(d, d, var); }
  const float rstd = gvp::f_rsqrt(quad_sum(var) * (1.0f / S) + gvp::kLnEps);
#pragma unroll
  for (int t = 0; t < S / 16; ++t) {
    const f4 ga = *reinterpret_cast<const f4*>(ln + 16 * t + 4 * g);
    const f4 be = *reinterpret_cast<const f4*>(ln + S + 16 * t + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[t][r] = fmaf((s[t][r] - mean) * rstd, ga[r], be[r]);
  }
  float n2 = 0.f;
#pragma unroll
  for (int r = 0; r < ceil4(NV); ++r)
    if (4 * r + g < NV)
      n2 += gvp::f_max(v[0][r] * v[0][r] + v[1][r] * v[1][r] + v[2][r] * v[2][r], gvp::kNormEps);
  const float rvn = gvp::f_rsqrt(quad_sum(n2) * (1.0f / NV));
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int r = 0; r < ceil4(NV); ++r) v[p][r] *= rvn;
}

// ------------------------------------------------------------- the LBA encoder
using namespace gvp;

// gvp_node.0: (NT+17, 3) -> (16, 4); x_s columns as k-slot c = 4s + g, x_v channel g.
template <int NT>
using QNode = GvpQ<NT, NODE_IN_S, NODE_IN_V, NS, NV, NV, false, Segs<Seg<P2, NT, NODE_IN_S>>, Segs<Seg<P2, 0, NODE_IN_V>>>;
// gvp_edge.0: (NT+32, 1) -> (32, 1); e_s as two float4 tiles, e_v on group 0.
template <int NT>
using QEdge = GvpQ<NT, EDGE_IN_S, EDGE_IN_V, ES, EV, EV, false, Segs<Seg<P1, NT, EDGE_IN_S>>, Segs<Seg<P2, 0, EDGE_IN_V>>>;
// message_func.0: cat(s_j 16, e_s 32, s_i 16 | v_j 4, e_v 1, v_i 4) -> (16, 4), h = 9.
// Vector k-slots: step 0 = v_j[g] (cols 0..3), step 1 = v_i[g] (cols 5..8), step 2 = e_v on g = 0 (col 4).
using QMsg0 = GvpQ<0, MS, MV, NS, NV, MV, true, Segs<Seg<P1, 0, MS>>,
                   Segs<Seg<P2, 0, NV>, Seg<P2, NV + EV, NV>, Seg<P2, NV, EV>>>;
using QMsg1 = GvpQ<0, NS, NV, NS, NV, NV, true, Segs<Seg<P1, 0, NS>>, Segs<Seg<P2, 0, NV>>>;
using QMsg2 = GvpQ<0, NS, NV, NS, NV, NV, false, Segs<Seg<P1, 0, NS>>, Segs<Seg<P2, 0, NV>>>;
using QFf0 = GvpQ<0, NS, NV, FS, FV, FV, true, Segs<Seg<P1, 0, NS>>, Segs<Seg<P2, 0, NV>>>;
using QFf1 = GvpQ<0, FS, FV, NS, NV, FV, false, Segs<Seg<P1, 0, FS>>, Segs<Seg<P2, 0, FV>>>;
using QHead = GvpQ<0, NS, NV, OUT, 0, NV, true, Segs<Seg<P1, 0, NS>>, Segs<Seg<P2, 0, NV>>>;

// Image = what the kernels copy to LDS, one slice per kernel:
//   embed slice : QNode | ln (2*16)
//   conv slice l: QEdge | edge ln (2*32) | QMsg0 | QMsg1 | QMsg2
//   node slice l: ln0 (32) | QFf0 | QFf1 | ln1 (32)
//   head slice  : ln_out (32) | QHead
template <int NTN, int NTE>
struct Image {
  static constexpr int EMB_GVP = 0, EMB_LN = QNode<NTN>::SIZE, EMB_SIZE = EMB_LN + 2 * NS;
  static constexpr int CV_EDGE = 0, CV_ELN = QEdge<NTE>::SIZE, CV_M0 = CV_ELN + 2 * ES,
                       CV_M1 = CV_M0 + QMsg0::SIZE, CV_M2 = CV_M1 + QMsg1::SIZE, CV_SIZE = CV_M2 + QMsg2::SIZE;
  static constexpr int ND_LN0 = 0, ND_FF0 = 2 * NS, ND_FF1 = ND_FF0 + QFf0::SIZE, ND_LN1 = ND_FF1 + QFf1::SIZE,
                       ND_SIZE = ND_LN1 + 2 * NS;
  static constexpr int HD_LN = 0, HD_GVP = 2 * NS, HD_SIZE = HD_GVP + QHead::SIZE;
  static __host__ __device__ int emb() { return 0; }
  static __host__ __device__ int conv(int l) { return EMB_SIZE + l * (CV_SIZE + ND_SIZE); }
  static __host__ __device__ int node(int l) { return conv(l) + CV_SIZE; }
  static __host__ __device__ int head(int num_convs) { return conv(num_convs); }
  static __host__ __device__ int total(int num_convs) { return head(num_convs) + HD_SIZE; }
};

}  // namespace gq
