// gine_quad_kernels.hip -- backward of one GINEConv layer (PyG GINEConv + MLP as used by
// HomoMoleculeGNN_GINE, molecule_gnn.py:240-280) on the MFMA quad-layout blocks of gvp_quad.h.
//
// Work unit: a tile of 16 ATOMS per wave, lane = (item i = l & 15, group g = l >> 4), channels in
// pattern P1 (c = 16 t + 4 g + r).  A wave owns its atoms' incoming edges (dst-sorted CSR), which
// it walks in tiles of 16 EDGES with the same lane mapping:
//   A. edges   : e = W_e [onehot(type) | bond features] + b_e (MFMA), m = relu(x_src + e),
//                DPP segmented sum over the sorted targets -> LDS rows agg[atom][c]
//   B. atoms   : h = (1+eps) x + agg;  t = lrelu(W0 h + b0);  y = W1 t + b1 (MFMA chains);
//                dy -> dt = W1^T dy -> dh = W0^T dt (transposed fragments);
//                dW1 = dy (x) t, dW0 = dt (x) h as MFMA outer products over the 16 atoms (operands
//                transposed on the matrix cores), biases by DPP row sums
//   C. edges   : dm = relu'(m) dh[target]  -> float atomics into g_x[source];
//                dW_e = dm (x) [onehot | features], db_e, d eps
// Weight-gradient partials go to a wave-private block in LDS (AccPriv, state_dict order of the
// layer); the workgroup sums its four blocks and writes ONE slab row.  Fragments are built from
// the plain nn.Linear weights while staging them into LDS (the matrices are tiny), so GINE needs
// no fragment image.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "gvp_internal.h"
#include "gvp_quad.h"

using namespace gq;

namespace {

constexpr int WAVE = 64, TILE = 16;
constexpr int NTL = 4;                 // edge tiles per chunk (64 edges)

constexpr int pad4i(int x) { return (x + 3) / 4 * 4; }

__device__ __forceinline__ int opaque_lane(int lane) {     // see gvp_quad_bwd_kernels.hip
  asm volatile("" : "+v"(lane));
  return lane;
}

// P1 slots over W (a multiple of 16) channels of which only the first VALID exist.
// Ordering of a wave's OWN LDS traffic between phases (rows written by some lanes, read by others of the same wave).
// CGVP_GINE_FENCE = 1: release / acquire fences at workgroup scope, which also drain every outstanding GLOBAL access
// (`s_waitcnt vmcnt(0)`: the g_x atomics of the edge phase, loads in flight); 0: wait for the LDS counter only -- a
// wave's LDS instructions execute in issue order, and the "memory" clobber keeps the compiler from moving accesses across.
// A/B in one GPU-box call (tools/ab_libs.sh): drug chain alone 170.4 (fences) vs 171.0 us (LDS counter only), step with
// both encoders 258.3 vs 258.5 us: no difference, the formally ordered form stays.
#ifndef CGVP_GINE_FENCE
#define CGVP_GINE_FENCE 1
#endif
#if CGVP_GINE_FENCE
#define WAVE_LDS_SYNC()                                         \
  do {                                                          \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      \
    __builtin_amdgcn_wave_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      \
  } while (0)
#else
#define WAVE_LDS_SYNC()                                         \
  do {                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
    __builtin_amdgcn_wave_barrier();                            \
  } while (0)
#endif

#ifdef CGVP_STAMPS     // diagnostic build only (tools/stamp_gine_bwd.py): s_memtime at phase boundaries, one row of 16 per wave
__device__ unsigned long long* g_stamp_buf_gine = nullptr;
#define GSTAMP(slot)                                                                         \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (g_stamp_buf_gine && (threadIdx.x & 63) == 0)                                         \
      g_stamp_buf_gine[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = t_;   \
  } while (0)
#else
#define GSTAMP(slot) do {} while (0)
#endif

template <int W, int VALID>
struct SegClip {
  static_assert(W % 16 == 0, "whole 16-channel tiles");
  static constexpr int steps = W / 4;
  static __host__ __device__ int col(int s, int g) {
    const int c = 16 * (s >> 2) + 4 * g + (s & 3);
    return c < VALID ? c : -1;
  }
};

struct GineQArgs {
  const float* x; const int64_t* ntypes; const float* eattr; const int64_t* etypes;
  const int32_t* rowptr; const int32_t* eperm; const int32_t* esrc; const int32_t* edst; int64_t N;
  const float* eps; const float* we; const float* be; const float* w0; const float* b0;
  const float* w1; const float* b1; float slope;
  const float* mask; gvp::RngArgs rng; const float* g_out; float* g_x; float* slab;
  const float* agg_in; const uint16_t* pos_in;       // SAVED instantiation: the forward's aggregates [N][CINP] and ReLU patterns
};

// dropout factors of channels 16 mt + 4 g .. + 3 of atom n (given mask row of width COUT, or regenerated)
template <int COUT>
__device__ __forceinline__ f4 gine_dropout(const float* mask, const gvp::RngArgs& rng, int64_t n, int mt, int g) {
  if (mask) return *reinterpret_cast<const f4*>(mask + n * COUT + 16 * mt + 4 * g);
  if (rng.seed) {
    float f[4];
    gvp::dropout4(rng.seed[0], rng.seed[1], rng.stream, n, 4 * mt + g, rng.p, f);
    return f4{f[0], f[1], f[2], f[3]};
  }
  return f4{1.f, 1.f, 1.f, 1.f};
}

template <int CIN, int CHID, int COUT, int NT, int NET, int ED>
struct GineQ {
  static_assert(CHID % 16 == 0 && COUT % 16 == 0 && COUT <= 64, "hidden / output widths are whole 16-channel tiles (<= 4 output tiles)");
  static constexpr int KE = NET + ED, XW = CIN - NT, CINP = (CIN + 15) / 16 * 16;
  static constexpr int MI = CINP / 16, MH = CHID / 16, MO = COUT / 16;
  // layer gradient block, state_dict order (= GineLay of gvp_kernels.hip)
  static constexpr int L_EPS = 0, L_W0 = 1, L_B0 = L_W0 + CHID * CIN, L_W1 = L_B0 + CHID, L_B1 = L_W1 + COUT * CHID,
                       L_WE = L_B1 + COUT, L_BE = L_WE + CIN * KE, L_SIZE = L_BE + CIN, BLK = pad4i(L_SIZE);
  typedef SegClip<CINP, CIN> KIn;
  typedef Seg<P1, 0, CHID> KHid;
  typedef Seg<P1, 0, COUT> KOut;
  typedef Seg<P2, 0, KE> KFeat;
  static_assert(KFeat::steps <= 4, "bond one-hot + features fit one slot tile");
  typedef Gemm<P1, CIN, KE, KFeat> GE;          // lin      [CIN][KE]
  typedef Gemm<P1, CHID, CIN, KIn> G0;          // lins.0   [CHID][CIN]
  typedef Gemm<P1, COUT, CHID, KHid> G1;        // lins.1   [COUT][CHID]
  typedef GemmT<KHid, KOut, CHID> G1T;
  typedef GemmT<KIn, KHid, CIN> G0T;
  static constexpr int F_E = 0, F_0 = F_E + GE::NFRAG * 64, F_0T = F_0 + G0::NFRAG * 64, F_1 = F_0T + G0T::NFRAG * 64,
                       F_1T = F_1 + G1::NFRAG * 64,
                       // the three bias vectors sit behind the fragments (be zero-padded to CINP): the tile loops read
                       // them from LDS -- a global load per use put a memory round trip in front of every GEMM
                       V_BE = F_1T + G1T::NFRAG * 64, V_B0 = V_BE + CINP, V_B1 = V_B0 + CHID, F_SIZE = V_B1 + COUT;
  static constexpr int ROWS = TILE * CINP;       // per wave: agg rows, then dh rows
  // Waves per workgroup: as many as one CU's LDS holds (the grid is capped at the 16 CUs the protein
  // backward leaves free, so waves per CU is what sets the number of tiles a wave has to walk).
  static constexpr int AVAIL = 160 * 256 - 64 * 8 - F_SIZE;
  static constexpr int wpb(int per_wave) { return AVAIL / per_wave >= 8 ? 8 : AVAIL / per_wave; }
  static constexpr int WPB = wpb(BLK + ROWS);
  static_assert(WPB >= 1, "one wave's blocks must fit");
  // operand transposes through an LDS scratch (gvp_quad.h) when that costs no wave, else on the matrix cores
  static constexpr bool LDS_T = wpb(BLK + ROWS + TSCR_FLOATS) == WPB;
  static constexpr int PER_WAVE = BLK + ROWS + (LDS_T ? TSCR_FLOATS : 0);
  static constexpr int TPB = WAVE * WPB;
  static constexpr int LDS_FLOATS = TPB + F_SIZE + WPB * PER_WAVE;
  static_assert(LDS_FLOATS * 4 <= 160 * 1024, "GINE backward LDS plan exceeds the CU");
};

// Fragment image of one weight matrix, built while staging: eight independent (unconditional) loads in
// flight per thread -- an element-at-a-time loop with a load under a branch costs one memory round
// trip per element (~20 us per launch for the 64x64 layer).
template <class G, int NTHR>
__device__ __forceinline__ void stage_fragments(float* lds, const float* __restrict__ W) {
  constexpr int TOTAL = G::NFRAG * 64, B = 8;
  for (int base = threadIdx.x; base < TOTAL; base += NTHR * B) {
    int off[B];
    float v[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const int idx = base + k * NTHR;
      off[k] = idx < TOTAL ? G::offset(idx) : -1;
      v[k] = W[off[k] >= 0 ? off[k] : 0];
    }
#pragma unroll
    for (int k = 0; k < B; ++k) {
      const int idx = base + k * NTHR;
      if (idx < TOTAL) lds[idx] = off[k] >= 0 ? v[k] : 0.f;
    }
  }
}

#define KEEP_UNCONDITIONAL(V) asm volatile("" : "+v"(V))
// CGVP_GINE_PIN = 1 pins every gather (below) in front of the selects (52-wide backward: 334 -> 17 `vmcnt(0)` waits,
// 538 -> 109 branches).  A/B in one GPU-box call at davis_b64 (tools/build_variant.sh + tools/ab_libs.sh, unprofiled
// replays, custom-op host path): drug chain alone 170.6 (un-pinned) vs 172.4 us (pinned), step with both encoders 260.0
// vs 262.5 us -- the loads hit in L2 and were overlapping already; pinning only adds registers.  Default 0.
#ifndef CGVP_GINE_PIN
#define CGVP_GINE_PIN 0
#endif
constexpr bool G_PIN = CGVP_GINE_PIN != 0;
// BRANCH-FREE GATHERS.  Every global load below is unconditional on a CLAMPED index and the value is selected afterwards:
// a load under a per-lane branch is its own `s_waitcnt vmcnt(0)` round trip (188 of them in the 52-wide backward before
// this), an unconditional one joins the others in flight.  Row 0 of every table exists whenever a tile / chunk does.
//   chunk metadata of sorted position c0 + lane (dst = -1 behind the chunk's last edge)
template <int NET, typename ArgsT>
__device__ __forceinline__ void chunk_meta(const ArgsT& a, int32_t c0, int32_t e1, int lane, int32_t& m_eid, int32_t& m_src,
                                           int32_t& m_dst, int32_t& m_et) {
  const bool in = c0 + lane < e1;
  const int32_t p = in ? c0 + lane : e1 - 1;
  m_eid = a.eperm[p];
  m_src = a.esrc[p];
  const int32_t d = a.edst[p];
  m_et = 0;
  if (NET > 0) {
    const int et = (int)a.etypes[m_eid];
    m_et = et < 0 ? 0 : (et >= NET ? NET - 1 : et);
  }
  m_dst = in ? d : -1;
}
//   k-slots [onehot(bond type) | bond features] of edge `eid` and the (typed) feature row of its source atom
template <class Q, int CIN, int NT, int NET, int ED, bool PIN, typename ArgsT>
__device__ __forceinline__ void edge_inputs(const ArgsT& a, int32_t eid, int32_t et, int32_t src, bool active, int g,
                                            float (&fs)[4], f4 (&xj)[Q::MI]) {
  constexpr int KE = Q::KE, XW = Q::XW;
  const int64_t er = (int64_t)(active ? eid : 0) * ED, xr = (int64_t)(active ? src : 0) * XW;
  int nty = -1;
  if (NT > 0) nty = (int)a.ntypes[active ? src : 0];
  float f[4], xv[Q::MI][4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    int col = 4 * s_ + g - NET;
    col = col < 0 ? 0 : (col >= ED ? ED - 1 : col);
    f[s_] = a.eattr[er + col];
  }
#pragma unroll
  for (int mt = 0; mt < Q::MI; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int col = 16 * mt + 4 * g + r - NT;
      col = col < 0 ? 0 : (col >= XW ? XW - 1 : col);
      xv[mt][r] = a.x[xr + col];
    }
  // every load above is CONSUMED here, outside any branch: without this the compiler sinks each load into the taken
  // side of the select below and the round trips are back
  if (PIN) {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) KEEP_UNCONDITIONAL(f[s_]);
#pragma unroll
    for (int mt = 0; mt < Q::MI; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) KEEP_UNCONDITIONAL(xv[mt][r]);
  }
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    const int idx = 4 * s_ + g;
    const float v = idx < NET ? (et == idx ? 1.f : 0.f) : (idx < KE ? f[s_] : 0.f);
    fs[s_] = active ? v : 0.f;
  }
#pragma unroll
  for (int mt = 0; mt < Q::MI; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * mt + 4 * g + r;
      const float v = c < NT ? (nty == c ? 1.f : 0.f) : (c < CIN ? xv[mt][r] : 0.f);
      xj[mt][r] = active ? v : 0.f;
    }
}
//   k-slots [onehot(bond type) | bond features] of edge `eid` alone (the backward from saved aggregates needs no source row)
template <class Q, int NET, int ED, typename ArgsT>
__device__ __forceinline__ void edge_features(const ArgsT& a, int32_t eid, int32_t et, bool active, int g, float (&fs)[4]) {
  constexpr int KE = Q::KE;
  const int64_t er = (int64_t)(active ? eid : 0) * ED;
  float f[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    int col = 4 * s_ + g - NET;
    col = col < 0 ? 0 : (col >= ED ? ED - 1 : col);
    f[s_] = a.eattr[er + col];
  }
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    const int idx = 4 * s_ + g;
    const float v = idx < NET ? (et == idx ? 1.f : 0.f) : (idx < KE ? f[s_] : 0.f);
    fs[s_] = active ? v : 0.f;
  }
}
//   the (typed) feature row of atom n0 + i of the tile (zeros for the lanes behind the last atom)
template <class Q, int CIN, int NT, bool PIN, typename ArgsT>
__device__ __forceinline__ void node_inputs(const ArgsT& a, int64_t n, bool valid, int64_t n0, int g, f4 (&xi)[Q::MI]) {
  constexpr int XW = Q::XW;
  const int64_t nr = valid ? n : n0;
  int nty = -1;
  if (NT > 0) nty = (int)a.ntypes[nr];
  float xv[Q::MI][4];
#pragma unroll
  for (int mt = 0; mt < Q::MI; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int col = 16 * mt + 4 * g + r - NT;
      col = col < 0 ? 0 : (col >= XW ? XW - 1 : col);
      xv[mt][r] = a.x[nr * XW + col];
    }
  if (PIN) {
#pragma unroll
    for (int mt = 0; mt < Q::MI; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) KEEP_UNCONDITIONAL(xv[mt][r]);
  }
#pragma unroll
  for (int mt = 0; mt < Q::MI; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * mt + 4 * g + r;
      const float v = c < NT ? (nty == c ? 1.f : 0.f) : (c < CIN ? xv[mt][r] : 0.f);
      xi[mt][r] = valid ? v : 0.f;
    }
}

template <int CIN, int CINP, int CHID, int COUT, int NTHR>
__device__ __forceinline__ void stage_biases(float* v_be, float* v_b0, float* v_b1, const float* __restrict__ be,
                                             const float* __restrict__ b0, const float* __restrict__ b1) {
  for (int k = threadIdx.x; k < CINP; k += NTHR) v_be[k] = k < CIN ? be[k] : 0.f;
  for (int k = threadIdx.x; k < CHID; k += NTHR) v_b0[k] = b0[k];
  for (int k = threadIdx.x; k < COUT; k += NTHR) v_b1[k] = b1[k];
}

// rows [16 X, 16 X + 16) of dW1 = dy (x) t
template <class Q, int X>
__device__ __forceinline__ void dw1_band(const f4 (&AT)[Q::MO], const f4 (&BT)[Q::MH], float* blk, int lane) {
  if constexpr (X < Q::MO) {
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const f4 a1[1] = {AT[X]};
    f4 acc[1][Q::MH];
#pragma unroll
    for (int y = 0; y < Q::MH; ++y) acc[0][y] = zero;
    outer_items<1, Q::MH>(a1, BT, acc);
    flush_slots<AccPriv, Seg<P1, 16 * X, 16>, typename Q::KHid, 1, Q::MH>(blk + Q::L_W1, false, Q::KHid::steps * 4, acc, lane);
  }
}

// SAVED: the forward pass of the same step stored the aggregated messages and every message's ReLU pattern
// (cgvp_gine_fwd_ws.agg / .pos): phase A -- the gather of the source rows (CIN floats per edge), W_e e, ReLU and the segmented
// sum, 48 % of a tile's cycles in the round-3 stamps -- is replaced by one row load per atom, and the edge phase reads 8 bytes
// of pattern per edge instead of recomputing the messages.  !SAVED: the fine-grained entry point (no forward workspace).
template <int CIN, int CHID, int COUT, int NT, int NET, int ED, bool SAVED>
__global__ __launch_bounds__((WAVE * GineQ<CIN, CHID, COUT, NT, NET, ED>::WPB)) void gine_quad_bwd_kernel(GineQArgs a) {
  typedef GineQ<CIN, CHID, COUT, NT, NET, ED> Q;
  constexpr int GQ_WPB = Q::WPB, GQ_TPB = Q::TPB;
  constexpr int KE = Q::KE, XW = Q::XW, CINP = Q::CINP, MI = Q::MI, MH = Q::MH, MO = Q::MO;
  extern __shared__ __attribute__((aligned(16))) float lds[];       // lds[0..GQ_TPB): AccPriv::trash()
  float* frag = lds + GQ_TPB;
  const int lane0 = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* blocks = frag + Q::F_SIZE;
  float* blk = blocks + w * Q::BLK;                                  // this wave's private gradient block
  float* rows = blocks + GQ_WPB * Q::BLK + w * Q::ROWS;              // [16][CINP]
  float* const tscr = Q::LDS_T ? blocks + GQ_WPB * (Q::BLK + Q::ROWS) + w * TSCR_FLOATS : nullptr;
  stage_fragments<typename Q::GE, GQ_TPB>(frag + Q::F_E, a.we);
  stage_fragments<typename Q::G0, GQ_TPB>(frag + Q::F_0, a.w0);
  stage_fragments<typename Q::G0T, GQ_TPB>(frag + Q::F_0T, a.w0);
  stage_fragments<typename Q::G1, GQ_TPB>(frag + Q::F_1, a.w1);
  stage_fragments<typename Q::G1T, GQ_TPB>(frag + Q::F_1T, a.w1);
  stage_biases<CIN, CINP, CHID, COUT, GQ_TPB>(frag + Q::V_BE, frag + Q::V_B0, frag + Q::V_B1, a.be, a.b0, a.b1);
  for (int k = lane0; k < Q::BLK / 4; k += WAVE) reinterpret_cast<f4*>(blk)[k] = f4{0.f, 0.f, 0.f, 0.f};
  GSTAMP(0);
  __syncthreads();
  GSTAMP(1);

  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const float eps1 = 1.0f + a.eps[0];
  float acc_eps = 0.f;
  const int64_t ntiles = (a.N + TILE - 1) / TILE;
  for (int64_t tile = (int64_t)blockIdx.x * GQ_WPB + w; tile < ntiles; tile += (int64_t)gridDim.x * GQ_WPB) {
    const int64_t n0 = tile * TILE;
    const int nn = (int)((a.N - n0 < TILE) ? (a.N - n0) : TILE);
    const int32_t e0 = a.rowptr[n0], e1 = a.rowptr[n0 + nn];
    for (int k = lane0; k < Q::ROWS / 4; k += WAVE) reinterpret_cast<f4*>(rows)[k] = zero;
    WAVE_LDS_SYNC();

    // Edges are taken in CHUNKS of up to 64 (4 edge tiles).  A chunk's metadata is fetched
    // lane-parallel (lane l <- edge c0 + l) and handed to the (edge i, group g) lanes of each tile by
    // ds_bpermute; then every load of the chunk (bond features, source rows) is in flight at once:
    // a chunk costs ~4 dependent memory hops instead of 3 per edge tile.  Molecules have <= 64 incoming
    // edges per 16 atoms almost always; then phase C reuses the registers and issues no loads at all.
    int32_t c_src[NTL], c_dst[NTL];
    float c_fs[NTL][1][4];
    unsigned c_pos[NTL];                                   // bit 4 mt + r: message channel passed the ReLU
    const bool single = !SAVED && e1 - e0 <= NTL * TILE;

    auto load_chunk = [&](int32_t c0, int lane, f4 (&xj)[NTL][MI]) {
      const int i = lane & 15, g = lane >> 4;
      int32_t m_eid, m_src, m_dst, m_et;
      chunk_meta<NET>(a, c0, e1, lane, m_eid, m_src, m_dst, m_et);
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int sl = 16 * t + i;
        const int32_t eid = __shfl(m_eid, sl), et = __shfl(m_et, sl);
        c_src[t] = __shfl(m_src, sl);
        c_dst[t] = __shfl(m_dst, sl);                      // -1: no such edge
        edge_inputs<Q, CIN, NT, NET, ED, G_PIN>(a, eid, et, c_src[t], c_dst[t] >= 0, g, c_fs[t][0], xj[t]);
      }
    };
    // message pre-activation of tile t of the loaded chunk -> ReLU pattern (and the messages)
    auto messages = [&](int t, int lane, const f4 (&xj)[NTL][MI], f4 (&m)[MI]) {
      const int g = lane >> 4;
      unsigned pos = 0u;
#pragma unroll
      for (int mt = 0; mt < MI; ++mt) {
        f4 acc[1] = {*reinterpret_cast<const f4*>(frag + Q::V_BE + 16 * mt + 4 * g)};
        apply<typename Q::GE, 1>(frag + Q::F_E, mt, c_fs[t], acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mj = xj[t][mt][r] + acc[0][r];
          const bool on = c_dst[t] >= 0 && mj > 0.f;
          m[mt][r] = on ? mj : 0.f;
          pos |= on ? (1u << (4 * mt + r)) : 0u;
        }
      }
      c_pos[t] = pos;
    };

    // ---- A. aggregate relu(x_src + e) over the incoming edges of the owned atoms (SAVED: read in phase B instead)
    for (int32_t c0 = e0; !SAVED && c0 < e1; c0 += NTL * TILE) {
      const int lane = opaque_lane(lane0), i = lane & 15, g = lane >> 4;
      f4 xj[NTL][MI];
      GSTAMP(8);
      load_chunk(c0, lane, xj);
      GSTAMP(9);
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        if (c0 + t * TILE >= e1) break;
        f4 m[MI];
        messages(t, lane, xj, m);
        float xs[4 * MI];
#pragma unroll
        for (int mt = 0; mt < MI; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) xs[4 * mt + r] = m[mt][r];
        const int32_t dst = c_dst[t];
        seg_scan16<4 * MI>(dst, xs);
        const int nxt = __builtin_amdgcn_update_dpp(-1, dst, 0x100 | 1, 0xf, 0xf, false);
        if (dst >= 0 && (i == TILE - 1 || nxt != dst)) {
          float* row = rows + (dst - (int)n0) * CINP;
#pragma unroll
          for (int mt = 0; mt < MI; ++mt) {
            f4* q = reinterpret_cast<f4*>(row + 16 * mt + 4 * g);
            *q = *q + f4{xs[4 * mt], xs[4 * mt + 1], xs[4 * mt + 2], xs[4 * mt + 3]};
          }
        }
        WAVE_LDS_SYNC();
      }
    }

    GSTAMP(2);
    // ---- B. the MLP of the 16 atoms, forward and backward
    {
      const int lane = opaque_lane(lane0), i = lane & 15, g = lane >> 4;
      const bool valid = i < nn;
      const int64_t n = n0 + i;
      f4 xi[MI], h[MI];
      node_inputs<Q, CIN, NT, G_PIN>(a, n, valid, n0, g, xi);
#pragma unroll
      for (int mt = 0; mt < MI; ++mt) {
        f4 ag;
        if (SAVED) {
          ag = *reinterpret_cast<const f4*>(a.agg_in + (valid ? n : n0) * CINP + 16 * mt + 4 * g);
          if (!valid) ag = zero;
        } else {
          ag = *reinterpret_cast<const f4*>(rows + i * CINP + 16 * mt + 4 * g);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mt][r] = fmaf(eps1, xi[mt][r], ag[r]);
      }
      float bh[1][Q::KIn::steps];
#pragma unroll
      for (int s = 0; s < Q::KIn::steps; ++s) bh[0][s] = h[s >> 2][s & 3];
      f4 tpre[MH], t[MH];
#pragma unroll
      for (int mt = 0; mt < MH; ++mt) {
        f4 acc[1] = {*reinterpret_cast<const f4*>(frag + Q::V_B0 + 16 * mt + 4 * g)};
        apply<typename Q::G0, 1>(frag + Q::F_0, mt, bh, acc, lane);
        tpre[mt] = acc[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[mt][r] = acc[0][r] > 0.f ? acc[0][r] : acc[0][r] * a.slope;
      }
      float bt[1][Q::KHid::steps];
#pragma unroll
      for (int s = 0; s < Q::KHid::steps; ++s) bt[0][s] = t[s >> 2][s & 3];
      f4 dy[MO];
#pragma unroll
      for (int mt = 0; mt < MO; ++mt) {
        f4 acc[1] = {*reinterpret_cast<const f4*>(frag + Q::V_B1 + 16 * mt + 4 * g)};
        apply<typename Q::G1, 1>(frag + Q::F_1, mt, bt, acc, lane);
        f4 gy = *reinterpret_cast<const f4*>(a.g_out + (valid ? n : n0) * COUT + 16 * mt + 4 * g);
        gy = gy * gine_dropout<COUT>(a.mask, a.rng, valid ? n : n0, mt, g);
        if (!valid) gy = zero;
#pragma unroll
        for (int r = 0; r < 4; ++r) dy[mt][r] = gy[r] * (acc[0][r] > 0.f ? 1.f : a.slope);
      }
      float bdy[1][Q::KOut::steps];
#pragma unroll
      for (int s = 0; s < Q::KOut::steps; ++s) bdy[0][s] = dy[s >> 2][s & 3];
      f4 dtp[MH];
#pragma unroll
      for (int mt = 0; mt < MH; ++mt) {
        f4 acc[1] = {zero};
        apply<typename Q::G1T, 1>(frag + Q::F_1T, mt, bdy, acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) dtp[mt][r] = acc[0][r] * (tpre[mt][r] > 0.f ? 1.f : a.slope);
      }
      float bdt[1][Q::KHid::steps];
#pragma unroll
      for (int s = 0; s < Q::KHid::steps; ++s) bdt[0][s] = dtp[s >> 2][s & 3];
      f4 dh[MI];
#pragma unroll
      for (int mt = 0; mt < MI; ++mt) {
        f4 acc[1] = {zero};
        apply<typename Q::G0T, 1>(frag + Q::F_0T, mt, bdt, acc, lane);
        dh[mt] = acc[0];
      }
      GSTAMP(3);
      // ---- weight gradients of the two Linear layers
      {
        f4 AT[MO], BT[MH];
        transpose_slots<Q::KOut::steps>(bdy[0], AT, lane, tscr);
        transpose_slots<Q::KHid::steps>(bt[0], BT, lane, tscr);
        // one 16-row band of dW1 at a time: 4 accumulator tiles live instead of MO x MH
        dw1_band<Q, 0>(AT, BT, blk, lane);
        dw1_band<Q, 1>(AT, BT, blk, lane);
        dw1_band<Q, 2>(AT, BT, blk, lane);
        dw1_band<Q, 3>(AT, BT, blk, lane);
#pragma unroll
        for (int mt = 0; mt < MO; ++mt) {
          float tot[4];
          float* q[4];
          bool on[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            tot[r] = row_total(dy[mt][r]);
            q[r] = blk + Q::L_B1 + 16 * mt + 4 * g + r;
            on[r] = i == 15;
          }
          add_where<AccPriv, 4>(q, on, tot);
        }
      }
      {
        f4 AT[MH], BT[MI], acc[MH][MI];
        transpose_slots<Q::KHid::steps>(bdt[0], AT, lane, tscr);
        transpose_slots<Q::KIn::steps>(bh[0], BT, lane, tscr);
#pragma unroll
        for (int x = 0; x < MH; ++x)
#pragma unroll
          for (int y = 0; y < MI; ++y) acc[x][y] = zero;
        outer_items<MH, MI>(AT, BT, acc);
        flush_slots<AccPriv, typename Q::KHid, typename Q::KIn, MH, MI>(blk + Q::L_W0, false, CIN, acc, lane);
#pragma unroll
        for (int mt = 0; mt < MH; ++mt) {
          float tot[4];
          float* q[4];
          bool on[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            tot[r] = row_total(dtp[mt][r]);
            q[r] = blk + Q::L_B0 + 16 * mt + 4 * g + r;
            on[r] = i == 15;
          }
          add_where<AccPriv, 4>(q, on, tot);
        }
      }
      GSTAMP(4);
      // ---- d eps, d x (self term), and dh rows for the edge phase
#pragma unroll
      for (int mt = 0; mt < MI; ++mt) {
        *reinterpret_cast<f4*>(rows + i * CINP + 16 * mt + 4 * g) = dh[mt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * mt + 4 * g + r;
          acc_eps = fmaf(dh[mt][r], xi[mt][r], acc_eps);
          if (a.g_x && valid && c >= NT && c < CIN) atomicAdd(a.g_x + n * XW + (c - NT), eps1 * dh[mt][r]);
        }
      }
      WAVE_LDS_SYNC();
    }

    GSTAMP(5);
    // ---- C. edges again: d message -> sources, lin weight gradients
    for (int32_t c0 = e0; c0 < e1; c0 += NTL * TILE) {
      const int lane = opaque_lane(lane0), i = lane & 15, g = lane >> 4;
      if (SAVED) {                                 // metadata, bond features and the saved ReLU patterns: no source rows
        int32_t m_eid, m_src, m_dst, m_et;
        chunk_meta<NET>(a, c0, e1, lane, m_eid, m_src, m_dst, m_et);
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          const int sl = 16 * t + i;
          const int32_t eid = __shfl(m_eid, sl), et = __shfl(m_et, sl);
          c_src[t] = __shfl(m_src, sl);
          c_dst[t] = __shfl(m_dst, sl);
          edge_features<Q, NET, ED>(a, eid, et, c_dst[t] >= 0, g, c_fs[t][0]);
          const int32_t p = c0 + sl;
          const unsigned bits = a.pos_in[(int64_t)(p < e1 ? p : e1 - 1) * 4 + g];
          c_pos[t] = c_dst[t] >= 0 ? bits : 0u;
        }
      } else if (!single) {                        // more than one chunk: its registers were overwritten
        f4 xj[NTL][MI];
        load_chunk(c0, lane, xj);
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          if (c0 + t * TILE >= e1) break;
          f4 m[MI];
          messages(t, lane, xj, m);
        }
      }
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        if (c0 + t * TILE >= e1) break;
        const int32_t src = c_src[t], dst = c_dst[t];
        const bool active = dst >= 0;
        f4 dm[MI];
        float dms[Q::KIn::steps];
#pragma unroll
        for (int mt = 0; mt < MI; ++mt) {
          f4 d = zero;
          if (active) d = *reinterpret_cast<const f4*>(rows + (dst - (int)n0) * CINP + 16 * mt + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * mt + 4 * g + r;
            dm[mt][r] = ((c_pos[t] >> (4 * mt + r)) & 1u) ? d[r] : 0.f;
            dms[4 * mt + r] = dm[mt][r];
            if (a.g_x && c >= NT && c < CIN && dm[mt][r] != 0.f)
              atomicAdd(a.g_x + (int64_t)src * XW + (c - NT), dm[mt][r]);
          }
        }
        f4 AT[MI], BT[1], acc[MI][1];
        transpose_slots<Q::KIn::steps>(dms, AT, lane, tscr);
        transpose_slots<4>(c_fs[t][0], BT, lane, tscr);
#pragma unroll
        for (int x = 0; x < MI; ++x) acc[x][0] = zero;
        outer_items<MI, 1>(AT, BT, acc);
        flush_slots<AccPriv, typename Q::KIn, typename Q::KFeat, MI, 1>(blk + Q::L_WE, false, KE, acc, lane);
#pragma unroll
        for (int mt = 0; mt < MI; ++mt) {
          float tot[4];
          float* q[4];
          bool on[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * mt + 4 * g + r;
            tot[r] = row_total(dm[mt][r]);
            q[r] = blk + Q::L_BE + c;
            on[r] = i == 15 && c < CIN;
          }
          add_where<AccPriv, 4>(q, on, tot);
        }
      }
    }
    WAVE_LDS_SYNC();
  }
  GSTAMP(6);
  // d eps: sum over the lanes of the wave, into the private block
  for (int off = 32; off > 0; off >>= 1) acc_eps += __shfl_down(acc_eps, off);
  if (lane0 == 0) blk[Q::L_EPS] += acc_eps;
  // one slab row per workgroup
  __syncthreads();
  float* out = a.slab + (size_t)blockIdx.x * Q::L_SIZE;
  for (int k = threadIdx.x; k < Q::L_SIZE; k += GQ_TPB) {
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < GQ_WPB; ++ww) s += blocks[ww * Q::BLK + k];
    out[k] = s;
  }
  GSTAMP(7);
}

// ---------------------------------------------------------------- forward on the same tiles
// out = act(W1 act(W0 ((1+eps) x + sum_edges relu(x_src + e)) + b0) + b1) [* mask]: phases A and the
// forward half of B of the kernel above (no gradient blocks -> 8 waves per workgroup everywhere).
struct GineFArgs {
  const float* x; const int64_t* ntypes; const float* eattr; const int64_t* etypes;
  const int32_t* rowptr; const int32_t* eperm; const int32_t* esrc; const int32_t* edst; int64_t N;
  const float* eps; const float* we; const float* be; const float* w0; const float* b0;
  const float* w1; const float* b1; float slope; const float* mask; gvp::RngArgs rng; float* out;
  float* agg_out; uint16_t* pos_out;                 // training passes: saved for the backward (may be null)
};
// Waves per workgroup of the forward kernel.  Same A/B: 4 waves (40 workgroups at davis_b64) -> drug chain alone 163 us,
// step with both encoders 266-270 us; 8 waves (20 workgroups) -> 171 us alone, 260 us with both: fewer CUs are taken
// away from the one-workgroup-per-CU protein backward kernels while a forward workgroup lives.
#ifndef CGVP_GINE_FWD_WAVES
#define CGVP_GINE_FWD_WAVES 8
#endif
constexpr int GF_WPB = CGVP_GINE_FWD_WAVES, GF_TPB = WAVE * GF_WPB;

template <int CIN, int CHID, int COUT, int NT, int NET, int ED>
__global__ __launch_bounds__(GF_TPB) void gine_quad_fwd_kernel(GineFArgs a) {
  typedef GineQ<CIN, CHID, COUT, NT, NET, ED> Q;
  constexpr int KE = Q::KE, XW = Q::XW, CINP = Q::CINP, MI = Q::MI, MH = Q::MH, MO = Q::MO;
  constexpr int FE = 0, F0 = FE + Q::GE::NFRAG * 64, F1 = F0 + Q::G0::NFRAG * 64, VBE = F1 + Q::G1::NFRAG * 64,
                VB0 = VBE + CINP, VB1 = VB0 + CHID, FSZ = VB1 + COUT;      // fragments, then the bias vectors (as in the backward)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* frag = lds;
  const int lane0 = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* rows = frag + FSZ + w * Q::ROWS;                            // [16][CINP]
  stage_fragments<typename Q::GE, GF_TPB>(frag + FE, a.we);
  stage_fragments<typename Q::G0, GF_TPB>(frag + F0, a.w0);
  stage_fragments<typename Q::G1, GF_TPB>(frag + F1, a.w1);
  stage_biases<CIN, CINP, CHID, COUT, GF_TPB>(frag + VBE, frag + VB0, frag + VB1, a.be, a.b0, a.b1);
  __syncthreads();

  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const float eps1 = 1.0f + a.eps[0];
  const int64_t ntiles = (a.N + TILE - 1) / TILE;
  for (int64_t tile = (int64_t)blockIdx.x * GF_WPB + w; tile < ntiles; tile += (int64_t)gridDim.x * GF_WPB) {
    const int64_t n0 = tile * TILE;
    const int nn = (int)((a.N - n0 < TILE) ? (a.N - n0) : TILE);
    const int32_t e0 = a.rowptr[n0], e1 = a.rowptr[n0 + nn];
    for (int k = lane0; k < Q::ROWS / 4; k += WAVE) reinterpret_cast<f4*>(rows)[k] = zero;
    WAVE_LDS_SYNC();
    // ---- A. messages of the incoming edges, 64 at a time (see the backward kernel)
    for (int32_t c0 = e0; c0 < e1; c0 += NTL * TILE) {
      const int lane = opaque_lane(lane0), i = lane & 15, g = lane >> 4;
      int32_t m_eid, m_src, m_dst, m_et;
      chunk_meta<NET>(a, c0, e1, lane, m_eid, m_src, m_dst, m_et);
      int32_t c_dst[NTL];
      float c_fs[NTL][1][4];
      f4 xj[NTL][MI];
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int sl = 16 * t + i;
        const int32_t eid = __shfl(m_eid, sl), et = __shfl(m_et, sl), src = __shfl(m_src, sl);
        c_dst[t] = __shfl(m_dst, sl);
        edge_inputs<Q, CIN, NT, NET, ED, G_PIN>(a, eid, et, src, c_dst[t] >= 0, g, c_fs[t][0], xj[t]);
      }
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        if (c0 + t * TILE >= e1) break;
        float xs[4 * MI];
        unsigned pos = 0u;
#pragma unroll
        for (int mt = 0; mt < MI; ++mt) {
          f4 acc[1] = {*reinterpret_cast<const f4*>(frag + VBE + 16 * mt + 4 * g)};
          apply<typename Q::GE, 1>(frag + FE, mt, c_fs[t], acc, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float mj = xj[t][mt][r] + acc[0][r];
            const bool on = c_dst[t] >= 0 && mj > 0.f;
            xs[4 * mt + r] = on ? mj : 0.f;
            pos |= on ? (1u << (4 * mt + r)) : 0u;
          }
        }
        if (a.pos_out && c_dst[t] >= 0)               // the message's ReLU pattern, for the backward of this step
          a.pos_out[(int64_t)(c0 + 16 * t + i) * 4 + g] = (uint16_t)pos;
        const int32_t dst = c_dst[t];
        seg_scan16<4 * MI>(dst, xs);
        const int nxt = __builtin_amdgcn_update_dpp(-1, dst, 0x100 | 1, 0xf, 0xf, false);
        if (dst >= 0 && (i == TILE - 1 || nxt != dst)) {
          float* row = rows + (dst - (int)n0) * CINP;
#pragma unroll
          for (int mt = 0; mt < MI; ++mt) {
            f4* q = reinterpret_cast<f4*>(row + 16 * mt + 4 * g);
            *q = *q + f4{xs[4 * mt], xs[4 * mt + 1], xs[4 * mt + 2], xs[4 * mt + 3]};
          }
        }
        WAVE_LDS_SYNC();
      }
    }
    // ---- B. the MLP of the 16 atoms
    {
      const int lane = opaque_lane(lane0), i = lane & 15, g = lane >> 4;
      const bool valid = i < nn;
      const int64_t n = n0 + i;
      float bh[1][Q::KIn::steps];
      f4 xi[MI];
      node_inputs<Q, CIN, NT, G_PIN>(a, n, valid, n0, g, xi);
#pragma unroll
      for (int mt = 0; mt < MI; ++mt) {
        const f4 ag = *reinterpret_cast<const f4*>(rows + i * CINP + 16 * mt + 4 * g);
        if (a.agg_out && valid) *reinterpret_cast<f4*>(a.agg_out + n * CINP + 16 * mt + 4 * g) = ag;
#pragma unroll
        for (int r = 0; r < 4; ++r) bh[0][4 * mt + r] = fmaf(eps1, xi[mt][r], ag[r]);
      }
      float bt[1][Q::KHid::steps];
#pragma unroll
      for (int mt = 0; mt < MH; ++mt) {
        f4 acc[1] = {*reinterpret_cast<const f4*>(frag + VB0 + 16 * mt + 4 * g)};
        apply<typename Q::G0, 1>(frag + F0, mt, bh, acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) bt[0][4 * mt + r] = acc[0][r] > 0.f ? acc[0][r] : acc[0][r] * a.slope;
      }
#pragma unroll
      for (int mt = 0; mt < MO; ++mt) {
        f4 acc[1] = {*reinterpret_cast<const f4*>(frag + VB1 + 16 * mt + 4 * g)};
        apply<typename Q::G1, 1>(frag + F1, mt, bt, acc, lane);
        f4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = acc[0][r] > 0.f ? acc[0][r] : acc[0][r] * a.slope;
        if (valid) {
          y = y * gine_dropout<COUT>(a.mask, a.rng, n, mt, g);
          *reinterpret_cast<f4*>(a.out + n * COUT + 16 * mt + 4 * g) = y;
        }
      }
    }
    WAVE_LDS_SYNC();
  }
}

template <int CIN, int CHID, int COUT, int NT, int NET, int ED>
int launch_fwd(GineFArgs& a, hipStream_t st) {
  typedef GineQ<CIN, CHID, COUT, NT, NET, ED> Q;
  const int64_t tiles = (a.N + TILE - 1) / TILE;
  int64_t wgs = (tiles + GF_WPB - 1) / GF_WPB;
  static const int cap = [] {                  // CGVP_GINE_FWD_WGS=<n>: workgroup cap of the forward (A/B knob, read once)
    const char* e = getenv("CGVP_GINE_FWD_WGS");
    const int v = e ? atoi(e) : 0;
    return v >= 1 ? v : 1024;
  }();
  const int G = (int)(wgs < 1 ? 1 : (wgs > cap ? cap : wgs));
  const size_t lds = (size_t)((Q::GE::NFRAG + Q::G0::NFRAG + Q::G1::NFRAG) * 64 + Q::CINP + CHID + COUT + GF_WPB * Q::ROWS) * sizeof(float);
  CGVP_SET_DYN_LDS_ONCE((gine_quad_fwd_kernel<CIN, CHID, COUT, NT, NET, ED>), lds);
  hipLaunchKernelGGL((gine_quad_fwd_kernel<CIN, CHID, COUT, NT, NET, ED>), dim3(G), dim3(GF_TPB), lds, st, a);
  return 0;
}

template <int CIN, int CHID, int COUT, int NT, int NET, int ED>
int launch(GineQArgs& a, int cap, int* rows, int* row_len, hipStream_t st) {
  typedef GineQ<CIN, CHID, COUT, NT, NET, ED> Q;
  constexpr int GQ_WPB = Q::WPB, GQ_TPB = Q::TPB;
  const int64_t tiles = (a.N + TILE - 1) / TILE;
  int64_t wgs = (tiles + GQ_WPB - 1) / GQ_WPB;
  const int G = (int)(wgs < 1 ? 1 : (wgs > cap ? cap : wgs));
  const size_t lds = (size_t)Q::LDS_FLOATS * sizeof(float);
  if (a.agg_in && a.pos_in) {
    CGVP_SET_DYN_LDS_ONCE((gine_quad_bwd_kernel<CIN, CHID, COUT, NT, NET, ED, true>), lds);
    hipLaunchKernelGGL((gine_quad_bwd_kernel<CIN, CHID, COUT, NT, NET, ED, true>), dim3(G), dim3(GQ_TPB), lds, st, a);
  } else {
    CGVP_SET_DYN_LDS_ONCE((gine_quad_bwd_kernel<CIN, CHID, COUT, NT, NET, ED, false>), lds);
    hipLaunchKernelGGL((gine_quad_bwd_kernel<CIN, CHID, COUT, NT, NET, ED, false>), dim3(G), dim3(GQ_TPB), lds, st, a);
  }
  *rows = G;
  *row_len = Q::L_SIZE;
  return 0;
}

}  // namespace

namespace quad {

int gine_bwd(int cin, int chid, int cout, int nt, int net, int ed, const float* x, const int64_t* ntypes,
             const float* eattr, const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm,
             const int32_t* esrc, const int32_t* edst, int64_t N, const cgvp_gine_w* w, float slope,
             const float* mask, gvp::RngArgs rng, const float* g_out, float* g_x, float* slab, int max_workgroups,
             int* rows, int* row_len, hipStream_t st, const float* agg_in, const uint16_t* pos_in) {
  const int cap = max_workgroups <= 0 ? kGineBwdDefaultGrid : (max_workgroups > kGineBwdMaxGrid ? kGineBwdMaxGrid : max_workgroups);
  if (((uintptr_t)agg_in & 15) || ((uintptr_t)pos_in & 1)) return CGVP_ERR_BAD_ARG;
  GineQArgs a{x, ntypes, eattr, etypes, rowptr, eperm, esrc, edst, N, w->eps, w->we, w->be, w->w0, w->b0,
              w->w1, w->b1, slope, mask, rng, g_out, g_x, slab, agg_in, pos_in};
  // compiled for the layer shapes of HomoMoleculeGNN_GINE in CASTER-DTA (molecule_gnn.py:240-250)
  if (cin == 52 && chid == 16 && cout == 16 && nt == 11 && net == 5 && ed == 9) return launch<52, 16, 16, 11, 5, 9>(a, cap, rows, row_len, st);
  if (cin == 52 && chid == 16 && cout == 16 && nt == 0 && net == 5 && ed == 9) return launch<52, 16, 16, 0, 5, 9>(a, cap, rows, row_len, st);   // nn.Embedding atom types (11-wide), materialised by the host
  if (cin == 16 && chid == 64 && cout == 64 && nt == 0 && net == 5 && ed == 9) return launch<16, 64, 64, 0, 5, 9>(a, cap, rows, row_len, st);
  if (cin == 16 && chid == 16 && cout == 16 && nt == 0 && net == 5 && ed == 9) return launch<16, 16, 16, 0, 5, 9>(a, cap, rows, row_len, st);   // middle layers of deeper stacks
  return CGVP_ERR_UNSUPPORTED_DIMS;
}

// Forward on tiles for the compiled layer shapes; returns 1 when the shape is not compiled (the
// caller then uses the generic one-wave-per-atom kernel).
int gine_fwd(int cin, int chid, int cout, int nt, int net, int ed, const float* x, const int64_t* ntypes,
             const float* eattr, const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm,
             const int32_t* esrc, const int32_t* edst, int64_t N, const cgvp_gine_w* w, float slope,
             const float* mask, gvp::RngArgs rng, float* out, float* agg_out, uint16_t* pos_out, hipStream_t st) {
  if ((uintptr_t)agg_out & 15) return CGVP_ERR_BAD_ARG;
  GineFArgs a{x, ntypes, eattr, etypes, rowptr, eperm, esrc, edst, N, w->eps, w->we, w->be, w->w0, w->b0,
              w->w1, w->b1, slope, mask, rng, out, agg_out, pos_out};
  if (cin == 52 && chid == 16 && cout == 16 && nt == 11 && net == 5 && ed == 9) return launch_fwd<52, 16, 16, 11, 5, 9>(a, st);
  if (cin == 52 && chid == 16 && cout == 16 && nt == 0 && net == 5 && ed == 9) return launch_fwd<52, 16, 16, 0, 5, 9>(a, st);
  if (cin == 16 && chid == 64 && cout == 64 && nt == 0 && net == 5 && ed == 9) return launch_fwd<16, 64, 64, 0, 5, 9>(a, st);
  if (cin == 16 && chid == 16 && cout == 16 && nt == 0 && net == 5 && ed == 9) return launch_fwd<16, 16, 16, 0, 5, 9>(a, st);
  return 1;
}

}  // namespace quad

#ifdef CGVP_STAMPS
extern "C" int cgvp_debug_set_stamp_buffer_gine(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf_gine), &buf, sizeof(buf));
}
#endif
