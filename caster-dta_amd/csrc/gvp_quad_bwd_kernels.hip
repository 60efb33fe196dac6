// gvp_quad_bwd_kernels.hip -- backward of the LBA protein encoder on the same
// MFMA quad-layout blocks as the forward (gvp_quad.h).
//
// Every kernel recomputes its stage's forward for a tile of 16 items from the
// saved stage inputs (h_l, dh_l, raw features) instead of reading saved
// activations, then back-propagates:
//   * data gradients   : transposed weight fragments, chained in registers like
//                        the forward (tile mt, register r <-> forward k-slot);
//   * weight gradients : dW = sum_items dY (x) X as MFMA outer products whose
//                        operands are transposed through a per-wave LDS scratch;
//                        every wave accumulates an arena-layout block of its own (no
//                        atomics) -- in LDS with a read-add-write per tile (node / head /
//                        embed / edge kernels), in REGISTERS for all of a wave's tiles in
//                        the conv backward (conv_bwd2_kernel, GvpQ::WAcc) -- the workgroup
//                        sums its waves' blocks into one slab row, one reduce launch sums
//                        the rows in a fixed order;
//   * d h[dst]         : in-register segmented scan over the sorted edges (owned
//                        rows, plain stores);  d h[src]: float atomics (the only
//                        ones in the library; sources are unsorted).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "gvp_internal.h"
#include "gvp_quad.h"

using namespace gq;

#ifdef CGVP_STAMPS
__device__ unsigned long long* g_stamp_buf_bwd = nullptr;
extern "C" int cgvp_debug_set_stamp_buffer_bwd(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf_bwd), &buf, sizeof(buf));
}
#endif

namespace {

constexpr int WAVE = 64;
constexpr int TILE = 16;

#ifdef CGVP_STAMPS     // diagnostic build only, see gvp_quad_kernels.hip
#define STAMP(slot)                                                                          \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (g_stamp_buf_bwd && (threadIdx.x & 63) == 0)                                          \
      g_stamp_buf_bwd[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = t_;    \
  } while (0)
// wall-clock start / end of every wave of a kernel (s_memrealtime: 100 MHz, one counter for the whole device), kept per
// kernel kind behind the cycle stamps: buffer[4096 * 16 + (kind * 4096 + wave) * 2 + {0: start, 1: end}]
struct WallStamp {
  int kind;
  __device__ __forceinline__ explicit WallStamp(int k) : kind(k) { put(0); }
  __device__ __forceinline__ ~WallStamp() { put(1); }
  __device__ __forceinline__ void put(int which) const {
    unsigned long long t_;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g_stamp_buf_bwd && (threadIdx.x & 63) == 0 && wave < 4096)
      g_stamp_buf_bwd[4096 * 16 + ((size_t)kind * 4096 + wave) * 2 + which] = t_;
  }
};
#define WALL_STAMP(kind) WallStamp wall_stamp_(kind)
#else
#define STAMP(slot) do {} while (0)
#define WALL_STAMP(kind) do {} while (0)
#endif
// Every backward kernel runs ONE workgroup of 8 waves per CU (2 per SIMD) that owns the CU's
// LDS: [trash word per thread | image slices | one private weight-gradient block per wave (AccPriv)].  Waves take
// tiles round-robin across workgroups, so a small batch still spreads over all CUs; at the
// end the workgroup sums its waves' blocks and writes ONE slab row.
constexpr int BW_WPB = 8, BW_TPB = WAVE * BW_WPB, BW_MAX_GRID = kBwdMaxGrid;
static_assert(BW_MAX_GRID <= 256, "at most one workgroup per CU");

// Lane id the compiler cannot see through.  Taken at the top of every tile: the ~150 lane-dependent
// flush addresses (and friends) of a tile are then recomputed per tile with a few VALU ops instead of
// being hoisted out of the tile loop, spilled to scratch and reloaded one memory round trip at a time.
__device__ __forceinline__ int opaque_lane(int lane) {
  asm volatile("" : "+v"(lane));
  return lane;
}

template <int BLK>
__device__ __forceinline__ void zero_block(float* gblk, int lane) {
  static_assert(BLK % 4 == 0, "blocks are whole float4s");
  for (int k = lane; k < BLK / 4; k += WAVE) reinterpret_cast<f4*>(gblk)[k] = f4{0.f, 0.f, 0.f, 0.f};
}
// slab row of this workgroup = sum of its waves' private blocks (stride PW floats apart)
template <int BLK, int PW, int WPB_ = 8>
__device__ __forceinline__ void write_slab_row(float* slab, const float* blocks, int row = blockIdx.x) {
  __syncthreads();
  float* out = slab + (size_t)row * BLK;
  for (int k = threadIdx.x; k < BLK; k += WAVE * WPB_) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < WPB_; ++ww) t += blocks[ww * PW + k];
    out[k] = t;
  }
}

template <int NFLOATS, int NTHR>
__device__ __forceinline__ void stage_slice(float* lds, const float* __restrict__ src, int tid) {
  static_assert(NFLOATS % 4 == 0, "image slices are whole float4s");
  constexpr int NF4 = NFLOATS / 4, IT = (NF4 + NTHR - 1) / NTHR;
  const f4* s = reinterpret_cast<const f4*>(src);
  f4* d = reinterpret_cast<f4*>(lds);
  f4 v[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) { const int idx = tid + k * NTHR; if (idx < NF4) v[k] = s[idx]; }
#pragma unroll
  for (int k = 0; k < IT; ++k) { const int idx = tid + k * NTHR; if (idx < NF4) d[idx] = v[k]; }
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }
// ONE = every wave of the launch has at most one tile (the host checks tiles <= grid x waves): a wave's private block is
// still all zeros when its only tile adds to it, so the adds are plain stores -- no LDS read in front of each write
// (the node / head / embed stages at 64 x 300 residues are exactly this case)
template <bool ONE> using AccFor = typename std::conditional<ONE, AccStoreOnce, AccPriv>::type;
constexpr int pad4(int x) { return (x + 3) / 4 * 4; }

// LayerNorm parameter gradients of a tile -> [gamma | beta] block in LDS.
template <class Acc, int S>
__device__ __forceinline__ void ln_param_grads(float* blk, bool first, int lane, bool active,
                                               const f4 (&dgamma)[S / 16], const f4 (&dbeta)[S / 16]) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < S / 16; ++t) {
    float tot[8];
    float* q[8];
    bool on[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      tot[r] = row_total(active ? dgamma[t][r] : 0.f);
      tot[4 + r] = row_total(active ? dbeta[t][r] : 0.f);
      q[r] = blk + 16 * t + 4 * g + r;
      q[4 + r] = blk + S + 16 * t + 4 * g + r;
      on[r] = on[4 + r] = i == 15;
    }
    add_where<Acc, 8>(q, on, tot);
  }
}

// ===================================================================== node update
// Arena-layout gradient block of the node side of a conv layer:
//   [norm.0 (32) | norm.1 (32) | ff_func.0 | ff_func.1]   (+ [gvp_norm_before_scalar (32) | gvp_to_scalar])
constexpr int NB_LN0 = 0, NB_LN1 = 2 * NS, NB_FF0 = 4 * NS, NB_FF1 = NB_FF0 + LFf0::size(0),
              NODE_BLK = NB_FF1 + LFf1::size(0);
constexpr int HB_LN = 0, HB_GVP = 2 * NS, HEAD_BLK = HB_GVP + LHead::size(0);
constexpr int NODE_GB = pad4(NODE_BLK), HEAD_GB = pad4(HEAD_BLK);

constexpr int MROW = NS + NV;          // dropout mask row: 16 scalar + 4 vector-channel factors
// factors of node n for lane (i, g): given mask row or regenerated (same function as the forward, gvp_quad_kernels.hip)
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
struct NodeBArgs {
  const float* img_node; const float* imgT_node;
  const float* h; const float* dh; const float* mask0; const float* mask1; gvp::RngArgs rng;
  const float* g_up0; const float* g_up1; const float* g_up2;
  int64_t N; float* g_dh; float* g_h; float* zero_rows; float* slab;
};
constexpr int node_bwd_lds_floats() { return BW_TPB + Image<0, 0>::ND_SIZE + Image<0, 0>::TN_SIZE + BW_WPB * (NODE_GB + TSCR_FLOATS); }
static_assert(node_bwd_lds_floats() * 4 <= 160 * 1024, "node backward LDS plan exceeds the CU");

// g_up0 may alias g_dh (the head backward leaves d h_out there): a tile's rows are read
// by the same lanes that overwrite them at the end of the tile.
template <typename ST, bool ONE = false>
__device__ __forceinline__ void node_bwd_body(const NodeBArgs& a, float* lds) {
  typedef Image<0, 0> IM;
  typedef AccFor<ONE> Acc;
  float* f_node = lds + BW_TPB;                                     // lds[0..BW_TPB): AccPriv::trash()
  float* t_node = f_node + IM::ND_SIZE;
  float* blocks = t_node + IM::TN_SIZE;
  const int lane0 = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* gblk = blocks + w * NODE_GB;                               // this wave's private gradient block
  float* tscr = blocks + BW_WPB * NODE_GB + w * TSCR_FLOATS;        // operand-transpose scratch
  stage_slice<IM::ND_SIZE, BW_TPB>(f_node, a.img_node, threadIdx.x);
  stage_slice<IM::TN_SIZE, BW_TPB>(t_node, a.imgT_node, threadIdx.x);
  zero_block<NODE_GB>(gblk, lane0);
  STAMP(0);
  __syncthreads();
  STAMP(1);

  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const int zt[1] = {0};
  const bool first = false;
  const int64_t ntiles = (a.N + TILE - 1) / TILE;
  for (int64_t tile = (int64_t)w * gridDim.x + blockIdx.x; tile < ntiles; tile += (int64_t)gridDim.x * BW_WPB) {
    const int lane = opaque_lane(lane0), i = lane & 15, g = lane >> 4;
    const int64_t n = tile * TILE + i;
    const bool active = n < a.N;
    // ---- recompute the forward of this tile
    f4 x0[1] = {zero};
    float xv0[3][1] = {{0.f}, {0.f}, {0.f}};
    f4 m0s = {1.f, 1.f, 1.f, 1.f}, m1s = {1.f, 1.f, 1.f, 1.f};
    float m0v = 1.f, m1v = 1.f;
    f4 gs[1] = {zero};
    float gv[3][1] = {{0.f}, {0.f}, {0.f}};
    if (active) {
      const int64_t hr = n * ROW;
      node_dropout(a.mask0, a.rng, 0, n, g, m0s, m0v);      // the factors the forward applied (given, or regenerated)
      node_dropout(a.mask1, a.rng, 1, n, g, m1s, m1v);
      x0[0] = Io<ST>::ld4(a.h, hr + 4 * g) + Io<ST>::ld4(a.dh, hr + 4 * g) * m0s;
#pragma unroll
      for (int p = 0; p < 3; ++p) xv0[p][0] = Io<ST>::ld(a.h, hr + NS + 3 * g + p) + Io<ST>::ld(a.dh, hr + NS + 3 * g + p) * m0v;
      // upstream gradient of this stage's output (sum of up to three buffers)
      const float* ups[3] = {a.g_up0, a.g_up1, a.g_up2};
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (ups[u]) {
          const float* r_ = ups[u] + n * ROW;
          gs[0] += *reinterpret_cast<const f4*>(r_ + 4 * g);
#pragma unroll
          for (int p = 0; p < 3; ++p) gv[p][0] += r_[NS + 3 * g + p];
        }
    }
    f4 y[1] = {x0[0]};
    float yv[3][1] = {{xv0[0][0]}, {xv0[1][0]}, {xv0[2][0]}};
    ln_quad<NS, NV>(f_node + IM::ND_LN0, lane, y, yv);
    f4 hs[1][4], s2[1][1];
    float hv[1][3][2], v2[1][3][1];
    float bs0[1][4], bv0[1][3][1], bs1[1][16], bv1[1][3][2];
    typename Ff0<ST>::Cache c0[1];
    typename Ff1<ST>::Cache c1[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) bs0[0][r] = y[0][r];
#pragma unroll
    for (int p = 0; p < 3; ++p) bv0[0][p][0] = yv[p][0];
    Ff0<ST>::template forward<1, Io<ST>::BF>(f_node + IM::ND_FF0, lane, zt, bs0, bv0, hs, hv, c0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) bs1[0][4 * t + r] = hs[0][t][r];
#pragma unroll
    for (int p = 0; p < 3; ++p) { bv1[0][p][0] = hv[0][p][0]; bv1[0][p][1] = hv[0][p][1]; }
    Ff1<ST>::template forward<1, Io<ST>::BF>(f_node + IM::ND_FF1, lane, zt, bs1, bv1, s2, v2, c1);
    f4 z[1] = {y[0] + s2[0][0] * m1s};
    float zv[3][1];
#pragma unroll
    for (int p = 0; p < 3; ++p) zv[p][0] = yv[p][0] + v2[0][p][0] * m1v;

    STAMP(2);
    STAMP(3);
    // ---- norm.1, feed-forward, residual, norm.0
    {
      f4 dga[1], dbe[1];
      ln_quad_bwd<NS, NV>(f_node + IM::ND_LN1, lane, z, zv, gs, gv, dga, dbe);      // gs/gv := d z
      ln_param_grads<Acc, NS>(gblk + NB_LN1, first, lane, active, dga, dbe);
    }
    float d_hs[16], d_hv[3][2];
    {
      f4 d_so[1] = {gs[0] * m1s};
      float d_vo[3][1] = {{gv[0][0] * m1v}, {gv[1][0] * m1v}, {gv[2][0] * m1v}};
      typename Ff1<ST>::Grads gr1;
      STAMP(4);
      Ff1<ST>::template backward<Io<ST>::BF>(t_node + IM::TN_FF1, lane, c1[0], d_so, d_vo, d_hs, d_hv, gr1);
      STAMP(5);
      Ff1<ST>::template weight_grads<Acc, Io<ST>::BF>(gblk + NB_FF1, first, lane, 0, active, bs1[0], bv1[0], c1[0], gr1, tscr);
    }
    {
      f4 d_so[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) d_so[t] = f4{d_hs[4 * t], d_hs[4 * t + 1], d_hs[4 * t + 2], d_hs[4 * t + 3]};
      float d_ys[4], d_yv[3][1];
      typename Ff0<ST>::Grads gr0;
      STAMP(6);
      Ff0<ST>::template backward<Io<ST>::BF>(t_node + IM::TN_FF0, lane, c0[0], d_so, d_hv, d_ys, d_yv, gr0);
      STAMP(7);
      Ff0<ST>::template weight_grads<Acc, Io<ST>::BF>(gblk + NB_FF0, first, lane, 0, active, bs0[0], bv0[0], c0[0], gr0, tscr);
#pragma unroll
      for (int r = 0; r < 4; ++r) gs[0][r] += d_ys[r];
#pragma unroll
      for (int p = 0; p < 3; ++p) gv[p][0] += d_yv[p][0];
    }
    STAMP(8);
    {
      f4 dga[1], dbe[1];
      ln_quad_bwd<NS, NV>(f_node + IM::ND_LN0, lane, x0, xv0, gs, gv, dga, dbe);    // gs/gv := d (h + dh)
      ln_param_grads<Acc, NS>(gblk + NB_LN0, first, lane, active, dga, dbe);
    }
    if (active) {               // d h (residual path) and d dh = mask0 * d h (equal without dropout)
      if (a.g_h) {
        float* row = a.g_h + n * ROW;
        *reinterpret_cast<f4*>(row + 4 * g) = gs[0];
#pragma unroll
        for (int p = 0; p < 3; ++p) row[NS + 3 * g + p] = gv[p][0];
      }
      float* row = a.g_dh + n * ROW;
      *reinterpret_cast<f4*>(row + 4 * g) = gs[0] * m0s;
#pragma unroll
      for (int p = 0; p < 3; ++p) row[NS + 3 * g + p] = gv[p][0] * m0v;
      if (a.zero_rows) {          // the atomics target of the conv backward that follows: saves its memset launch
        float* z = a.zero_rows + n * ROW;
        *reinterpret_cast<f4*>(z + 4 * g) = zero;
#pragma unroll
        for (int p = 0; p < 3; ++p) z[NS + 3 * g + p] = 0.f;
      }
    }
  }
  STAMP(9);
  write_slab_row<NODE_GB, NODE_GB>(a.slab, blocks);
  STAMP(10);
}
template <typename ST, bool ONE = false>
__global__ __launch_bounds__(BW_TPB, 2) void node_bwd_kernel(NodeBArgs a) {
  WALL_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  node_bwd_body<ST, ONE>(a, lds);
}

// ===================================================================== output head
// gvp_norm_before_scalar + gvp_to_scalar (protein_gnn.py:385-386) on h_out, the saved output of
// the last conv layer: d h_out and the head's weight gradients from g_out [N][64].
struct HeadBArgs {
  const float* img_head; const float* imgT_head; const float* h_out; const float* g_out;
  int64_t N; float* g_h_out; float* slab;
};
constexpr int head_bwd_lds_floats() { return BW_TPB + Image<0, 0>::HD_SIZE + Image<0, 0>::TH_SIZE + BW_WPB * (HEAD_GB + TSCR_FLOATS); }

template <typename ST, bool ONE = false>
__device__ __forceinline__ void head_bwd_body(const HeadBArgs& a, float* lds) {
  typedef Image<0, 0> IM;
  typedef AccFor<ONE> Acc;
  float* f_head = lds + BW_TPB;
  float* t_head = f_head + IM::HD_SIZE;
  float* blocks = t_head + IM::TH_SIZE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* gblk = blocks + w * HEAD_GB;
  float* tscr = blocks + BW_WPB * HEAD_GB + w * TSCR_FLOATS;
  stage_slice<IM::HD_SIZE, BW_TPB>(f_head, a.img_head, threadIdx.x);
  stage_slice<IM::TH_SIZE, BW_TPB>(t_head, a.imgT_head, threadIdx.x);
  zero_block<HEAD_GB>(gblk, lane);
  __syncthreads();

  const int i = lane & 15, g = lane >> 4;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const int zt[1] = {0};
  const bool first = false;
  const int64_t ntiles = (a.N + TILE - 1) / TILE;
  for (int64_t tile = (int64_t)w * gridDim.x + blockIdx.x; tile < ntiles; tile += (int64_t)gridDim.x * BW_WPB) {
    const int64_t n = tile * TILE + i;
    const bool active = n < a.N;
    f4 o1[1] = {zero};
    float ov1[3][1] = {{0.f}, {0.f}, {0.f}};
    f4 d_o[4] = {zero, zero, zero, zero};
    if (active) {
      const int64_t hr = n * ROW;
      o1[0] = Io<ST>::ld4(a.h_out, hr + 4 * g);
#pragma unroll
      for (int p = 0; p < 3; ++p) ov1[p][0] = Io<ST>::ld(a.h_out, hr + NS + 3 * g + p);
      const float* gr_ = a.g_out + n * OUT;
#pragma unroll
      for (int t = 0; t < 4; ++t) d_o[t] = *reinterpret_cast<const f4*>(gr_ + 16 * t + 4 * g);
    }
    f4 wsn[1] = {o1[0]};
    float wvn[3][1] = {{ov1[0][0]}, {ov1[1][0]}, {ov1[2][0]}};
    ln_quad<NS, NV>(f_head + IM::HD_LN, lane, wsn, wvn);
    float bsh[1][4], bvh[1][3][1], dummy[1][3][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) bsh[0][r] = wsn[0][r];
#pragma unroll
    for (int p = 0; p < 3; ++p) bvh[0][p][0] = wvn[p][0];
    f4 o[1][4];
    QHead::Cache ch[1];
    QHead::template forward<1, Io<ST>::BF>(f_head + IM::HD_GVP, lane, zt, bsh, bvh, o, dummy, ch);
    float d_vo[3][1] = {{0.f}, {0.f}, {0.f}}, d_bs[4], d_bv[3][1];
    QHead::Grads grh;
    QHead::template backward<Io<ST>::BF>(t_head, lane, ch[0], d_o, d_vo, d_bs, d_bv, grh);
    QHead::template weight_grads<Acc, Io<ST>::BF>(gblk + HB_GVP, first, lane, 0, active, bsh[0], bvh[0], ch[0], grh, tscr);
    f4 dws[1] = {f4{d_bs[0], d_bs[1], d_bs[2], d_bs[3]}};
    float dwv[3][1] = {{d_bv[0][0]}, {d_bv[1][0]}, {d_bv[2][0]}};
    f4 dga[1], dbe[1];
    ln_quad_bwd<NS, NV>(f_head + IM::HD_LN, lane, o1, ov1, dws, dwv, dga, dbe);
    ln_param_grads<Acc, NS>(gblk + HB_LN, first, lane, active, dga, dbe);
    if (active) {
      float* row = a.g_h_out + n * ROW;
      *reinterpret_cast<f4*>(row + 4 * g) = dws[0];
#pragma unroll
      for (int p = 0; p < 3; ++p) row[NS + 3 * g + p] = dwv[p][0];
    }
  }
  write_slab_row<HEAD_GB, HEAD_GB>(a.slab, blocks);
}
template <typename ST>
__global__ __launch_bounds__(BW_TPB, 2) void head_bwd_kernel(HeadBArgs a) {
  WALL_STAMP(1);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  head_bwd_body<ST>(a, lds);
}

// The last layer's node stage with the output head in front of it, ONE launch: both stages give tile t to the same
// wave (w * grid + block), and lane (i, g) of the node stage reads exactly the words of d h_out that the same lane
// of the head stage wrote (g_h_out = the node stage's g_up0), so the hand-over needs no grid-wide ordering -- only a
// workgroup barrier before the node stage reuses the head stage's LDS.  Saves a launch, its ramp and its drain on
// the backward's critical path (round 4: head_bwd alone was 10.4 us at davis_b64 for ~3 us of tile work).
template <typename ST, bool ONE = false>
__global__ __launch_bounds__(BW_TPB, 2) void node_head_bwd_kernel(HeadBArgs hd, NodeBArgs a) {
  WALL_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  head_bwd_body<ST, ONE>(hd, lds);
  __syncthreads();
  node_bwd_body<ST, ONE>(a, lds);
}

// ===================================================================== conv
// The edge embedding (gvp_edge + LayerNorm, protein_gnn.py:376) is NOT part of this kernel: the first conv layer's
// forward stored it in sorted-edge order (EROW floats per edge), this kernel reads it as an input of message_func.0
// and writes its gradient d e to `g_e` (same layout, same storage type); edge_bwd_kernel below back-propagates the SUM of the layers'
// d e through LayerNorm and gvp_edge once per step.  (Round 1 recomputed the embedding and ran its backward and
// weight gradients inside every conv layer's backward: 32 + ~70 of that kernel's 319 MFMAs per 16 edges, a
// 4.8 KB slice of every wave's private gradient block and 14.6 KB of LDS images.)
// Gradient block: [message_func.0 | .1 | .2].
constexpr int EROW = quad::kEdgeRow;
template <int NTE>
struct ConvBlk {
  static constexpr int M0 = 0, M1 = M0 + LMsg0::size(0), M2 = M1 + LMsg::size(0), SIZE = pad4(M2 + LMsg::size(0));
};
// One workgroup of 8 waves per CU (2 per SIMD):
//   [message slices of the forward image | of the transposed image | per wave: private gradient block, scratch]
#ifndef CGVP_CONV_BWD_WAVES
#define CGVP_CONV_BWD_WAVES 8       // 12 (3 waves per SIMD, 168 VGPRs with 16 spilled) measured 56 us vs 43 us per launch at davis_b64: not worth it
#endif
constexpr int CB_WPB = CGVP_CONV_BWD_WAVES, CB_TPB = WAVE * CB_WPB, CB_MAX_GRID = BW_MAX_GRID;
constexpr int CB_SCR = TSCR_FLOATS;                  // per wave: operand-transpose scratch; also half a tile of [28]-rows + 16 ids (g_src)
static_assert(CB_SCR >= (TILE / 2) * ROW + TILE, "g_src transpose fits the scratch");
template <int NTE>
struct ConvBImg {
  typedef Image<0, NTE> IM;
  static constexpr int F0 = IM::CV_M0, FSIZE = IM::CV_SIZE - IM::CV_M0;       // forward message slices
  static constexpr int T0 = IM::TC_M0, TSIZE = IM::TC_SIZE - IM::TC_M0;       // transposed message slices
};
template <int NTE>
constexpr int conv_bwd_lds_floats() { return CB_TPB + ConvBImg<NTE>::FSIZE + ConvBImg<NTE>::TSIZE + CB_WPB * (ConvBlk<NTE>::SIZE + CB_SCR); }
static_assert(conv_bwd_lds_floats<1>() * 4 <= 160 * 1024 && conv_bwd_lds_floats<0>() * 4 <= 160 * 1024, "conv backward LDS plan exceeds the CU");

struct ConvBArgs {
  const float* img; const float* imgT;
  const float* h; const float* e_emb;
  const int32_t* rowptr; const int32_t* esrc; const int32_t* edst;
  int64_t N; int npw; int mean; const float* g_dh; float* g_src; float* g_dst; float* g_e; float* slab;
};

template <int NTE, typename ST>
__global__ __launch_bounds__(CB_TPB, CB_WPB / 4) void conv_bwd_kernel(ConvBArgs a) {
  WALL_STAMP(2);
  typedef Image<0, NTE> IM;
  typedef ConvBImg<NTE> BI;
  typedef ConvBlk<NTE> B;
  constexpr int PW = B::SIZE + CB_SCR;                                // per-wave LDS floats
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds + CB_TPB - BI::F0;                                 // indexed with the IM::CV_* offsets of the message slices
  float* imgT = lds + CB_TPB + BI::FSIZE - BI::T0;                    // ... and IM::TC_*
  const int lane0 = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* blocks = lds + CB_TPB + BI::FSIZE + BI::TSIZE;
  float* gblk = blocks + w * PW;                                      // this wave's private gradient block (AccPriv)
  float* scr = gblk + B::SIZE;
#ifdef CGVP_MFMA_TRANSPOSE
  float* const CB_TSCR = nullptr;
#else
  float* const CB_TSCR = scr;
#endif
  stage_slice<BI::FSIZE, CB_TPB>(lds + CB_TPB, a.img + BI::F0, threadIdx.x);
  stage_slice<BI::TSIZE, CB_TPB>(lds + CB_TPB + BI::FSIZE, a.imgT + BI::T0, threadIdx.x);
  for (int k = lane0; k < B::SIZE / 4; k += WAVE) reinterpret_cast<f4*>(gblk)[k] = f4{0.f, 0.f, 0.f, 0.f};
  STAMP(0);
  __syncthreads();
  STAMP(1);

  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const int zt[1] = {0};
  const int64_t ngroups = (a.N + a.npw - 1) / a.npw;
  const int64_t grp0 = (int64_t)blockIdx.x * CB_WPB + w;
  const bool first = false;                            // AccPriv: the block starts zeroed
  float carry[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // lane i == 0: running sum of the target that straddles into this tile
  int carry_dst = -1;
  for (int64_t grp = grp0; grp < ngroups; grp += (int64_t)gridDim.x * CB_WPB) {
    const int64_t n0 = grp * a.npw;
    const int nn = (int)((a.N - n0 < a.npw) ? (a.N - n0) : a.npw);
    const int32_t e0 = a.rowptr[n0], e1 = a.rowptr[n0 + nn];
    // owned targets without incoming edges get a zero row (every other owned row is stored below)
    for (int k = lane0; k < nn * ROW; k += WAVE) {
      const int64_t nd = n0 + k / ROW;
      if (a.rowptr[nd + 1] == a.rowptr[nd]) a.g_dst[n0 * ROW + k] = 0.f;
    }
    for (int32_t base = e0; base < e1; base += TILE) {
      const int lane = opaque_lane(lane0), i = lane & 15, g = lane >> 4;
      const int32_t p = base + i;
      const bool active = p < e1;
      // ---- gather: stored edge embedding (sequential) + source / target rows + d(aggregated message)
      f4 es0 = zero, es1 = zero, sj = zero, si = zero, d_ms = zero;
      float ev[3] = {0.f, 0.f, 0.f}, vj[3] = {0.f, 0.f, 0.f}, vi[3] = {0.f, 0.f, 0.f}, d_mv[3] = {0.f, 0.f, 0.f};
      int32_t src = 0, dst = -1;
      if (active) {
        src = a.esrc[p];
        dst = a.edst[p];
        const int64_t er = (int64_t)p * EROW;
        es0 = Io<ST>::ld4(a.e_emb, er + 4 * g);
        es1 = Io<ST>::ld4(a.e_emb, er + 16 + 4 * g);
        if (g == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) ev[d] = Io<ST>::ld(a.e_emb, er + ES + d);
        }
        const int64_t hj = (int64_t)src * ROW, hi = (int64_t)dst * ROW;
        const float* gd = a.g_dh + (int64_t)dst * ROW;       // d(aggregated message) of the target
        sj = Io<ST>::ld4(a.h, hj + 4 * g);
        si = Io<ST>::ld4(a.h, hi + 4 * g);
        d_ms = *reinterpret_cast<const f4*>(gd + 4 * g);
#pragma unroll
        for (int d = 0; d < 3; ++d) { vj[d] = Io<ST>::ld(a.h, hj + NS + 3 * g + d); vi[d] = Io<ST>::ld(a.h, hi + NS + 3 * g + d); d_mv[d] = gd[NS + 3 * g + d]; }
        if (a.mean) {
          const int deg = a.rowptr[dst + 1] - a.rowptr[dst];
          const float sc = 1.0f / (float)(deg > 1 ? deg : 1);
          d_ms *= sc;
#pragma unroll
          for (int d = 0; d < 3; ++d) d_mv[d] *= sc;
        }
      }
      STAMP(2);
      // ---- recompute the three message GVPs of the tile, keeping every GVP's cache
      float b0[1][16], bv0[1][3][3], b1[1][4], bv1[1][3][1], b2[1][4], bv2[1][3][1];
#pragma unroll
      for (int r = 0; r < 4; ++r) { b0[0][r] = sj[r]; b0[0][4 + r] = es0[r]; b0[0][8 + r] = es1[r]; b0[0][12 + r] = si[r]; }
#pragma unroll
      for (int d = 0; d < 3; ++d) { bv0[0][d][0] = vj[d]; bv0[0][d][1] = vi[d]; bv0[0][d][2] = ev[d]; }
      f4 s1[1][1], s2[1][1], s3[1][1];
      float v1[1][3][1], v2[1][3][1], v3[1][3][1];
      typename Msg0<ST>::Cache c0[1];
      typename Msg1<ST>::Cache c1[1];
      typename Msg2<ST>::Cache c2[1];
      Msg0<ST>::template forward<1, Io<ST>::BF>(img + IM::CV_M0, lane, zt, b0, bv0, s1, v1, c0);
#pragma unroll
      for (int r = 0; r < 4; ++r) b1[0][r] = s1[0][0][r];
#pragma unroll
      for (int d = 0; d < 3; ++d) bv1[0][d][0] = v1[0][d][0];
      Msg1<ST>::template forward<1, Io<ST>::BF>(img + IM::CV_M1, lane, zt, b1, bv1, s2, v2, c1);
#pragma unroll
      for (int r = 0; r < 4; ++r) b2[0][r] = s2[0][0][r];
#pragma unroll
      for (int d = 0; d < 3; ++d) bv2[0][d][0] = v2[0][d][0];
      Msg2<ST>::template forward<1, Io<ST>::BF>(img + IM::CV_M2, lane, zt, b2, bv2, s3, v3, c2);

      STAMP(3);
      // ---- backward through the three message GVPs
      // Lanes past the wave's last edge carry d_ms = d_mv = 0 (initialised above, loaded only `if (active)`), and every
      // gradient below is linear in them: their dY operands of the weight-gradient outer products are exactly zero, so
      // the per-operand `active ? x : 0` selects of weight_grads are compiled out here (~60 v_cndmask per tile).
      constexpr bool kAllActive = true;
      float d_b[4], d_bv[3][1];
      {
        f4 d_so[1] = {d_ms};
        float d_vo[3][1] = {{d_mv[0]}, {d_mv[1]}, {d_mv[2]}};
        typename Msg2<ST>::Grads gr;
        Msg2<ST>::template backward<Io<ST>::BF>(imgT + IM::TC_M2, lane, c2[0], d_so, d_vo, d_b, d_bv, gr);
        STAMP(10);
        Msg2<ST>::template weight_grads<AccPriv, Io<ST>::BF>(gblk + B::M2, first, lane, 0, kAllActive, b2[0], bv2[0], c2[0], gr, CB_TSCR);
      }
      {
        f4 d_so[1] = {f4{d_b[0], d_b[1], d_b[2], d_b[3]}};
        float d_vo[3][1] = {{d_bv[0][0]}, {d_bv[1][0]}, {d_bv[2][0]}};
        typename Msg1<ST>::Grads gr;
        STAMP(11);
        Msg1<ST>::template backward<Io<ST>::BF>(imgT + IM::TC_M1, lane, c1[0], d_so, d_vo, d_b, d_bv, gr);
        STAMP(12);
        Msg1<ST>::template weight_grads<AccPriv, Io<ST>::BF>(gblk + B::M1, first, lane, 0, kAllActive, b1[0], bv1[0], c1[0], gr, CB_TSCR);
      }
      float d_b0[16], d_bv0[3][3];
      {
        f4 d_so[1] = {f4{d_b[0], d_b[1], d_b[2], d_b[3]}};
        float d_vo[3][1] = {{d_bv[0][0]}, {d_bv[1][0]}, {d_bv[2][0]}};
        typename Msg0<ST>::Grads gr;
        STAMP(13);
        Msg0<ST>::template backward<Io<ST>::BF>(imgT + IM::TC_M0, lane, c0[0], d_so, d_vo, d_b0, d_bv0, gr);
        STAMP(14);
        Msg0<ST>::template weight_grads<AccPriv, Io<ST>::BF>(gblk + B::M0, first, lane, 0, kAllActive, b0[0], bv0[0], c0[0], gr, CB_TSCR);
      }
      STAMP(4);
      // ---- d(edge embedding) of this layer -> g_e (plain stores, sorted-edge order)
      if (active) {
        const int64_t gr_ = (int64_t)p * EROW;
        Io<ST>::st4(a.g_e, gr_ + 4 * g, f4{d_b0[4], d_b0[5], d_b0[6], d_b0[7]});
        Io<ST>::st4(a.g_e, gr_ + 16 + 4 * g, f4{d_b0[8], d_b0[9], d_b0[10], d_b0[11]});
        if (g == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) Io<ST>::st(a.g_e, gr_ + ES + d, d_bv0[d][2]);
        }
      }
      STAMP(5);
      // ---- d h[src]: unsorted sources -> float atomics on the zero-initialised g_src.  The rows
      // are transposed through LDS (half a tile at a time) so that one wave-instruction adds
      // two whole 112-B rows (lane-per-row atomics run an order of magnitude slower).
      {
        float* trow = scr;                          // [8][28]
        int* tsrc = reinterpret_cast<int*>(scr + (TILE / 2) * ROW);
        if (g == 0) tsrc[i] = active ? src : -1;
        const int half = lane / ROW, col = lane - half * ROW;      // lanes 0..27 -> row 2k, 28..55 -> row 2k+1
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          if ((i >> 3) == hb) {
            float* tr = trow + (i & 7) * ROW;
            *reinterpret_cast<f4*>(tr + 4 * g) = f4{d_b0[0], d_b0[1], d_b0[2], d_b0[3]};
#pragma unroll
            for (int d = 0; d < 3; ++d) tr[NS + 3 * g + d] = d_bv0[d][0];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          if (half < 2) {
#pragma unroll
            for (int k = 0; k < TILE / 4; ++k) {
              const int r = 2 * k + half;
              const int sr = tsrc[8 * hb + r];
              if (sr >= 0) atomicAdd(a.g_src + (int64_t)sr * ROW + col, trow[r * ROW + col]);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
      }
      STAMP(6);
      // ---- d h[dst]: the wave owns its targets and their edges are consecutive, so a row is
      // the segment sum of a DPP scan plus (for the one target that straddles tiles) a carry
      // kept in registers; plain stores, the later store of a straddling row overwrites the
      // partial one stored at the end of the previous tile.
      {
        float x[7] = {d_b0[12], d_b0[13], d_b0[14], d_b0[15], d_bv0[0][1], d_bv0[1][1], d_bv0[2][1]};
        if (i == 0 && active && dst == carry_dst) {
#pragma unroll
          for (int k = 0; k < 7; ++k) x[k] += carry[k];
        }
        seg_scan16<7>(dst, x);
        const int nxt = __builtin_amdgcn_update_dpp(-1, dst, 0x100 | 1, 0xf, 0xf, false);
        if (active && (i == TILE - 1 || nxt != dst)) {
          float* row = a.g_dst + (int64_t)dst * ROW;
          *reinterpret_cast<f4*>(row + 4 * g) = f4{x[0], x[1], x[2], x[3]};
#pragma unroll
          for (int d = 0; d < 3; ++d) row[NS + 3 * g + d] = x[4 + d];
        }
        carry_dst = __builtin_amdgcn_update_dpp(-1, dst, 0x100 | 15, 0xf, 0xf, false);       // lane 0 <- lane 15
#pragma unroll
        for (int k = 0; k < 7; ++k)
          carry[k] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[k]), 0x100 | 15, 0xf, 0xf, false));
      }
    }
    STAMP(7);
  }
  STAMP(8);
  write_slab_row<B::SIZE, PW, CB_WPB>(a.slab, blocks);
  STAMP(9);
}

// ===================================================================== conv, version 2 (round 4)
// Same inputs, outputs and gradient block as conv_bwd_kernel; what changed is how a wave works:
//   * ONE wave per SIMD (4-wave workgroups, one per CU, 512 registers per lane) running TN = 2 tiles of 16 sorted edges IN
//     LOCKSTEP through recompute / data gradients / weight gradients: every LDS weight fragment is read once for both tiles
//     and feeds two independent MFMA chains, the tiles' operand transposes overlap;
//   * the weight-gradient blocks stay in REGISTERS for all the tiles a wave processes (GvpQ::WAcc) and are mapped into the
//     wave's LDS block once, at the end -- no per-tile LDS read-add-write, no per-tile bias row reductions;
//   * waves own contiguous EDGE ranges (whole 32-edge iterations), not target ranges: no half-empty tiles on high-degree
//     graphs (kNN-20: 2 tiles per ~20-edge target before).  A target whose edges start in an earlier wave's range gets
//     this wave's partial sum by float atomics into g_src (which the consumers add to g_dst anyway and which is zero on
//     entry); the wave that holds a target's FIRST edge stores its partial row to g_dst with plain stores; every row of
//     g_dst is written exactly once (targets without edges: zero rows, by node range);
//   * a self loop's d h[src] goes into its target's row sum instead of an atomic (a third of the edges of a 4 A graph).
// Shape of a workgroup: (TN tiles in lockstep per wave, WPB waves): (2, 4) = one wave per SIMD with 512 registers (default);
// (1, 8) = two waves per SIMD with 256 registers each, one tile per wave (the hardware interleaves the two waves' phases).
template <int TN, int WPB>
struct C2Shape {
  static constexpr int TPB = WAVE * WPB, GS_ROWS = TN * TILE;               // d h[src] rows of one iteration
  static constexpr int SCR = TN * TSCR_FLOATS + GS_ROWS * ROW + GS_ROWS;    // per wave: transposes | g_src rows | their ids
  static_assert((TN * TSCR_FLOATS) % 4 == 0 && (GS_ROWS * ROW) % 4 == 0, "scratch regions stay 16-B aligned");
  template <int NTE>
  static constexpr int lds_floats() { return TPB + ConvBImg<NTE>::FSIZE + ConvBImg<NTE>::TSIZE + WPB * (ConvBlk<NTE>::SIZE + SCR); }
};
static_assert(C2Shape<2, 4>::lds_floats<1>() * 4 <= 160 * 1024 && C2Shape<1, 8>::lds_floats<1>() * 4 <= 160 * 1024, "conv backward v2 LDS plan exceeds the CU");

// What one lane gathers for one sorted edge.  BRANCH-FREE: every load is unconditional on a CLAMPED position (a load under a
// per-lane branch is its own wait; unconditional ones join the others in flight and can be issued an iteration ahead);
// lanes behind the wave's last edge read the last edge's (finite) data and get zero upstream gradients.
struct C2Rows {
  f4 es0, es1, sj, si, d_ms;
  float ev[3], vj[3], vi[3], d_mv[3];
};
template <typename ST>
__device__ __forceinline__ void c2_load_rows(const ConvBArgs& a, int32_t pc, int32_t src, int32_t dst, int g, C2Rows& r) {
  const int64_t er = (int64_t)pc * EROW;
  r.es0 = Io<ST>::ld4(a.e_emb, er + 4 * g);
  r.es1 = Io<ST>::ld4(a.e_emb, er + 16 + 4 * g);
#pragma unroll
  for (int d = 0; d < 3; ++d) r.ev[d] = Io<ST>::ld(a.e_emb, er + ES + d);
  const int64_t hj = (int64_t)src * ROW, hi = (int64_t)dst * ROW;
  const float* gd = a.g_dh + (int64_t)dst * ROW;
  r.sj = Io<ST>::ld4(a.h, hj + 4 * g);
  r.si = Io<ST>::ld4(a.h, hi + 4 * g);
  r.d_ms = *reinterpret_cast<const f4*>(gd + 4 * g);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    r.vj[d] = Io<ST>::ld(a.h, hj + NS + 3 * g + d);
    r.vi[d] = Io<ST>::ld(a.h, hi + NS + 3 * g + d);
    r.d_mv[d] = gd[NS + 3 * g + d];
  }
  if (a.mean) {
    const int deg = a.rowptr[dst + 1] - a.rowptr[dst];
    const float sc = 1.0f / (float)(deg > 1 ? deg : 1);
    r.d_ms *= sc;
#pragma unroll
    for (int d = 0; d < 3; ++d) r.d_mv[d] *= sc;
  }
}

template <int NTE, typename ST, int TN, int C2_WPB>
__global__ __launch_bounds__(WAVE * C2_WPB) __attribute__((amdgpu_waves_per_eu(C2_WPB / 4, C2_WPB / 4))) void conv_bwd2_kernel(ConvBArgs a) {
  WALL_STAMP(2);
  typedef Image<0, NTE> IM;
  typedef ConvBImg<NTE> BI;
  typedef ConvBlk<NTE> B;
  typedef C2Shape<TN, C2_WPB> SH;
  constexpr int C2_TPB = SH::TPB, C2_GS_ROWS = SH::GS_ROWS, C2_SCR = SH::SCR;
  constexpr int PW = B::SIZE + C2_SCR;
  constexpr bool BF = Io<ST>::BF;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds + C2_TPB - BI::F0;
  float* imgT = lds + C2_TPB + BI::FSIZE - BI::T0;
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* blocks = lds + C2_TPB + BI::FSIZE + BI::TSIZE;
  float* gblk = blocks + w * PW;
  float* tscr = gblk + B::SIZE;
  float* trow = tscr + TN * TSCR_FLOATS;                                   // [C2_GS_ROWS][ROW]
  int* tsrc = reinterpret_cast<int*>(trow + C2_GS_ROWS * ROW);
  stage_slice<BI::FSIZE, C2_TPB>(lds + C2_TPB, a.img + BI::F0, threadIdx.x);
  stage_slice<BI::TSIZE, C2_TPB>(lds + C2_TPB + BI::FSIZE, a.imgT + BI::T0, threadIdx.x);
  for (int k = lane; k < B::SIZE / 4; k += WAVE) reinterpret_cast<f4*>(gblk)[k] = f4{0.f, 0.f, 0.f, 0.f};

  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const int64_t waves = (int64_t)gridDim.x * C2_WPB, wid = (int64_t)blockIdx.x * C2_WPB + w;
  // ---- owned NODES (an even split, independent of the edge ranges): targets without incoming edges get their zero row
  {
    const int64_t npw = (a.N + waves - 1) / waves;
    const int64_t n0 = wid * npw;
    const int64_t nn = n0 >= a.N ? 0 : (a.N - n0 < npw ? a.N - n0 : npw);
    for (int64_t k = lane; k < nn * ROW; k += WAVE) {
      const int64_t nd = n0 + k / ROW;
      if (a.rowptr[nd + 1] == a.rowptr[nd]) a.g_dst[n0 * ROW + k] = 0.f;
    }
  }
  // ---- owned EDGES: whole iterations of TN * 16 sorted positions, contiguous per wave
  const int32_t E = a.rowptr[a.N];                     // valid sorted edges (dropped ones sit behind them)
  const int32_t iters = (E + TN * TILE - 1) / (TN * TILE);
  const int32_t ipw = (int32_t)((iters + waves - 1) / waves);
  const int64_t it0 = wid * ipw;
  const int32_t c_lo = (int32_t)(it0 < iters ? it0 : iters) * (TN * TILE);
  int32_t c_hi = (int32_t)(it0 + ipw < iters ? it0 + ipw : iters) * (TN * TILE);
  c_hi = c_hi < E ? c_hi : E;
  // the target whose edges began in an earlier wave's range (its partial sum here goes to g_src by atomics)
  int32_t lead_dst = -1;
  if (c_lo > 0 && c_lo < c_hi) {
    const int32_t d0 = a.edst[c_lo], dm = a.edst[c_lo - 1];
    lead_dst = d0 == dm ? d0 : -1;
  }
  typename Msg0<ST>::WAcc w0;
  typename Msg1<ST>::WAcc w1;
  typename Msg2<ST>::WAcc w2;
  Msg0<ST>::wacc_zero(w0);
  Msg1<ST>::wacc_zero(w1);
  Msg2<ST>::wacc_zero(w2);
  float carry[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int carry_dst = -1;
  // ---- the wave's first iteration: indices, then rows -- issued BEFORE the workgroup waits for its fragment images (the
  // one exposed gather; every later one is issued an iteration ahead)
  const int32_t p_last = c_hi - 1;                         // (only used when the range is not empty)
  // With two waves per SIMD the partner wave covers the gather and the registers are scarce: no gather-ahead there.
  constexpr bool PREFETCH = C2_WPB == 4;
  int32_t csrc[TN], cdst[TN];
  C2Rows cur[TN];
  if (PREFETCH && c_lo < c_hi) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int32_t p = c_lo + j * TILE + i, pc = p < p_last ? p : p_last;
      csrc[j] = a.esrc[pc];
      cdst[j] = a.edst[pc];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int32_t p = c_lo + j * TILE + i, pc = p < p_last ? p : p_last;
      c2_load_rows<ST>(a, pc, csrc[j], cdst[j], g, cur[j]);
    }
  }
  STAMP(0);
  __syncthreads();
  STAMP(1);

  const int zt[TN] = {};
  for (int32_t base = c_lo; base < c_hi; base += TN * TILE) {
    STAMP(15);
    // ---- indices of the NEXT iteration (clamped: harmless on the last one); its rows are issued after the data gradients
    const int32_t nxt_base = base + TN * TILE;
    int32_t nsrc[TN], ndst[TN], npc[TN];
    if constexpr (PREFETCH) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int32_t p = nxt_base + j * TILE + i;
        npc[j] = p < p_last ? p : p_last;
        nsrc[j] = a.esrc[npc[j]];
        ndst[j] = a.edst[npc[j]];
      }
    } else {
      int32_t pc[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int32_t p = base + j * TILE + i;
        pc[j] = p < p_last ? p : p_last;
        csrc[j] = a.esrc[pc[j]];
        cdst[j] = a.edst[pc[j]];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) c2_load_rows<ST>(a, pc[j], csrc[j], cdst[j], g, cur[j]);
    }
    f4 es0[TN], es1[TN], sj[TN], si[TN], d_ms[TN];
    float ev[TN][3], vj[TN][3], vi[TN][3], d_mv[TN][3];
    int32_t src[TN], dst[TN];
    bool active[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      active[j] = base + j * TILE + i < c_hi;
      src[j] = csrc[j];
      dst[j] = active[j] ? cdst[j] : -1;
      es0[j] = cur[j].es0; es1[j] = cur[j].es1; sj[j] = cur[j].sj; si[j] = cur[j].si;
      d_ms[j] = active[j] ? cur[j].d_ms : zero;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        ev[j][d] = g == 0 ? cur[j].ev[d] : 0.f;             // the edge vector sits in group 0's k-slot only
        vj[j][d] = cur[j].vj[d];
        vi[j][d] = cur[j].vi[d];
        d_mv[j][d] = active[j] ? cur[j].d_mv[d] : 0.f;
      }
    }
    STAMP(2);
    // ---- recompute the three message GVPs of both tiles
    float b0[TN][16], bv0[TN][3][3], b1[TN][4], bv1[TN][3][1], b2[TN][4], bv2[TN][3][1];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { b0[j][r] = sj[j][r]; b0[j][4 + r] = es0[j][r]; b0[j][8 + r] = es1[j][r]; b0[j][12 + r] = si[j][r]; }
#pragma unroll
      for (int d = 0; d < 3; ++d) { bv0[j][d][0] = vj[j][d]; bv0[j][d][1] = vi[j][d]; bv0[j][d][2] = ev[j][d]; }
    }
    f4 s1[TN][1], s2[TN][1], s3[TN][1];
    float v1[TN][3][1], v2[TN][3][1], v3[TN][3][1];
    typename Msg0<ST>::Cache c0[TN];
    typename Msg1<ST>::Cache c1[TN];
    typename Msg2<ST>::Cache c2[TN];
    Msg0<ST>::template forward<TN, BF>(img + IM::CV_M0, lane, zt, b0, bv0, s1, v1, c0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) b1[j][r] = s1[j][0][r];
#pragma unroll
      for (int d = 0; d < 3; ++d) bv1[j][d][0] = v1[j][d][0];
    }
    Msg1<ST>::template forward<TN, BF>(img + IM::CV_M1, lane, zt, b1, bv1, s2, v2, c1);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) b2[j][r] = s2[j][0][r];
#pragma unroll
      for (int d = 0; d < 3; ++d) bv2[j][d][0] = v2[j][d][0];
    }
    Msg2<ST>::template forward<TN, BF>(img + IM::CV_M2, lane, zt, b2, bv2, s3, v3, c2);
    STAMP(3);
    // ---- backward through the three message GVPs (inactive lanes carry d_ms = d_mv = 0, so every gradient below is zero there)
    float d_b[TN][4], d_bv[TN][3][1];
    {
      f4 d_so[TN][1];
      float d_vo[TN][3][1];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        d_so[j][0] = d_ms[j];
#pragma unroll
        for (int d = 0; d < 3; ++d) d_vo[j][d][0] = d_mv[j][d];
      }
      typename Msg2<ST>::Grads gr[TN];
      Msg2<ST>::template backward_tn<TN, BF>(imgT + IM::TC_M2, lane, c2, d_so, d_vo, d_b, d_bv, gr);
      STAMP(10);
      if constexpr (!PREFETCH) {            // two waves per SIMD: b2 / bv2 were dropped after the forward, rebuilt from msg1's cache
        float b2r[TN][4], bv2r[TN][3][1];
#pragma unroll
        for (int j = 0; j < TN; ++j) Msg1<ST>::outputs_from_cache(c1[j], b2r[j], bv2r[j]);
        Msg2<ST>::template wacc_accumulate<TN, BF>(w2, lane, zt, b2r, bv2r, c2, gr, tscr);
      } else {
        Msg2<ST>::template wacc_accumulate<TN, BF>(w2, lane, zt, b2, bv2, c2, gr, tscr);
      }
    }
    {
      f4 d_so[TN][1];
      float d_vo[TN][3][1];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        d_so[j][0] = f4{d_b[j][0], d_b[j][1], d_b[j][2], d_b[j][3]};
#pragma unroll
        for (int d = 0; d < 3; ++d) d_vo[j][d][0] = d_bv[j][d][0];
      }
      typename Msg1<ST>::Grads gr[TN];
      STAMP(11);
      Msg1<ST>::template backward_tn<TN, BF>(imgT + IM::TC_M1, lane, c1, d_so, d_vo, d_b, d_bv, gr);
      STAMP(12);
      if constexpr (!PREFETCH) {
        float b1r[TN][4], bv1r[TN][3][1];
#pragma unroll
        for (int j = 0; j < TN; ++j) Msg0<ST>::outputs_from_cache(c0[j], b1r[j], bv1r[j]);
        Msg1<ST>::template wacc_accumulate<TN, BF>(w1, lane, zt, b1r, bv1r, c1, gr, tscr);
      } else {
        Msg1<ST>::template wacc_accumulate<TN, BF>(w1, lane, zt, b1, bv1, c1, gr, tscr);
      }
    }
    float d_b0[TN][16], d_bv0[TN][3][3];
    {
      f4 d_so[TN][1];
      float d_vo[TN][3][1];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        d_so[j][0] = f4{d_b[j][0], d_b[j][1], d_b[j][2], d_b[j][3]};
#pragma unroll
        for (int d = 0; d < 3; ++d) d_vo[j][d][0] = d_bv[j][d][0];
      }
      typename Msg0<ST>::Grads gr[TN];
      STAMP(13);
      Msg0<ST>::template backward_tn<TN, BF>(imgT + IM::TC_M0, lane, c0, d_so, d_vo, d_b0, d_bv0, gr);
      STAMP(14);
      // rows of the next iteration: in flight behind the weight gradients and the tail of this one
      if constexpr (PREFETCH) {
#pragma unroll
        for (int j = 0; j < TN; ++j) c2_load_rows<ST>(a, npc[j], nsrc[j], ndst[j], g, cur[j]);
      }
      Msg0<ST>::template wacc_accumulate<TN, BF>(w0, lane, zt, b0, bv0, c0, gr, tscr);
    }
    // first target of the next iteration (-1 at the end of the wave's range): where the last segment of this one ends
    int32_t next_first = -1;
    if (nxt_base < c_hi) next_first = PREFETCH ? __builtin_amdgcn_readfirstlane(ndst[0]) : a.edst[nxt_base];
    STAMP(4);
    // ---- d(edge embedding) of this layer -> g_e (plain stores, sorted-edge order)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (active[j]) {          // in the storage type (bf16 storage: the rows the edge stage re-reads are half the bytes)
        const int64_t gr_ = (int64_t)(base + j * TILE + i) * EROW;
        Io<ST>::st4(a.g_e, gr_ + 4 * g, f4{d_b0[j][4], d_b0[j][5], d_b0[j][6], d_b0[j][7]});
        Io<ST>::st4(a.g_e, gr_ + 16 + 4 * g, f4{d_b0[j][8], d_b0[j][9], d_b0[j][10], d_b0[j][11]});
        if (g == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) Io<ST>::st(a.g_e, gr_ + ES + d, d_bv0[j][d][2]);
        }
      }
    STAMP(5);
    // ---- d h[src]: rows transposed through LDS so that one wave-instruction adds two whole 112-B rows; a self loop's row
    // joins its target's sum below instead
    bool self[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      self[j] = active[j] && src[j] == dst[j];
      float* tr = trow + (j * TILE + i) * ROW;
      *reinterpret_cast<f4*>(tr + 4 * g) = f4{d_b0[j][0], d_b0[j][1], d_b0[j][2], d_b0[j][3]};
#pragma unroll
      for (int d = 0; d < 3; ++d) tr[NS + 3 * g + d] = d_bv0[j][d][0];
      if (g == 0) tsrc[j * TILE + i] = (active[j] && !self[j]) ? src[j] : -1;
    }
    __builtin_amdgcn_wave_barrier();
    {
      const int half = lane / ROW, col = lane - half * ROW;      // lanes 0..27 -> row 2k, 28..55 -> row 2k+1
      const int hr = half & 1;                                   // (lanes 56..63 read row 2k too and add nothing)
      int sr[C2_GS_ROWS / 2];
      float val[C2_GS_ROWS / 2];
#pragma unroll
      for (int k = 0; k < C2_GS_ROWS / 2; ++k) {                 // all the LDS reads first: two round trips, not 32
        sr[k] = tsrc[2 * k + hr];
        val[k] = trow[(2 * k + hr) * ROW + (half < 2 ? col : 0)];
      }
#pragma unroll
      for (int k = 0; k < C2_GS_ROWS / 2; ++k)
        if (half < 2 && sr[k] >= 0) atomicAdd(a.g_src + (int64_t)sr[k] * ROW + col, val[k]);
    }
    __builtin_amdgcn_wave_barrier();
    STAMP(6);
    // ---- d h[dst]: segment sums of the sorted edges (DPP scan + a register carry across tiles), stored where a segment
    // truly ends inside the wave's range
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float x[7] = {d_b0[j][12], d_b0[j][13], d_b0[j][14], d_b0[j][15], d_bv0[j][0][1], d_bv0[j][1][1], d_bv0[j][2][1]};
      if (self[j]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] += d_b0[j][k];
#pragma unroll
        for (int d = 0; d < 3; ++d) x[4 + d] += d_bv0[j][d][0];
      }
      if (i == 0 && active[j] && dst[j] == carry_dst) {
#pragma unroll
        for (int k = 0; k < 7; ++k) x[k] += carry[k];
      }
      seg_scan16<7>(dst[j], x);
      int nxt = __builtin_amdgcn_update_dpp(-1, dst[j], 0x100 | 1, 0xf, 0xf, false);        // lane i <- lane i + 1
      if (i == TILE - 1) nxt = next_first;
      if (j + 1 < TN) {
        const int nf = __builtin_amdgcn_update_dpp(-1, dst[j + 1 < TN ? j + 1 : j], 0x110 | 15, 0xf, 0xf, false);   // lane 15 <- lane 0 of the next tile
        if (i == TILE - 1) nxt = nf;
      }
      if (active[j] && nxt != dst[j]) {
        if (dst[j] != lead_dst) {
          float* row = a.g_dst + (int64_t)dst[j] * ROW;
          *reinterpret_cast<f4*>(row + 4 * g) = f4{x[0], x[1], x[2], x[3]};
#pragma unroll
          for (int d = 0; d < 3; ++d) row[NS + 3 * g + d] = x[4 + d];
        } else {
          float* row = a.g_src + (int64_t)dst[j] * ROW;
#pragma unroll
          for (int k = 0; k < 4; ++k) atomicAdd(row + 4 * g + k, x[k]);
#pragma unroll
          for (int d = 0; d < 3; ++d) atomicAdd(row + NS + 3 * g + d, x[4 + d]);
        }
      }
      carry_dst = __builtin_amdgcn_update_dpp(-1, dst[j], 0x100 | 15, 0xf, 0xf, false);       // lane 0 <- lane 15
#pragma unroll
      for (int k = 0; k < 7; ++k)
        carry[k] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[k]), 0x100 | 15, 0xf, 0xf, false));
    }
    if constexpr (PREFETCH) {
#pragma unroll
      for (int j = 0; j < TN; ++j) { csrc[j] = nsrc[j]; cdst[j] = ndst[j]; }
    }
    STAMP(7);
  }
  STAMP(8);
  // every element of the (zeroed) block is the target of at most one flush: plain stores, no read-add-write round trips
  Msg0<ST>::template wacc_flush<AccStoreOnce>(w0, gblk + B::M0, lane);
  Msg1<ST>::template wacc_flush<AccStoreOnce>(w1, gblk + B::M1, lane);
  Msg2<ST>::template wacc_flush<AccStoreOnce>(w2, gblk + B::M2, lane);
  write_slab_row<B::SIZE, PW, C2_WPB>(a.slab, blocks);
  STAMP(9);
}

// ===================================================================== edge embedding
// Backward of gvp_edge = Sequential(GVP, LayerNorm) (protein_gnn.py:331-335, :376) for a tile of 16 sorted edges,
// ONCE per step: the upstream gradient is the sum of the conv layers' d e buffers.  Raw edge features get no
// gradient, so only what the weight gradients need is back-propagated.
// Gradient block: [gvp_edge.0 | gvp_edge.1 (gamma, beta)].
template <int NTE>
struct EdgeBlk {
  static constexpr int GVP = 0, LN = LEdgeGvp::size(NTE), SIZE = pad4(LN + 2 * ES);
};
constexpr int EB_MAX_LAYERS = 8;
struct EdgeBArgs {
  const float* img; const float* imgT;
  const float* e_s; const float* e_v; const int64_t* etypes; const int32_t* eperm; int64_t E;
  const float* g_e[EB_MAX_LAYERS]; int n_g; float* slab;
  float* g_e_s; float* g_e_v;     // optional: d(loss)/d(raw edge features), fp32, ORIGINAL edge order (DX instantiation)
};
template <int NTE>
constexpr int edge_bwd_lds_floats() { return BW_TPB + Image<0, NTE>::CV_M0 + Image<0, NTE>::TC_M0 + BW_WPB * (EdgeBlk<NTE>::SIZE + TSCR_FLOATS); }

template <int NTE, typename ST, bool DX>
__device__ __forceinline__ void edge_bwd_body(const EdgeBArgs& a, float* lds, int bid, int nblk) {
  typedef Image<0, NTE> IM;
  typedef EdgeBlk<NTE> B;
  constexpr int PW = B::SIZE + TSCR_FLOATS;
  float* img = lds + BW_TPB;                      // [QEdge | edge LN]  (= the first CV_M0 floats of the conv slice)
  float* imgT = img + IM::CV_M0;                  // QEdge transposed  (= the first TC_M0 floats of the convT slice)
  const int lane0 = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* blocks = imgT + IM::TC_M0;
  float* gblk = blocks + w * PW;
  float* tscr = gblk + B::SIZE;
  stage_slice<IM::CV_M0, BW_TPB>(img, a.img, threadIdx.x);
  stage_slice<IM::TC_M0, BW_TPB>(imgT, a.imgT, threadIdx.x);
  for (int k = lane0; k < B::SIZE / 4; k += WAVE) reinterpret_cast<f4*>(gblk)[k] = f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const bool first = false;
  const int64_t tiles = (a.E + TILE - 1) / TILE;
  // gvp_edge's weight gradients stay in registers over all of the wave's tiles (GvpQ::WAcc, as in the conv backward: a
  // wave walks ~20 tiles at long_graph_x64) and go to the wave's LDS block once, after the loop; inactive lanes carry
  // zero gradients (d_es / d_ev are zero there and the LayerNorm backward is linear in them)
  typename QEdge<NTE>::WAcc wacc;
  QEdge<NTE>::wacc_zero(wacc);
  for (int64_t t = (int64_t)w * nblk + bid; t < tiles; t += (int64_t)nblk * BW_WPB) {
    const int lane = opaque_lane(lane0), i = lane & 15, g = lane >> 4;
    const int64_t p = t * TILE + i;
    // positions behind the last VALID edge hold eperm = -1 (edges with out-of-range endpoints are dropped by the CSR
    // build, which leaves E - rowptr[N] unused positions at the end of the sorted tables)
    const int32_t eid = p < a.E ? a.eperm[p] : -1;
    const bool active = eid >= 0;
    f4 es0 = zero, es1 = zero;
    float ev[3] = {0.f, 0.f, 0.f};
    int et[1] = {0};
    f4 d_es[2] = {zero, zero};
    float d_ev[3][1] = {{0.f}, {0.f}, {0.f}};
    if (active) {
      const int64_t er = (int64_t)eid * EDGE_IN_S;
      es0 = Io<ST>::ld4(a.e_s, er + 4 * g);
      es1 = Io<ST>::ld4(a.e_s, er + 16 + 4 * g);
      if (g == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) ev[d] = Io<ST>::ld(a.e_v, (int64_t)eid * 3 + d);
      }
      if (NTE > 0) {
        et[0] = (int)a.etypes[eid];
        et[0] = et[0] < 0 ? 0 : (et[0] >= NTE ? NTE - 1 : et[0]);
      }
      for (int l = 0; l < a.n_g; ++l) {                    // sum of the conv layers' d(edge embedding)
        const int64_t gr_ = p * EROW;
        d_es[0] += Io<ST>::ld4(a.g_e[l], gr_ + 4 * g);
        d_es[1] += Io<ST>::ld4(a.g_e[l], gr_ + 16 + 4 * g);
        if (g == 0) {
#pragma unroll
          for (int d = 0; d < 3; ++d) d_ev[d][0] += Io<ST>::ld(a.g_e[l], gr_ + ES + d);
        }
      }
    }
    float bse[1][8], bve[1][3][1];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bse[0][r] = es0[r]; bse[0][4 + r] = es1[r]; }
#pragma unroll
    for (int d = 0; d < 3; ++d) bve[0][d][0] = ev[d];
    f4 e_pre[1][2];
    float ev_pre[1][3][1];
    typename QEdge<NTE>::Cache ce[1];
    QEdge<NTE>::template forward<1, Io<ST>::BF>(img + IM::CV_EDGE, lane, et, bse, bve, e_pre, ev_pre, ce);
    f4 dga[2], dbe[2];
    ln_quad_bwd<ES, EV>(img + IM::CV_ELN, lane, e_pre[0], ev_pre[0], d_es, d_ev, dga, dbe);
    ln_param_grads<AccPriv, ES>(gblk + B::LN, first, lane, active, dga, dbe);
    float d_in[8], d_inv[3][1];
    typename QEdge<NTE>::Grads gr[1];
    QEdge<NTE>::template backward<Io<ST>::BF>(imgT + IM::TC_EDGE, lane, ce[0], d_es, d_ev, d_in, d_inv, gr[0]);
    if (DX && active) {               // gradients of the raw features: every valid edge id is written exactly once
      const int64_t er = (int64_t)eid * EDGE_IN_S;
      *reinterpret_cast<f4*>(a.g_e_s + er + 4 * g) = f4{d_in[0], d_in[1], d_in[2], d_in[3]};
      *reinterpret_cast<f4*>(a.g_e_s + er + 16 + 4 * g) = f4{d_in[4], d_in[5], d_in[6], d_in[7]};
      if (g == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) a.g_e_v[(int64_t)eid * 3 + d] = d_inv[d][0];
      }
    }
    QEdge<NTE>::template wacc_accumulate<1, Io<ST>::BF>(wacc, lane, et, bse, bve, ce, gr, tscr);
  }
  QEdge<NTE>::template wacc_flush<AccStoreOnce>(wacc, gblk + B::GVP, lane0);
  write_slab_row<B::SIZE, PW>(a.slab, blocks, bid);
}
template <int NTE, typename ST, bool DX = false>
__global__ __launch_bounds__(BW_TPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void edge_bwd_kernel(EdgeBArgs a) {
  WALL_STAMP(3);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  edge_bwd_body<NTE, ST, DX>(a, lds, blockIdx.x, gridDim.x);
}

// ===================================================================== node embed
template <int NTN>
struct EmbBlk {
  static constexpr int GVP = 0, LN = LNodeGvp::size(NTN), SIZE = pad4(LN + 2 * NS);
};

struct EmbBArgs {
  const float* img; const float* imgT; const float* x_s; const float* x_v; const int64_t* ntypes; int64_t N;
  const float* g_up0; const float* g_up1; const float* g_up2; float* g_x_s; float* g_x_v; float* slab;
};

template <int NTN>
constexpr int embed_bwd_lds_floats() { return BW_TPB + Image<NTN, 0>::EMB_SIZE + Image<NTN, 0>::TE_SIZE + BW_WPB * (EmbBlk<NTN>::SIZE + TSCR_FLOATS); }

template <int NTN, typename ST, bool ONE = false>
__device__ __forceinline__ void embed_bwd_body(const EmbBArgs& a, float* lds, int bid, int nblk) {
  typedef Image<NTN, 0> IM;
  typedef AccFor<ONE> Acc;
  typedef QNode<NTN> Q;
  typedef EmbBlk<NTN> B;
  float* img = lds + BW_TPB;
  float* imgT = img + IM::EMB_SIZE;
  float* blocks = imgT + IM::TE_SIZE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* gblk = blocks + w * B::SIZE;                               // this wave's private gradient block
  float* tscr = blocks + BW_WPB * B::SIZE + w * TSCR_FLOATS;
  stage_slice<IM::EMB_SIZE, BW_TPB>(img, a.img, threadIdx.x);
  stage_slice<IM::TE_SIZE, BW_TPB>(imgT, a.imgT, threadIdx.x);
  zero_block<B::SIZE>(gblk, lane);
  __syncthreads();
  const int i = lane & 15, g = lane >> 4;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const int64_t ntiles = (a.N + TILE - 1) / TILE;
  const bool first = false;
  for (int64_t tile = (int64_t)w * nblk + bid; tile < ntiles; tile += (int64_t)nblk * BW_WPB) {
    const int64_t n = tile * TILE + i;
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
    f4 s_pre[1][1];
    float v_pre[1][3][1];
    typename Q::Cache c[1];
    Q::template forward<1, Io<ST>::BF>(img + IM::EMB_GVP, lane, type, bs, bv, s_pre, v_pre, c);
    f4 gs[1] = {zero};
    float gv[3][1] = {{0.f}, {0.f}, {0.f}};
    if (active) {
      const float* ups[3] = {a.g_up0, a.g_up1, a.g_up2};
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (ups[u]) {
          const float* r_ = ups[u] + n * ROW;
          gs[0] += *reinterpret_cast<const f4*>(r_ + 4 * g);
#pragma unroll
          for (int p = 0; p < 3; ++p) gv[p][0] += r_[NS + 3 * g + p];
        }
    }
    f4 dga[1], dbe[1];
    ln_quad_bwd<NS, NV>(img + IM::EMB_LN, lane, s_pre[0], v_pre[0], gs, gv, dga, dbe);
    ln_param_grads<Acc, NS>(gblk + B::LN, first, lane, active, dga, dbe);
    float d_bs[Q::SSTEPS], d_bv[3][1];
    typename Q::Grads gr;
    Q::template backward<Io<ST>::BF>(imgT, lane, c[0], gs, gv, d_bs, d_bv, gr);
    Q::template weight_grads<Acc, Io<ST>::BF>(gblk + B::GVP, first, lane, type[0], active, bs[0], bv[0], c[0], gr, tscr);
    if (active && a.g_x_s) {
#pragma unroll
      for (int s = 0; s < Q::SSTEPS; ++s) {
        const int cc = 4 * s + g;
        if (cc < NODE_IN_S) a.g_x_s[n * NODE_IN_S + cc] = d_bs[s];
      }
      if (g < NODE_IN_V) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a.g_x_v[n * 3 * NODE_IN_V + 3 * g + p] = d_bv[p][0];
      }
    }
  }
  write_slab_row<B::SIZE, B::SIZE>(a.slab, blocks, bid);
}
template <int NTN, typename ST, bool ONE = false>
__global__ __launch_bounds__(BW_TPB, 2) void embed_bwd_kernel(EmbBArgs a) {
  WALL_STAMP(4);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  embed_bwd_body<NTN, ST, ONE>(a, lds, blockIdx.x, gridDim.x);
}

// dst[j] += sum_r slab[r][col0 + j], j < len.  A block owns 64 columns; its 16
// row groups each sum rows rg, rg+16, ... (coalesced 256-B reads) and the 16
// partials are added in a fixed order: deterministic, and ~16 dependent loads per
// thread instead of `rows`.
constexpr int RED_COLS = 64, RED_RG = 16;
__global__ __launch_bounds__(RED_COLS * RED_RG) void reduce_slab_kernel(const float* __restrict__ slab, int rows,
                                                                        int stride, int col0, int len,
                                                                        float* __restrict__ dst) {
  __shared__ float part[RED_RG][RED_COLS];
  const int c = threadIdx.x & (RED_COLS - 1), rg = threadIdx.x / RED_COLS;
  const int j = blockIdx.x * RED_COLS + c;
  float s = 0.f;
  if (j < len)
    for (int r = rg; r < rows; r += RED_RG) s += slab[(size_t)r * stride + col0 + j];
  part[rg][c] = s;
  __syncthreads();
  if (rg == 0 && j < len) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < RED_RG; ++k) t += part[k][c];
    dst[j] += t;
  }
}

__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ slab, int rows, int stride, int col0,
                                                          int len, float* __restrict__ dst) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= len) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += slab[(size_t)r * stride + col0 + j];
  dst[j] += s;
}

struct SegTable { cgvp_segment s[CGVP_MAX_SEGS]; };
__global__ __launch_bounds__(RED_COLS * RED_RG) void reduce_segments_kernel(SegTable t, float* __restrict__ grad, int overwrite) {
  __shared__ float part[RED_RG][RED_COLS];
  const cgvp_segment sg = t.s[blockIdx.y];
  const int c = threadIdx.x & (RED_COLS - 1), rg = threadIdx.x / RED_COLS;
  const int j = blockIdx.x * RED_COLS + c;
  if (blockIdx.x * RED_COLS >= sg.len) return;          // whole block out of range (uniform)
  // four independent partial sums: the row loads of a thread are 4 deep in flight instead of one dependent add at a time
  float s = 0.f;
  if (j < sg.len) {
    const float* col = sg.slab + sg.col0 + j;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = rg;
    for (; r + 3 * RED_RG < sg.rows; r += 4 * RED_RG) {
      s0 += col[(size_t)r * sg.stride];
      s1 += col[(size_t)(r + RED_RG) * sg.stride];
      s2 += col[(size_t)(r + 2 * RED_RG) * sg.stride];
      s3 += col[(size_t)(r + 3 * RED_RG) * sg.stride];
    }
    for (; r < sg.rows; r += RED_RG) s0 += col[(size_t)r * sg.stride];
    s = (s0 + s1) + (s2 + s3);
  }
  part[rg][c] = s;
  __syncthreads();
  if (rg == 0 && j < sg.len) {
    float tt = 0.f;
#pragma unroll
    for (int k = 0; k < RED_RG; ++k) tt += part[k][c];
    if (overwrite) grad[sg.dst + j] = tt;     // the caller guarantees disjoint destinations that cover what it reads
    else atomicAdd(grad + sg.dst + j, tt);    // ADD semantics of the per-call reductions
  }
}

// the same for segments of a handful of rows (the GINE backward on <= 32 workgroups): one thread per column, small
// workgroups that fit beside the protein backward's CU-filling kernels
__global__ __launch_bounds__(256) void reduce_segments_rows_kernel(SegTable t, float* __restrict__ grad, int overwrite) {
  const cgvp_segment sg = t.s[blockIdx.y];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= sg.len) return;
  const float* col = sg.slab + sg.col0 + j;
  float s = 0.f;
  for (int r = 0; r < sg.rows; ++r) s += col[(size_t)r * sg.stride];
  if (overwrite) grad[sg.dst + j] = s;
  else atomicAdd(grad + sg.dst + j, s);
}

// Workgroup cap of the backward kernels.  BW_MAX_GRID (240: slab rows are sized for it) when the launch is
// throughput-bound; 8 fewer when `units` (tiles, or pairs of tiles for the conv backward) fit the smaller grid's waves
// anyway -- the launch is then bound by ONE wave's latency, the spare CUs cost nothing and the drug encoder's kernels on
// the side stream find room (davis_b64: 0.2353 -> 0.2320 ms per step, long_graph_x64 unchanged).
// CGVP_BWD_GRID=<n> in the environment (read once) overrides both: an A/B knob.
inline int bwd_grid_cap(int64_t units) {
  static const int forced = [] {
    const char* e = getenv("CGVP_BWD_GRID");
    const int v = e ? atoi(e) : 0;
    return (v >= 1 && v <= BW_MAX_GRID) ? v : 0;
  }();
  if (forced) return forced;
  constexpr int SMALL = BW_MAX_GRID - 8;
  return units <= (int64_t)SMALL * BW_WPB ? SMALL : BW_MAX_GRID;
}
// CGVP_BWD_STORE_ONCE=0 in the environment (read once) keeps the read-add-write accumulation at every size: A/B knob
inline bool one_tile_per_wave(int64_t tiles, int grid) {
  static const bool on = [] { const char* e = getenv("CGVP_BWD_STORE_ONCE"); return !(e && e[0] == '0'); }();
  return on && tiles <= (int64_t)grid * BW_WPB;
}
inline int grid_for(int64_t tiles) {          // workgroups = slab rows; tiles go round-robin over them
  const int cap = bwd_grid_cap(tiles);
  return (int)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}

}  // namespace

namespace quad {

int bwd_block_sizes(int nt_node, int nt_edge, int* emb, int* conv_edge, int* conv_total, int* node, int* head) {
  if (nt_node == 0) *emb = EmbBlk<0>::SIZE; else if (nt_node == 20) *emb = EmbBlk<20>::SIZE;
  else if (nt_node == 21) *emb = EmbBlk<21>::SIZE; else return CGVP_ERR_UNSUPPORTED_DIMS;
  if (nt_edge == 0) { *conv_edge = EdgeBlk<0>::SIZE; *conv_total = ConvBlk<0>::SIZE; }     // edge_bwd / conv_bwd blocks
  else if (nt_edge == 1) { *conv_edge = EdgeBlk<1>::SIZE; *conv_total = ConvBlk<1>::SIZE; }
  else return CGVP_ERR_UNSUPPORTED_DIMS;
  *node = NODE_GB;
  *head = HEAD_GB;
  return 0;
}

int reduce_segments(const cgvp_segment* segs, int nsegs, float* grad_params, hipStream_t st, int overwrite) {
  if (nsegs <= 0) return 0;
  SegTable t;
  int maxlen = 0;
  int maxrows = 0;
  for (int i = 0; i < nsegs; ++i) {
    t.s[i] = segs[i];
    maxlen = segs[i].len > maxlen ? segs[i].len : maxlen;
    maxrows = segs[i].rows > maxrows ? segs[i].rows : maxrows;
  }
  if (maxrows <= 32) {
    hipLaunchKernelGGL(reduce_segments_rows_kernel, dim3((maxlen + 255) / 256, nsegs), dim3(256), 0, st, t, grad_params, overwrite);
    return 0;
  }
  hipLaunchKernelGGL(reduce_segments_kernel, dim3((maxlen + RED_COLS - 1) / RED_COLS, nsegs), dim3(RED_COLS * RED_RG), 0,
                     st, t, grad_params, overwrite);
  return 0;
}

int reduce_slab(const float* slab, int rows, int stride, int col0, int len, float* dst, hipStream_t st) {
  if (rows <= 32) {            // a handful of rows: one thread per column, small workgroups that fit beside anything
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((len + 255) / 256), dim3(256), 0, st, slab, rows, stride, col0, len, dst);
    return 0;
  }
  hipLaunchKernelGGL(reduce_slab_kernel, dim3((len + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_RG), 0, st, slab, rows, stride, col0, len, dst);
  return 0;
}

// one launch of a backward kernel in the requested activation storage type
#define BWD_LAUNCH(KERNEL, G, TPB_, LDS_, ...)                                                                              \
  do {                                                                                                                        \
    if (bf16) {                                                                                                               \
      CGVP_SET_DYN_LDS_ONCE(KERNEL(bf16s), LDS_);                                                                             \
      hipLaunchKernelGGL((KERNEL(bf16s)), dim3(G), dim3(TPB_), (LDS_), st, __VA_ARGS__);                                            \
    } else {                                                                                                                  \
      CGVP_SET_DYN_LDS_ONCE(KERNEL(float), LDS_);                                                                             \
      hipLaunchKernelGGL((KERNEL(float)), dim3(G), dim3(TPB_), (LDS_), st, __VA_ARGS__);                                            \
    }                                                                                                                         \
  } while (0)

// the same for the two kernels that exist for every layer kind (`bf16` = tile policy index, gvp_internal.h)
#define BWD_LAUNCH_KIND(KERNEL, G, TPB_, LDS_, ...)                                                                         \
  do {                                                                                                                        \
    if (bf16 == POLICY_GVPDEF) {                                                                                              \
      CGVP_SET_DYN_LDS_ONCE(KERNEL(f32_gvpdef), LDS_);                                                                        \
      hipLaunchKernelGGL((KERNEL(f32_gvpdef)), dim3(G), dim3(TPB_), (LDS_), st, __VA_ARGS__);                                       \
    } else if (bf16 == POLICY_LINEAR) {                                                                                       \
      CGVP_SET_DYN_LDS_ONCE(KERNEL(f32_linear), LDS_);                                                                        \
      hipLaunchKernelGGL((KERNEL(f32_linear)), dim3(G), dim3(TPB_), (LDS_), st, __VA_ARGS__);                                       \
    } else {                                                                                                                  \
      BWD_LAUNCH(KERNEL, G, TPB_, LDS_, __VA_ARGS__);                                                                         \
    }                                                                                                                         \
  } while (0)

int node_update_bwd(const float* img_node, const float* imgT_node, const float* h, const float* dh,
                    const float* mask0, const float* mask1, gvp::RngArgs rng, const float* g_up0, const float* g_up1,
                    const float* g_up2, int64_t N, float* g_dh, float* g_h, float* zero_rows, float* slab, int* grid,
                    int bf16, hipStream_t st) {
  NodeBArgs a{img_node, imgT_node, h, dh, mask0, mask1, rng, g_up0, g_up1, g_up2, N, g_dh, g_h, zero_rows, slab};
  const int G = grid_for((N + TILE - 1) / TILE);
  *grid = G;
  const size_t lds = (size_t)node_bwd_lds_floats() * sizeof(float);
  if (one_tile_per_wave((N + TILE - 1) / TILE, G)) {
#define K_(ST) node_bwd_kernel<ST, true>
    BWD_LAUNCH_KIND(K_, G, BW_TPB, lds, a);
#undef K_
    return 0;
  }
#define K_(ST) node_bwd_kernel<ST>
  BWD_LAUNCH_KIND(K_, G, BW_TPB, lds, a);
#undef K_
  return 0;
}

// head + node stage of the last layer in one launch (see node_head_bwd_kernel); g_h_out must be the node stage's g_up0
int node_head_bwd(const float* img_head, const float* imgT_head, const float* h_out, const float* g_out, float* g_h_out,
                  float* head_slab, const float* img_node, const float* imgT_node, const float* h, const float* dh,
                  const float* mask0, const float* mask1, gvp::RngArgs rng, int64_t N, float* g_dh, float* g_h,
                  float* zero_rows, float* slab, int* grid, int bf16, hipStream_t st) {
  HeadBArgs hd{img_head, imgT_head, h_out, g_out, N, g_h_out, head_slab};
  NodeBArgs a{img_node, imgT_node, h, dh, mask0, mask1, rng, g_h_out, nullptr, nullptr, N, g_dh, g_h, zero_rows, slab};
  const int G = grid_for((N + TILE - 1) / TILE);
  *grid = G;
  const int f = node_bwd_lds_floats() > head_bwd_lds_floats() ? node_bwd_lds_floats() : head_bwd_lds_floats();
  const size_t lds = (size_t)f * sizeof(float);
  if (one_tile_per_wave((N + TILE - 1) / TILE, G)) {
#define K_(ST) node_head_bwd_kernel<ST, true>
    BWD_LAUNCH(K_, G, BW_TPB, lds, hd, a);
#undef K_
    return 0;
  }
#define K_(ST) node_head_bwd_kernel<ST>
  BWD_LAUNCH(K_, G, BW_TPB, lds, hd, a);
#undef K_
  return 0;
}

int head_bwd(const float* img_head, const float* imgT_head, const float* h_out, const float* g_out, int64_t N,
             float* g_h_out, float* slab, int* grid, int bf16, hipStream_t st) {
  HeadBArgs a{img_head, imgT_head, h_out, g_out, N, g_h_out, slab};
  const int G = grid_for((N + TILE - 1) / TILE);
  *grid = G;
  const size_t lds = (size_t)head_bwd_lds_floats() * sizeof(float);
#define K_(ST) head_bwd_kernel<ST>
  BWD_LAUNCH(K_, G, BW_TPB, lds, a);
#undef K_
  return 0;
}

// Which conv backward runs (CGVP_CONV_BWD in the environment, read once per process; the tests cross-check them):
//   3 (default)  conv_bwd2_kernel<.., TN = 1, 8 waves>: two waves per SIMD, one tile each, register-resident weight gradients,
//                edge-balanced ranges -- measured at davis_b64 / long_graph_x64: 35.3 / 434 us per launch
//   2            conv_bwd2_kernel<.., TN = 2, 4 waves>: one wave per SIMD, two tiles in lockstep, gather-ahead: 40.3 / 515 us
//   1            the round-3 kernel (target ranges, per-tile LDS flushes): 42.0 / 867 us
int conv_bwd_version() {
  static const int v = [] {
    const char* e = getenv("CGVP_CONV_BWD");
    return (e && e[0] == '1') ? 1 : ((e && e[0] == '2') ? 2 : 3);
  }();
  return v;
}

template <int NTE, int TN, int WPB_>
int conv_bwd2_impl(ConvBArgs& a, int64_t E, int* grid, int bf16, hipStream_t st) {
  typedef C2Shape<TN, WPB_> SH;
  const int64_t iters = (E + TN * TILE - 1) / (TN * TILE);
  const int64_t wgs = (iters + WPB_ - 1) / WPB_;
  const int cap = bwd_grid_cap(((E + TILE - 1) / TILE + 1) / 2);       // two tiles per wave are this kernel's normal load
  const int G = (int)(wgs < 1 ? 1 : (wgs > cap ? cap : wgs));
  *grid = G;
  const size_t lds = (size_t)SH::template lds_floats<NTE>() * sizeof(float);
#define K_(ST) conv_bwd2_kernel<NTE, ST, TN, WPB_>
  if (bf16 >= POLICY_GVPDEF) {
    if constexpr (NTE == 0) BWD_LAUNCH_KIND(K_, G, SH::TPB, lds, a);
    else return CGVP_ERR_UNSUPPORTED_DIMS;
    return 0;
  }
  BWD_LAUNCH(K_, G, SH::TPB, lds, a);
#undef K_
  return 0;
}

template <int NTE>
int conv_bwd_impl(ConvBArgs& a, int* grid, int bf16, hipStream_t st) {
  const int64_t ngroups = (a.N + a.npw - 1) / a.npw;
  const int64_t wgs = (ngroups + CB_WPB - 1) / CB_WPB;
  const int cap = bwd_grid_cap(ngroups);
  const int G = (int)(wgs < 1 ? 1 : (wgs > cap ? cap : wgs));
  *grid = G;
  const size_t lds = (size_t)conv_bwd_lds_floats<NTE>() * sizeof(float);
#define K_(ST) conv_bwd_kernel<NTE, ST>
  if (bf16 >= POLICY_GVPDEF) {
    if constexpr (NTE == 0) BWD_LAUNCH_KIND(K_, G, CB_TPB, lds, a);
    else return CGVP_ERR_UNSUPPORTED_DIMS;
    return 0;
  }
  BWD_LAUNCH(K_, G, CB_TPB, lds, a);
#undef K_
  return 0;
}

int conv_bwd(int nt_edge, const float* img, const float* imgT, const float* h, const float* e_emb,
             const int32_t* rowptr, const int32_t* esrc, const int32_t* edst, int64_t N, int64_t E, int mean,
             const float* g_dh, float* g_src, float* g_dst, float* g_e, float* slab, int* grid, int bf16, hipStream_t st) {
  // targets per wave: two 16-edge tiles' worth (the forward's 32-edge passes), as in quad::conv
  int64_t deg = N > 0 ? (E + N - 1) / N : 1;
  if (deg < 1) deg = 1;
  int npw = (int)((2 * TILE - 2) / deg);
  npw = npw < 1 ? 1 : (npw > WAVE ? WAVE : npw);
  ConvBArgs a{img, imgT, h, e_emb, rowptr, esrc, edst, N, npw, mean, g_dh, g_src, g_dst, g_e, slab};
  if (conv_bwd_version() == 2) {
    if (nt_edge == 0) return conv_bwd2_impl<0, 2, 4>(a, E, grid, bf16, st);
    if (nt_edge == 1) return conv_bwd2_impl<1, 2, 4>(a, E, grid, bf16, st);
    return CGVP_ERR_UNSUPPORTED_DIMS;
  }
  if (conv_bwd_version() == 3) {
    if (nt_edge == 0) return conv_bwd2_impl<0, 1, 8>(a, E, grid, bf16, st);
    if (nt_edge == 1) return conv_bwd2_impl<1, 1, 8>(a, E, grid, bf16, st);
    return CGVP_ERR_UNSUPPORTED_DIMS;
  }
  if (nt_edge == 0) return conv_bwd_impl<0>(a, grid, bf16, st);
  if (nt_edge == 1) return conv_bwd_impl<1>(a, grid, bf16, st);
  return CGVP_ERR_UNSUPPORTED_DIMS;
}

template <int NTE>
int edge_bwd_impl(EdgeBArgs& a, int* grid, int bf16, hipStream_t st) {
  // 66 KB of LDS and 128 VGPRs per workgroup (capped: 4-8 spilled): TWO workgroups share a CU (4 waves per SIMD), so up to 2 x 240 slab rows
  const int64_t tiles = (a.E + TILE - 1) / TILE;
  const int G = (int)(tiles < 1 ? 1 : (tiles > 2 * bwd_grid_cap((tiles + 1) / 2) ? 2 * bwd_grid_cap((tiles + 1) / 2) : tiles));
  *grid = G;
  const size_t lds = (size_t)edge_bwd_lds_floats<NTE>() * sizeof(float);
  if (a.g_e_s) {
#define K_(ST) edge_bwd_kernel<NTE, ST, true>
    BWD_LAUNCH(K_, G, BW_TPB, lds, a);
#undef K_
    return 0;
  }
#define K_(ST) edge_bwd_kernel<NTE, ST>
  BWD_LAUNCH(K_, G, BW_TPB, lds, a);
#undef K_
  return 0;
}

int edge_embed_bwd(int nt_edge, const float* img, const float* imgT, const float* e_s, const float* e_v,
                   const int64_t* etypes, const int32_t* eperm, int64_t E, const float* const* g_e, int n_g,
                   float* g_e_s, float* g_e_v, float* slab, int* grid, int bf16, hipStream_t st) {
  if (n_g < 1 || n_g > EB_MAX_LAYERS) return CGVP_ERR_BAD_ARG;
  if ((g_e_s == nullptr) != (g_e_v == nullptr)) return CGVP_ERR_BAD_ARG;
  EdgeBArgs a{img, imgT, e_s, e_v, etypes, eperm, E, {}, n_g, slab, g_e_s, g_e_v};
  for (int l = 0; l < n_g; ++l) a.g_e[l] = g_e[l];
  if (nt_edge == 0) return edge_bwd_impl<0>(a, grid, bf16, st);
  if (nt_edge == 1) return edge_bwd_impl<1>(a, grid, bf16, st);
  return CGVP_ERR_UNSUPPORTED_DIMS;
}

template <int NTN>
int embed_bwd_impl(EmbBArgs& a, int* grid, int bf16, hipStream_t st) {
  const int G = grid_for((a.N + TILE - 1) / TILE);
  *grid = G;
  const size_t lds = (size_t)embed_bwd_lds_floats<NTN>() * sizeof(float);
  if (one_tile_per_wave((a.N + TILE - 1) / TILE, G)) {
#define K_(ST) embed_bwd_kernel<NTN, ST, true>
    BWD_LAUNCH(K_, G, BW_TPB, lds, a);
#undef K_
    return 0;
  }
#define K_(ST) embed_bwd_kernel<NTN, ST>
  BWD_LAUNCH(K_, G, BW_TPB, lds, a);
#undef K_
  return 0;
}

int node_embed_bwd(int nt_node, const float* img, const float* imgT, const float* x_s, const float* x_v,
                   const int64_t* ntypes, int64_t N, const float* g_up0, const float* g_up1, const float* g_up2,
                   float* g_x_s, float* g_x_v, float* slab, int* grid, int bf16, hipStream_t st) {
  EmbBArgs a{img, imgT, x_s, x_v, ntypes, N, g_up0, g_up1, g_up2, g_x_s, g_x_v, slab};
  if (nt_node == 0) return embed_bwd_impl<0>(a, grid, bf16, st);
  if (nt_node == 20) return embed_bwd_impl<20>(a, grid, bf16, st);
  if (nt_node == 21) return embed_bwd_impl<21>(a, grid, bf16, st);
  return CGVP_ERR_UNSUPPORTED_DIMS;
}


}  // namespace quad
