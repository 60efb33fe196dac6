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
// Weight fragments are built by a tiny prep step (cgvp_lba_prepare / the first launch of a
// pass) into an "image" in global memory; each workgroup copies its slice into LDS with
// straight float4 loads.  Inside a GEMM's region the k-steps are grouped by four,
// lane-major (frag_decode below): one conflict-free ds_read_b128 at 16 B x lane fetches the A
// operands of four MFMAs (the last steps % 4: one ds_read_b32 each, step-major).
#pragma once
#include <hip/hip_runtime.h>

#include "gvp_math.h"

namespace gq {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int P1 = 0, P2 = 1;
constexpr int ceil4(int x) { return (x + 3) / 4; }

// ---- activation storage type (cgvp_dims.storage): what the ACTIVATION buffers in HBM hold -- node / edge features,
// node rows h / dh between stages, the edge-embedding store, the residue embeddings `out`.  fp32 (default) or bf16
// ("bf16 storage / fp32 accumulate": every load widens to fp32, all arithmetic, LayerNorm statistics, MFMA
// accumulation and every GRADIENT buffer stay fp32; stores round to nearest even).  The argument structs keep
// `float*` fields for both; Io<ST> reinterprets them, indices are in ELEMENTS.
struct bf16s {};
template <typename ST> struct Io;
template <> struct Io<float> {
  static constexpr bool BF = false;          // matrix-core operand type of the channel GEMMs: fp32 (16x16x4) / bf16 (16x16x16)
  static __device__ __forceinline__ f4 ld4(const float* p, int64_t i) { return *reinterpret_cast<const f4*>(p + i); }
  static __device__ __forceinline__ float ld(const float* p, int64_t i) { return p[i]; }
  static __device__ __forceinline__ void st4(float* p, int64_t i, f4 v) { *reinterpret_cast<f4*>(p + i) = v; }
  static __device__ __forceinline__ void st(float* p, int64_t i, float v) { p[i] = v; }
  // the value a consumer of the stored element will read back (identity in fp32): a kernel that stores an activation
  // AND keeps using it in registers continues with this, so forward and backward see the same numbers
  static __device__ __forceinline__ float rt(float v) { return v; }
  static __device__ __forceinline__ f4 rt4(f4 v) { return v; }
};
// Tile policies: ST names the storage element type of the activation buffers AND the kind of GVPConvLayer the
// kernel computes (LayerKind<ST> below).  float / bf16s: CASTER-DTA's layers (relu, vector gate); the two below are
// fp32 storage with the reference's other two layer kinds (a11 of the survey: CPD-style and PocketMiner-style stacks).
struct f32_gvpdef {};   // activations (relu, sigmoid), vector_gate=False: the gvp_layers.py defaults
struct f32_linear {};   // activations (None, None), vector_gate=False
template <> struct Io<bf16s> {
  static constexpr bool BF = true;
  static __device__ __forceinline__ float widen(uint32_t hi16) { return __uint_as_float(hi16); }
  static __device__ __forceinline__ uint16_t narrow(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  static __device__ __forceinline__ f4 ld4(const float* p, int64_t i) {        // 4 consecutive bf16 = one 8-byte load
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + i);
    return f4{widen(u.x << 16), widen(u.x & 0xffff0000u), widen(u.y << 16), widen(u.y & 0xffff0000u)};
  }
  static __device__ __forceinline__ float ld(const float* p, int64_t i) {
    return widen((uint32_t)reinterpret_cast<const uint16_t*>(p)[i] << 16);
  }
  static __device__ __forceinline__ void st4(float* p, int64_t i, f4 v) {
    uint2 u;
    u.x = (uint32_t)narrow(v[0]) | ((uint32_t)narrow(v[1]) << 16);
    u.y = (uint32_t)narrow(v[2]) | ((uint32_t)narrow(v[3]) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p) + i) = u;
  }
  static __device__ __forceinline__ void st(float* p, int64_t i, float v) { reinterpret_cast<uint16_t*>(p)[i] = narrow(v); }
  static __device__ __forceinline__ float rt(float v) { return widen((uint32_t)narrow(v) << 16); }
  static __device__ __forceinline__ f4 rt4(f4 v) { return f4{rt(v[0]), rt(v[1]), rt(v[2]), rt(v[3])}; }
};

template <> struct Io<f32_gvpdef> : Io<float> {};
template <> struct Io<f32_linear> : Io<float> {};

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

// STEPS k-steps that map to nothing (padding inside a Segs list).
template <int STEPS>
struct SegNone {
  static constexpr int steps = STEPS;
  static __host__ __device__ int col(int, int) { return -1; }
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

// fp32 fragment image of one GEMM: per output tile mt a region of NSTEPS * 64 floats.  The first NSTEPS / 4 groups of four
// k-steps are stored LANE-MAJOR -- [group][lane][4 steps]: ONE ds_read_b128 at (16 B x lane, conflict-free) fetches the A
// operands of four MFMAs -- and the remaining NSTEPS % 4 steps step-major ([step][lane], one ds_read_b32 each, as every
// step was until round 3: a read and a wait per MFMA pair).  idx = position in the image; (mt, s, lane) = what it holds.
template <int NSTEPS>
__host__ __device__ inline void frag_decode(int idx, int& mt, int& s, int& lane) {
  constexpr int REGION = NSTEPS * 64, FULL = NSTEPS / 4;
  mt = idx / REGION;
  const int u = idx - mt * REGION;
  if (u < FULL * 256) { s = 4 * (u >> 8) + (u & 3); lane = (u & 255) >> 2; }
  else { const int v = u - FULL * 256; s = 4 * FULL + (v >> 6); lane = v & 63; }
}

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
  // what lane `lane` supplies as the A operand of k-step s of output tile mt: offset into W, or -1 (zero)
  static __host__ __device__ int offset_of(int mt, int s, int lane) {
    const int r = row(mt, lane & 15), c = KSegs::col(s, lane >> 4);
    return (r >= 0 && c >= 0) ? r * LD + c : -1;
  }
  static __host__ __device__ float value(const float* W, int mt, int s, int lane) {
    const int o = offset_of(mt, s, lane);
    return o >= 0 ? W[o] : 0.f;
  }
  // element `idx` of this GEMM's fp32 fragment image (layout: frag_decode): offset into W or -1 / value
  static __host__ __device__ int offset(int idx) {
    int mt, s_, lane;
    frag_decode<NSTEPS>(idx, mt, s_, lane);
    return offset_of(mt, s_, lane);
  }
  static __host__ __device__ float element(const float* W, int idx) {
    int mt, s_, lane;
    frag_decode<NSTEPS>(idx, mt, s_, lane);
    return value(W, mt, s_, lane);
  }
};

// The same weight matrix used TRANSPOSED (data gradients): output rows follow a
// forward k-slot map (so gradients land exactly where the forward operand lived:
// tile mt, register r <-> forward slot 4 mt + r), k-slots follow the map of the
// forward OUTPUT rows.  element = W[c_k][c_out].
template <class RowSegs, class KSegs, int LD>
struct GemmT {
  static constexpr int MT = ceil4(RowSegs::steps);
  static constexpr int NSTEPS = KSegs::steps;
  static constexpr int NFRAG = MT * NSTEPS;
  static __host__ __device__ int offset_of(int mt, int s, int lane) {
    const int m = lane & 15, slot = 4 * mt + (m & 3);
    const int c_out = slot < RowSegs::steps ? RowSegs::col(slot, m >> 2) : -1;
    const int c_k = KSegs::col(s, lane >> 4);
    return (c_out >= 0 && c_k >= 0) ? c_k * LD + c_out : -1;
  }
  static __host__ __device__ float value(const float* W, int mt, int s, int lane) {
    const int o = offset_of(mt, s, lane);
    return o >= 0 ? W[o] : 0.f;
  }
  static __host__ __device__ int offset(int idx) {
    int mt, s_, lane;
    frag_decode<NSTEPS>(idx, mt, s_, lane);
    return offset_of(mt, s_, lane);
  }
  static __host__ __device__ float element(const float* W, int idx) {
    int mt, s_, lane;
    frag_decode<NSTEPS>(idx, mt, s_, lane);
    return value(W, mt, s_, lane);
  }
};

__device__ __forceinline__ f4 mfma(float a, float b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- bf16 matrix-core path (activation storage bf16, template flag BF) ------------------------------------------
// v_mfma_f32_16x16x16_bf16: lane (m | n = l & 15, g = l >> 4) supplies k = 4g .. 4g+3 as four bf16, D as above.  Four
// consecutive k-STEPS of the fp32 scheme (step s, group g) become the four k's of one lane, so ONE instruction (4
// passes) replaces four v_mfma_f32_16x16x4_f32 (8 passes each) and the chaining property is untouched: an accumulator
// tile, rounded to bf16 two registers at a time (v_cvt_pk_bf16_f32), is still directly the next B operand.  The
// weight fragments of such a GEMM are PACKED in the image: 8 bytes per lane per group of four steps, in the first
// ceil(NSTEPS / 4) * 128 floats of the GEMM's (mt) region -- which fits the fp32 region for NSTEPS >= 2, so every
// offset of the image layout is unchanged.  One-step GEMMs (K <= 4: wh / wv of the 4-channel GVPs) stay on the fp32
// instruction with fp32 fragments.  Accumulation is fp32 either way.
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
template <class G> constexpr bool packs_bf16() { return G::NSTEPS >= 2; }
__host__ __device__ inline uint32_t bf16_bits(float v) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v); }   // RNE
template <class G>
__host__ __device__ float packed_element(const float* W, int idx) {
  constexpr int NS_ = G::NSTEPS, REGION = NS_ * 64;
  const int mt = idx / REGION, u = idx - mt * REGION, q = u >> 7;
  if (q >= ceil4(NS_)) return 0.f;
  const int lane = (u & 127) >> 1, s0 = 4 * q + 2 * (u & 1);
  const float lo = s0 < NS_ ? G::value(W, mt, s0, lane) : 0.f;
  const float hi = s0 + 1 < NS_ ? G::value(W, mt, s0 + 1, lane) : 0.f;
  return __builtin_bit_cast(float, bf16_bits(lo) | (bf16_bits(hi) << 16));
}
template <class G>
__host__ __device__ float gemm_element(const float* W, int idx, bool packed) {
  return (packed && packs_bf16<G>()) ? packed_element<G>(W, idx) : G::element(W, idx);
}
// two floats -> one register of two bf16 (RNE).  Written as two scalar conversions into a 2-vector: the backend
// pairs them into ONE v_cvt_pk_bf16_f32 (mostly; the rest costs a v_perm) and keeps track of the VALU -> MFMA wait
// states.  Two formulations that did NOT survive: an inline-asm v_cvt_pk_bf16_f32 is invisible to the hazard
// recogniser and fed the matrix core stale registers; __builtin_convertvector on a float2 left operand arrays in
// scratch memory (32 B/lane in every bf16 forward kernel).
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 r = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ bf4 pack_bf4(float a, float b, float c, float d) {
  return __builtin_bit_cast(bf4, uint2{cvt_pk_bf16(a, b), cvt_pk_bf16(c, d)});
}
__device__ __forceinline__ f4 mfma_bf(bf4 a, bf4 b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// acc[j] += sum over this GEMM's k-steps of A-fragment(mt, s) x b[j][s] for TN
// tiles in lockstep: one LDS fragment read feeds TN independent MFMA chains.
// With a single tile, long chains are split over two accumulators instead (a
// dependent f32 MFMA chain issues every 40 cycles, independent ones every 32).
// k-steps 4Q .. 4Q+3 of the packed path; the step indices are template constants, so every element of `b` is named at
// compile time (a run-time-looking index, even in a fully unrolled loop, left some operand arrays in scratch memory)
template <int S, int N>
__device__ __forceinline__ float step_or_zero(const float (&b)[N]) {
  if constexpr (S < N) return b[S]; else return 0.f;
}
template <class G, int TN, int Q>
__device__ __forceinline__ void apply_bf16_steps(const float* frag, const float (&b)[TN][G::NSTEPS], f4 (&acc)[TN]) {
  if constexpr (Q < ceil4(G::NSTEPS)) {
    // (read as a vector of FLOATS: the staging code stores floats, and a load through an integer type may be moved
    // across those stores by type-based alias analysis)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const bf4 a = __builtin_bit_cast(bf4, reinterpret_cast<const f2*>(frag)[Q * 64]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      acc[j] = mfma_bf(a, pack_bf4(step_or_zero<4 * Q>(b[j]), step_or_zero<4 * Q + 1>(b[j]), step_or_zero<4 * Q + 2>(b[j]),
                                   step_or_zero<4 * Q + 3>(b[j])), acc[j]);
    apply_bf16_steps<G, TN, Q + 1>(frag, b, acc);
  }
}

template <class G, int TN, bool BF = false>
__device__ __forceinline__ void apply(const float* frag, int mt, const float (&b)[TN][G::NSTEPS], f4 (&acc)[TN], int lane) {
  if constexpr (BF && packs_bf16<G>()) {
    apply_bf16_steps<G, TN, 0>(frag + mt * G::NSTEPS * 64 + 2 * lane, b, acc);
    return;
  }
  // fp32 image of this output tile (frag_decode): FULL lane-major groups of four steps, then REM step-major steps.  All the
  // fragment reads of the call are issued up front (NSTEPS registers, <= 19): the MFMAs then run back to back instead of
  // one LDS round trip per pair.
  constexpr int NS_ = G::NSTEPS, FULL = NS_ / 4, REM = NS_ % 4;
  const float* f = frag + mt * NS_ * 64;
  float a[NS_];
#pragma unroll
  for (int q = 0; q < FULL; ++q) {
    const f4 a4 = *reinterpret_cast<const f4*>(f + q * 256 + 4 * lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) a[4 * q + k] = a4[k];
  }
#pragma unroll
  for (int r = 0; r < REM; ++r) a[4 * FULL + r] = f[FULL * 256 + r * 64 + lane];
  if (TN == 1 && NS_ >= 8) {
    f4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s + 1 < NS_; s += 2) {
      acc[0] = mfma(a[s], b[0][s], acc[0]);
      acc2 = mfma(a[s + 1], b[0][s + 1], acc2);
    }
    if (NS_ & 1) acc[0] = mfma(a[NS_ - 1], b[0][NS_ - 1], acc[0]);
    acc[0] += acc2;
    return;
  }
#pragma unroll
  for (int s = 0; s < NS_; ++s) {
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = mfma(a[s], b[j][s], acc[j]);
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

// Sum over the 16 items of a tile (the 16 lanes of a group row); the total is
// valid in lane i == 15 of every group.
__device__ __forceinline__ float row_total(float x) {
  x += row_shr_f<1>(x);
  x += row_shr_f<2>(x);
  x += row_shr_f<4>(x);
  x += row_shr_f<8>(x);
  return x;
}

// ---- weight gradients on the matrix cores, without leaving registers --------
// dW[row][col] = sum_items dY[row][item] * X[col][item] reduces over ITEMS, but
// everything above keeps the item on lane & 15 ("item-on-lane").  An MFMA against
// a 0/1 selection fragment transposes a group of 4 k-steps (16 slots) exactly:
//     T[item][n] = sum_g V[item][slot (4T + r, g)] * [n == 4r + g]
// whose D tile holds, in lane (n, g') register rr, the value of slot
// (s = 4T + n/4, g = n%4) for item 4g' + rr: "slot-on-lane".  Two such tiles are
// directly the A and B operands of the item-reduction GEMM (k-step rr <-> items
// 4g' + rr).  Rows / columns of the result are in slot order and are mapped back
// to nn.Linear rows / columns by the segment maps when the tile is flushed.
__device__ __forceinline__ float sel_val(int lane, int r) { return ((lane & 15) == 4 * r + (lane >> 4)) ? 1.f : 0.f; }

// With a per-wave LDS scratch of TSCR_FLOATS floats (`tscr`), the same transposition is four
// ds_write_b32 + one ds_read_b128 per 16 slots instead of four MFMAs: it takes the operand
// transposes (~30 % of the conv backward's MFMAs) off the matrix-core pipe that two waves share.
constexpr int TSCR_LD = 20, TSCR_FLOATS = 16 * TSCR_LD;     // [slot n][item], rows padded to 80 B
#ifndef CGVP_TSCR_MIN_STEPS
#define CGVP_TSCR_MIN_STEPS 0
#endif
constexpr int TSCR_MIN_STEPS = CGVP_TSCR_MIN_STEPS;         // operands of at most this many slots stay on the MFMA path
template <int NSTEPS>
__device__ __forceinline__ void transpose_slots(const float (&v)[NSTEPS], f4 (&out)[ceil4(NSTEPS)], int lane,
                                                float* tscr = nullptr) {
  if (tscr && NSTEPS > TSCR_MIN_STEPS) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int T = 0; T < ceil4(NSTEPS); ++T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) tscr[(4 * r + g) * TSCR_LD + i] = (4 * T + r < NSTEPS) ? v[4 * T + r] : 0.f;
      out[T] = *reinterpret_cast<const f4*>(tscr + i * TSCR_LD + 4 * g);     // lane (n = i, g') <- slot n, items 4g'..4g'+3
    }
    return;
  }
  const float sel[4] = {sel_val(lane, 0), sel_val(lane, 1), sel_val(lane, 2), sel_val(lane, 3)};
#pragma unroll
  for (int T = 0; T < ceil4(NSTEPS); ++T) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * T + r < NSTEPS) acc = mfma(v[4 * T + r], sel[r], acc);
    out[T] = acc;
  }
}

// acc[mt][nt] = sum over the 16 items of the tile of A_T[mt] (x) B_T[nt]
template <int MT, int NT_, bool BF = false>
__device__ __forceinline__ void outer_items(const f4 (&A)[MT], const f4 (&B)[NT_], f4 (&acc)[MT][NT_]) {
  if constexpr (BF) {            // the 16 items of the tile are the 16 k's of ONE bf16 instruction per block
    bf4 a[MT], b[NT_];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = pack_bf4(A[mt][0], A[mt][1], A[mt][2], A[mt][3]);
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) b[nt] = pack_bf4(B[nt][0], B[nt][1], B[nt][2], B[nt][3]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT_; ++nt) acc[mt][nt] = mfma_bf(a[mt], b[nt], acc[mt][nt]);
    return;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT_; ++nt) acc[mt][nt] = mfma(A[mt][rr], B[nt][rr], acc[mt][nt]);
}

// Weight-gradient partials are accumulated WITHOUT atomics, in a block that exactly one
// wave owns: AccPriv = a wave-private block in LDS, zeroed at kernel start, read-add-write
// per tile; the workgroup sums its waves' blocks once at the end and writes ONE slab row.
// What this replaces, both measured on MI355X:
//   * a workgroup-SHARED LDS block with ds_add_f32: LDS float atomics retire ~1 lane per
//     3 cycles per CU (tools/ldsatomic_probe.hip: 193 cycles for one 64-lane instruction,
//     linear in active lanes and in waves) -- 63 % of the conv backward;
//   * a wave-private slab row in GLOBAL memory (plain stores): 11-30 us of store drain per
//     node-backward launch and a 4x larger reduction.
// A private ds_read / add / ds_write costs ~10 cycles per wave-instruction.
struct AccPriv {
  static constexpr bool BATCHED = true;
  // the first blockDim.x floats of the kernel's dynamic LDS: one trash word per thread
  static __device__ __forceinline__ float* trash() { extern __shared__ float cgvp_dyn_lds[]; return cgvp_dyn_lds + threadIdx.x; }
  static __device__ __forceinline__ float load(const float* p, bool) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void add(float* p, float v, bool) { *p += v; }
};

// The same block when every element is written by at most ONE flush (accumulators kept in registers and flushed once per
// wave into a zeroed block): the old value is zero by construction, so nothing is read.
struct AccStoreOnce {
  static constexpr bool BATCHED = true;
  static __device__ __forceinline__ float* trash() { return AccPriv::trash(); }
  static __device__ __forceinline__ float load(const float*, bool) { return 0.f; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void add(float* p, float v, bool) { *p = v; }
};

// v[k] is added to p[k] by the lanes with on[k], everybody else adds into the thread's trash word:
// N unconditional read-add-writes with the reads in flight together (bias / LayerNorm gradients,
// which only lane i == 15 of a group row holds after row_total).
template <class Acc, int N>
__device__ __forceinline__ void add_where(float* const (&p)[N], const bool (&on)[N], const float (&v)[N]) {
  float* q[N];
  float old[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { q[k] = on[k] ? p[k] : Acc::trash(); old[k] = Acc::load(q[k], false); }
#pragma unroll
  for (int k = 0; k < N; ++k) Acc::store(q[k], old[k] + v[k]);
}

// Add a slot-ordered weight-gradient tile grid into the W-layout block `dst` ([.][LD]).
// BATCHED policies read the old values of a column tile before writing its sums (the
// targets of one call are distinct elements), so the reads are in flight together
// instead of one read-add-write round trip per element.
template <class Acc, class RowSegs, class ColSegs, int MT, int NT_>
__device__ __forceinline__ void flush_slots(float* dst, bool first, int LD, const f4 (&acc)[MT][NT_], int lane) {
  const int n = lane & 15, gq_ = lane >> 4;
#ifdef CGVP_EXPERIMENT_NOFLUSH       // timing experiment only (wrong results): keep the producers alive, drop the LDS read-add-writes
  {
    float t = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) t += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
    *Acc::trash() += t;
    return;
  }
#endif
#pragma unroll
  for (int nt = 0; nt < NT_; ++nt) {
    const int cs = 4 * nt + (n >> 2);
    const int col = cs < ColSegs::steps ? ColSegs::col(cs, n & 3) : -1;
    if constexpr (Acc::BATCHED) {
      // One accumulator register quad at a time: 4 reads, 4 adds, 4 writes, all unconditional --
      // padding lanes are pointed at the thread's trash word.  (A load under a lane mask is
      // waited for at the join, one LDS round trip per element; letting the scheduler hoist all
      // reads of a call instead spills.)
      const int trash = (int)(Acc::trash() - dst);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int idx[4];
        float old[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * gq_ + r, rs = 4 * mt + (m >> 2);
          const int row = rs < RowSegs::steps ? RowSegs::col(rs, m & 3) : -1;
          idx[r] = (row >= 0 && col >= 0) ? row * LD + col : trash;
          old[r] = Acc::load(dst + idx[r], first);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Acc::store(dst + idx[r], old[r] + acc[mt][nt][r]);
#ifndef CGVP_FLUSH_NOBARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * gq_ + r, rs = 4 * mt + (m >> 2);
          const int row = rs < RowSegs::steps ? RowSegs::col(rs, m & 3) : -1;
          if (row >= 0 && col >= 0) Acc::add(dst + row * LD + col, acc[mt][nt][r], first);
        }
    }
  }
}

// ------------------------------------------------------------------ one GVP
// Image slice of a GVP: [Wh frags | Ws frags | Wv frags | Wsv frags] then a
// vector region [bs (SO) | type table (NT x SO, transposed one-hot columns) |
// bsv (16, P2 order is resolved at read time)].
//   SSegs: k-slots of the scalar inputs, columns relative to the ws row (after
//          the NT type columns are skipped by BASE offsets chosen by the caller)
//   VSegs: k-slots of the vector-channel inputs (columns of wh)
//   VM   : what scales the vector outputs (gvp_layers.py:155-166): VM_GATE  v * sigmoid(wsv(s))       vector_gate=True
//                                                                 VM_NORM  v * sigmoid(|v|)          vector_act=sigmoid
//                                                                 VM_NONE  v                         vector_act=None
//          (the wsv slots of the image / gradient block stay in place for every VM: zero, and zero gradient)
constexpr int VM_GATE = 0, VM_NORM = 1, VM_NONE = 2;
template <int NT, int SI, int VI, int SO, int VO, int H, bool RELU_, class SSegs, class VSegs, int VM = VM_GATE>
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
  static __host__ __device__ float element(const float* P, int idx, bool packed = false) {
    if (idx < F_WS * 64) return gemm_element<GWh>(P, idx, packed);
    if (idx < F_WV * 64) return gemm_element<GWs>(P + A::ws(NT), idx - F_WS * 64, packed);
    if (VO > 0 && idx < F_WSV * 64) return gemm_element<GWv>(P + A::wv(NT), idx - F_WV * 64, packed);
    if (VO > 0 && idx < V_BS) return gemm_element<GWsv>(P + A::wsv(NT), idx - F_WSV * 64, packed);
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
  template <int TN, bool BF = false>
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
      apply<GWh, TN, BF>(img + F_WH * 64, 0, b, acc, lane);
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
      apply<GWs, TN, BF>(img + F_WS * 64, t, bfull, acc, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) c[j].sp[t] = acc[j];
    }
    if (VO > 0) {
      if (VM == VM_GATE) {
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
        apply<GWsv, TN, BF>(img + F_WSV * 64, 0, bsp, gate, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) c[j].sg[r] = gvp::f_sigmoid(gate[j][r]);
      }
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
        apply<GWv, TN, BF>(img + F_WV * 64, 0, bh, acc, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) c[j].vp[p] = acc[j];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < VOR; ++r) {
          if (VM == VM_NORM) {          // sigmoid of the clamped norm of the un-scaled output (gvp_layers.py:164-166)
            const float n2 = c[j].vp[0][r] * c[j].vp[0][r] + c[j].vp[1][r] * c[j].vp[1][r] + c[j].vp[2][r] * c[j].vp[2][r];
            c[j].sg[r] = gvp::f_sigmoid(gvp::f_sqrt(gvp::f_max(n2, gvp::kNormEps)));
          } else if (VM == VM_NONE) {
            c[j].sg[r] = 1.0f;
          }
#pragma unroll
          for (int p = 0; p < 3; ++p) vo[j][p][r] = c[j].vp[p][r] * c[j].sg[r];
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) so[j][t][r] = RELU ? gvp::f_max(c[j].sp[t][r], 0.f) : c[j].sp[t][r];
  }

  // ---------------------------------------------------------------- backward
  // Transposed image slice: [WsT | WsvT | WvT | WhT] fragments.
  typedef GemmT<Segs<SSegs, Seg<P2, NT + SI, H>>, Segs<Seg<P1, 0, SO>>, K> TWs;      // d(inputs) = Ws^T dsp
  typedef GemmT<Segs<Seg<P1, 0, SO>>, Segs<Seg<P2, 0, (VO > 0 ? VO : 1)>>, SO> TWsv;  // dsp += Wsv^T dgate
  typedef GemmT<Segs<Seg<P2, 0, H>>, Segs<Seg<P2, 0, (VO > 0 ? VO : 1)>>, H> TWv;     // dvh = Wv^T dvp
  typedef GemmT<VSegs, Segs<Seg<P2, 0, H>>, VI> TWh;                                  // dV = Wh^T dvh
  static_assert(TWsv::MT == OT && TWv::MT == 1 && TWh::MT == 1, "tile bookkeeping");
  static constexpr int FT_WS = 0;
  static constexpr int FT_WSV = FT_WS + TWs::NFRAG;
  static constexpr int FT_WV = FT_WSV + (VO > 0 ? TWsv::NFRAG : 0);
  static constexpr int FT_WH = FT_WV + (VO > 0 ? TWv::NFRAG : 0);
  static constexpr int SIZE_T = (FT_WH + TWh::NFRAG) * 64;
  static __host__ __device__ float element_t(const float* P, int idx, bool packed = false) {
    if (idx < FT_WSV * 64) return gemm_element<TWs>(P + A::ws(NT), idx, packed);
    if (VO > 0 && idx < FT_WV * 64) return gemm_element<TWsv>(P + A::wsv(NT), idx - FT_WSV * 64, packed);
    if (VO > 0 && idx < FT_WH * 64) return gemm_element<TWv>(P + A::wv(NT), idx - FT_WV * 64, packed);
    return gemm_element<TWh>(P, idx - FT_WH * 64, packed);
  }

  struct Grads {          // per-lane gradients the weight-gradient GEMMs consume
    f4 dsp[OT];           // d(pre-activation scalars), P1
    f4 dgate;             // d(gate), P2 rows o
    f4 dvp[3];            // d(wv.vh), P2 rows o
    f4 dvh[3];            // d(wh.V), P2 rows h
  };

  // Backward of `forward` for one tile.  d_so / d_vo: gradients of the outputs;
  // d_bs / d_bv: gradients of the k-slot inputs (same slots as bs / bv).
  template <bool BF = false>
  static __device__ __forceinline__ void backward(const float* imgT, int lane, const Cache& c,
                                                  const f4 (&d_so)[OT], const float (&d_vo)[3][VOR],
                                                  float (&d_bs)[SSTEPS], float (&d_bv)[3][VSTEPS], Grads& gr) {
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) gr.dsp[t][r] = (RELU && c.sp[t][r] <= 0.f) ? 0.f : d_so[t][r];
    gr.dgate = zero;
#pragma unroll
    for (int p = 0; p < 3; ++p) gr.dvp[p] = gr.dvh[p] = zero;
    if (VO > 0) {
#pragma unroll
      for (int r = 0; r < VOR; ++r) {
        const float sg = c.sg[r];
        float dsg = 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) { dsg = fmaf(d_vo[p][r], c.vp[p][r], dsg); gr.dvp[p][r] = d_vo[p][r] * sg; }
        if (VM == VM_GATE) gr.dgate[r] = dsg * sg * (1.0f - sg);
        if (VM == VM_NORM) {            // vo = vp sigmoid(n), n = sqrt(max(|vp|^2, eps)): no gradient through n below the clamp
          const float n2 = c.vp[0][r] * c.vp[0][r] + c.vp[1][r] * c.vp[1][r] + c.vp[2][r] * c.vp[2][r];
          const float k = n2 > gvp::kNormEps ? dsg * sg * (1.0f - sg) * gvp::f_rsqrt(n2) : 0.f;
#pragma unroll
          for (int p = 0; p < 3; ++p) gr.dvp[p][r] = fmaf(k, c.vp[p][r], gr.dvp[p][r]);
        }
      }
      if (VM == VM_GATE) {
        float bg[1][TWsv::NSTEPS];
#pragma unroll
        for (int r = 0; r < TWsv::NSTEPS; ++r) bg[0][r] = gr.dgate[r];
#pragma unroll
        for (int t = 0; t < OT; ++t) {
          f4 acc[1] = {gr.dsp[t]};
          apply<TWsv, 1, BF>(imgT + FT_WSV * 64, t, bg, acc, lane);
          gr.dsp[t] = acc[0];
        }
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float b[1][TWv::NSTEPS];
#pragma unroll
        for (int r = 0; r < TWv::NSTEPS; ++r) b[0][r] = gr.dvp[p][r];
        f4 acc[1] = {zero};
        apply<TWv, 1, BF>(imgT + FT_WV * 64, 0, b, acc, lane);
        gr.dvh[p] = acc[0];
      }
    }
    float d_vn[HR];
    {
      float bd[1][4 * OT];
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) bd[0][4 * t + r] = gr.dsp[t][r];
#pragma unroll
      for (int mt = 0; mt < TWs::MT; ++mt) {
        f4 acc[1] = {zero};
        apply<TWs, 1, BF>(imgT + FT_WS * 64, mt, bd, acc, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int slot = 4 * mt + r;
          if (slot < SSTEPS) d_bs[slot] = acc[0][r];
          else if (slot < SSTEPS + HR) d_vn[slot - SSTEPS] = acc[0][r];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < HR; ++r) {      // through vn = sqrt(max(|vh|^2, eps)): no gradient below the clamp
      const float n2 = c.vh[0][r] * c.vh[0][r] + c.vh[1][r] * c.vh[1][r] + c.vh[2][r] * c.vh[2][r];
      const float k = n2 > gvp::kNormEps ? d_vn[r] * gvp::f_rcp(c.vn[r]) : 0.f;
#pragma unroll
      for (int p = 0; p < 3; ++p) gr.dvh[p][r] = fmaf(k, c.vh[p][r], gr.dvh[p][r]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      float b[1][TWh::NSTEPS];
#pragma unroll
      for (int r = 0; r < TWh::NSTEPS; ++r) b[0][r] = gr.dvh[p][r];
      f4 acc[1] = {zero};
      apply<TWh, 1, BF>(imgT + FT_WH * 64, 0, b, acc, lane);
#pragma unroll
      for (int s2 = 0; s2 < VSTEPS; ++s2) d_bv[p][s2] = acc[0][s2];
    }
  }

  // Weight gradients of this GVP for one tile, written (first) / added into the
  // arena-layout gradient block `gblk` of the wave's private slab row.  Registers
  // and MFMAs only.
  static constexpr int WG_SCRATCH = 0;
  static constexpr int NTS = ceil4(NT);                 // k-steps of the one-hot type columns
  typedef Segs<Seg<P2, 0, (NT > 0 ? NT : 1)>, SSegs, Seg<P2, NT + SI, H>> WsCols;   // [types | scalars | norms]
  static constexpr bool PACK_V = VO > 0 && VOR + HR <= 4 && HR + VSTEPS <= 4;   // dWv and dWh fit one block
  template <class Acc, bool BF = false>
  static __device__ __forceinline__ void weight_grads(float* gblk, bool first, int lane, int type, bool active,
                                                      const float (&bs)[SSTEPS], const float (&bv)[3][VSTEPS],
                                                      const Cache& c, const Grads& gr, float* tscr = nullptr) {
    const int i = lane & 15, g = lane >> 4;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    // ---- dWs = dsp (x) [onehot | s | vn],  dbs = sum dsp
    {
      float a[4 * OT];
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[4 * t + r] = active ? gr.dsp[t][r] : 0.f;
      f4 AT[OT];
      transpose_slots<4 * OT>(a, AT, lane, tscr);
      constexpr int NB = (NT > 0 ? NTS : 1) + SSTEPS + HR;
      float b[NB];
#pragma unroll
      for (int s = 0; s < (NT > 0 ? NTS : 1); ++s) b[s] = (NT > 0 && 4 * s + g == type) ? 1.f : 0.f;
#pragma unroll
      for (int s = 0; s < SSTEPS; ++s) b[(NT > 0 ? NTS : 1) + s] = bs[s];
#pragma unroll
      for (int r = 0; r < HR; ++r) b[(NT > 0 ? NTS : 1) + SSTEPS + r] = c.vn[r];
      f4 BT[ceil4(NB)];
      transpose_slots<NB>(b, BT, lane, tscr);
      f4 acc[OT][ceil4(NB)];
#pragma unroll
      for (int x = 0; x < OT; ++x)
#pragma unroll
        for (int y = 0; y < ceil4(NB); ++y) acc[x][y] = zero;
      outer_items<OT, ceil4(NB), BF>(AT, BT, acc);
      flush_slots<Acc, Segs<Seg<P1, 0, SO>>, WsCols, OT, ceil4(NB)>(gblk + A::ws(NT), first, K, acc, lane);
    }
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      float tot[4];
      float* q[4];
      bool on[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tot[r] = row_total(active ? gr.dsp[t][r] : 0.f);
        q[r] = gblk + A::bs(NT) + 16 * t + 4 * g + r;
        on[r] = i == 15;
      }
      add_where<Acc, 4>(q, on, tot);
    }
    if (VO > 0) {
      // ---- dWsv = dgate (x) sp, dbsv = sum dgate
      float a[VOR], b[4 * OT];
#pragma unroll
      for (int r = 0; r < VOR; ++r) a[r] = active ? gr.dgate[r] : 0.f;
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) b[4 * t + r] = c.sp[t][r];
      f4 AT[1], BT[OT], acc[1][OT];
      transpose_slots<VOR>(a, AT, lane, tscr);
      transpose_slots<4 * OT>(b, BT, lane, tscr);
#pragma unroll
      for (int y = 0; y < OT; ++y) acc[0][y] = zero;
      outer_items<1, OT, BF>(AT, BT, acc);
      flush_slots<Acc, Segs<Seg<P2, 0, (VO > 0 ? VO : 1)>>, Segs<Seg<P1, 0, SO>>, 1, OT>(gblk + A::wsv(NT), first, SO, acc, lane);
      {
        float tot[VOR];
        float* q[VOR];
        bool on[VOR];
#pragma unroll
        for (int r = 0; r < VOR; ++r) {
          tot[r] = row_total(active ? gr.dgate[r] : 0.f);
          q[r] = gblk + A::bsv(NT) + 4 * r + g;
          on[r] = i == 15 && 4 * r + g < VO;
        }
        add_where<Acc, VOR>(q, on, tot);
      }
      // ---- dWv = sum_planes dvp (x) vh   (alone only when it cannot share a block with dWh, below)
      if constexpr (!PACK_V) {
        f4 accv[1][1] = {{zero}};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          float av[VOR], bh[HR];
#pragma unroll
          for (int r = 0; r < VOR; ++r) av[r] = active ? gr.dvp[p][r] : 0.f;
#pragma unroll
          for (int r = 0; r < HR; ++r) bh[r] = c.vh[p][r];
          f4 AV[1], BH[1];
          transpose_slots<VOR>(av, AV, lane, tscr);
          transpose_slots<HR>(bh, BH, lane, tscr);
          outer_items<1, 1, BF>(AV, BH, accv);
        }
        flush_slots<Acc, Segs<Seg<P2, 0, (VO > 0 ? VO : 1)>>, Segs<Seg<P2, 0, H>>, 1, 1>(gblk + A::wv(NT), first, H, accv, lane);
      }
    }
    if constexpr (PACK_V) {
      // ---- dWv and dWh share one 16x16 block: rows [dvp slots | dvh slots], columns [vh slots | V_in
      // slots]; its two diagonal sub-blocks are the two gradients (the off-diagonal ones are dropped).
      // Half the outer-product MFMAs and half the operand transposes of the vector part.
      f4 acc2[1][1] = {{zero}};
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float a2[VOR + HR], b2[HR + VSTEPS];
#pragma unroll
        for (int r = 0; r < VOR; ++r) a2[r] = active ? gr.dvp[p][r] : 0.f;
#pragma unroll
        for (int r = 0; r < HR; ++r) a2[VOR + r] = active ? gr.dvh[p][r] : 0.f;
#pragma unroll
        for (int r = 0; r < HR; ++r) b2[r] = c.vh[p][r];
#pragma unroll
        for (int s_ = 0; s_ < VSTEPS; ++s_) b2[HR + s_] = bv[p][s_];
        f4 A2[1], B2[1];
        transpose_slots<VOR + HR>(a2, A2, lane, tscr);
        transpose_slots<HR + VSTEPS>(b2, B2, lane, tscr);
        outer_items<1, 1, BF>(A2, B2, acc2);
      }
      flush_slots<Acc, Segs<Seg<P2, 0, (VO > 0 ? VO : 1)>, SegNone<HR>>, Segs<Seg<P2, 0, H>, SegNone<VSTEPS>>, 1, 1>(
          gblk + A::wv(NT), first, H, acc2, lane);
      flush_slots<Acc, Segs<SegNone<VOR>, Seg<P2, 0, H>>, Segs<SegNone<HR>, VSegs>, 1, 1>(gblk, first, VI, acc2, lane);
    } else {
      // ---- dWh = sum_planes dvh (x) V_in
      f4 acch[1][1] = {{zero}};
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float ah[HR], bin[VSTEPS];
#pragma unroll
        for (int r = 0; r < HR; ++r) ah[r] = active ? gr.dvh[p][r] : 0.f;
#pragma unroll
        for (int s = 0; s < VSTEPS; ++s) bin[s] = bv[p][s];
        f4 AH[1], BI[1];
        static_assert(VSTEPS <= 4 && HR <= 4, "vector operands fit one slot tile");
        transpose_slots<HR>(ah, AH, lane, tscr);
        transpose_slots<VSTEPS>(bin, BI, lane, tscr);
        outer_items<1, 1, BF>(AH, BI, acch);
      }
      flush_slots<Acc, Segs<Seg<P2, 0, H>>, VSegs, 1, 1>(gblk, first, VI, acch, lane);
    }
  }

  // The outputs `forward` returned, re-derived from its cache (so = act(sp), vo = vp * sg): lets a caller that is short of
  // registers drop a GVP's outputs after the next GVP consumed them and rebuild them where its weight gradients need them as
  // the next GVP's inputs.  The asm statements hide the cache values' identity from common-subexpression elimination, which
  // would otherwise keep the first copy alive instead.
  static __device__ __forceinline__ void outputs_from_cache(const Cache& c, float (&so)[4 * OT], float (&vo)[3][VOR]) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = c.sp[t][r];
        asm volatile("" : "+v"(x));
        so[4 * t + r] = RELU ? gvp::f_max(x, 0.f) : x;
      }
#pragma unroll
    for (int r = 0; r < VOR; ++r) {
      float g_ = c.sg[r];
      asm volatile("" : "+v"(g_));
#pragma unroll
      for (int p = 0; p < 3; ++p) vo[p][r] = c.vp[p][r] * g_;
    }
  }

  // ================================================================ lockstep tiles (round 4)
  // The same backward for TN tiles IN LOCKSTEP: every transposed weight fragment is read from LDS once and feeds TN
  // independent MFMA chains (as `forward<TN>` does), so a wave that runs alone on its SIMD has independent work to issue
  // while one tile's chain waits for its accumulator.  Arithmetic per tile identical to `backward`.
  template <int TN, bool BF = false>
  static __device__ __forceinline__ void backward_tn(const float* imgT, int lane, const Cache (&c)[TN],
                                                     const f4 (&d_so)[TN][OT], const float (&d_vo)[TN][3][VOR],
                                                     float (&d_bs)[TN][SSTEPS], float (&d_bv)[TN][3][VSTEPS], Grads (&gr)[TN]) {
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) gr[j].dsp[t][r] = (RELU && c[j].sp[t][r] <= 0.f) ? 0.f : d_so[j][t][r];
      gr[j].dgate = zero;
#pragma unroll
      for (int p = 0; p < 3; ++p) gr[j].dvp[p] = gr[j].dvh[p] = zero;
    }
    if (VO > 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < VOR; ++r) {
          const float sg = c[j].sg[r];
          float dsg = 0.f;
#pragma unroll
          for (int p = 0; p < 3; ++p) { dsg = fmaf(d_vo[j][p][r], c[j].vp[p][r], dsg); gr[j].dvp[p][r] = d_vo[j][p][r] * sg; }
          if (VM == VM_GATE) gr[j].dgate[r] = dsg * sg * (1.0f - sg);
          if (VM == VM_NORM) {
            const float n2 = c[j].vp[0][r] * c[j].vp[0][r] + c[j].vp[1][r] * c[j].vp[1][r] + c[j].vp[2][r] * c[j].vp[2][r];
            const float k = n2 > gvp::kNormEps ? dsg * sg * (1.0f - sg) * gvp::f_rsqrt(n2) : 0.f;
#pragma unroll
            for (int p = 0; p < 3; ++p) gr[j].dvp[p][r] = fmaf(k, c[j].vp[p][r], gr[j].dvp[p][r]);
          }
        }
      if (VM == VM_GATE) {
        float bg[TN][TWsv::NSTEPS];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < TWsv::NSTEPS; ++r) bg[j][r] = gr[j].dgate[r];
#pragma unroll
        for (int t = 0; t < OT; ++t) {
          f4 acc[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[j] = gr[j].dsp[t];
          apply<TWsv, TN, BF>(imgT + FT_WSV * 64, t, bg, acc, lane);
#pragma unroll
          for (int j = 0; j < TN; ++j) gr[j].dsp[t] = acc[j];
        }
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float b[TN][TWv::NSTEPS];
        f4 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[j] = zero;
#pragma unroll
          for (int r = 0; r < TWv::NSTEPS; ++r) b[j][r] = gr[j].dvp[p][r];
        }
        apply<TWv, TN, BF>(imgT + FT_WV * 64, 0, b, acc, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) gr[j].dvh[p] = acc[j];
      }
    }
    float d_vn[TN][HR];
    {
      float bd[TN][4 * OT];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) bd[j][4 * t + r] = gr[j].dsp[t][r];
#pragma unroll
      for (int mt = 0; mt < TWs::MT; ++mt) {
        f4 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = zero;
        apply<TWs, TN, BF>(imgT + FT_WS * 64, mt, bd, acc, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int slot = 4 * mt + r;
            if (slot < SSTEPS) d_bs[j][slot] = acc[j][r];
            else if (slot < SSTEPS + HR) d_vn[j][slot - SSTEPS] = acc[j][r];
          }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < HR; ++r) {
        const float n2 = c[j].vh[0][r] * c[j].vh[0][r] + c[j].vh[1][r] * c[j].vh[1][r] + c[j].vh[2][r] * c[j].vh[2][r];
        const float k = n2 > gvp::kNormEps ? d_vn[j][r] * gvp::f_rcp(c[j].vn[r]) : 0.f;
#pragma unroll
        for (int p = 0; p < 3; ++p) gr[j].dvh[p][r] = fmaf(k, c[j].vh[p][r], gr[j].dvh[p][r]);
      }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      float b[TN][TWh::NSTEPS];
      f4 acc[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[j] = zero;
#pragma unroll
        for (int r = 0; r < TWh::NSTEPS; ++r) b[j][r] = gr[j].dvh[p][r];
      }
      apply<TWh, TN, BF>(imgT + FT_WH * 64, 0, b, acc, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int s2 = 0; s2 < VSTEPS; ++s2) d_bv[j][p][s2] = acc[j][s2];
    }
  }

  // Weight gradients kept IN REGISTERS across all the tiles a wave processes (the register budget of a wave that owns its
  // SIMD: 512 per lane): the slot-ordered outer-product blocks of `weight_grads` as persistent MFMA accumulators, the bias
  // gradients as per-lane partial sums.  `accumulate` adds TN tiles (operand transposes through the per-wave LDS scratch,
  // one region per tile so the tiles' write -> read round trips overlap); `flush` maps the blocks into the W-layout
  // gradient block ONCE per wave -- no per-tile LDS read-add-write, no per-tile row reductions.
  static constexpr int NBW = (NT > 0 ? NTS : 1) + SSTEPS + HR, NBT = ceil4(NBW);
  struct WAcc {
    f4 ws[OT][NBT];
    f4 wsv[1][OT];
    f4 vv[1][1];          // PACK_V: the shared dWv / dWh block; else dWv
    f4 vh_[1][1];         // !PACK_V: dWh
    f4 bs[OT];            // per-lane sums of dsp  (row-reduced at flush)
    float bsv[VOR];       // per-lane sums of dgate
  };
  static __device__ __forceinline__ void wacc_zero(WAcc& w) {
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < OT; ++t) {
#pragma unroll
      for (int y = 0; y < NBT; ++y) w.ws[t][y] = zero;
      w.wsv[0][t] = zero;
      w.bs[t] = zero;
    }
    w.vv[0][0] = zero;
    w.vh_[0][0] = zero;
#pragma unroll
    for (int r = 0; r < VOR; ++r) w.bsv[r] = 0.f;
  }
  // Lanes that hold no item must carry zero gradients in `gr` (the callers' d_so / d_vo are zero there).
  template <int TN, bool BF = false>
  static __device__ __forceinline__ void wacc_accumulate(WAcc& w, int lane, const int (&type)[TN],
                                                         const float (&bs)[TN][SSTEPS], const float (&bv)[TN][3][VSTEPS],
                                                         const Cache (&c)[TN], const Grads (&gr)[TN], float* tscr) {
    const int g = lane >> 4;
    // ---- dWs = dsp (x) [onehot | s | vn],  dbs += dsp
    {
      f4 AT[TN][OT], BT[TN][NBT];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float a[4 * OT];
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) a[4 * t + r] = gr[j].dsp[t][r];
        transpose_slots<4 * OT>(a, AT[j], lane, tscr + j * TSCR_FLOATS);
        float b[NBW];
#pragma unroll
        for (int s = 0; s < (NT > 0 ? NTS : 1); ++s) b[s] = (NT > 0 && 4 * s + g == type[j]) ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < SSTEPS; ++s) b[(NT > 0 ? NTS : 1) + s] = bs[j][s];
#pragma unroll
        for (int r = 0; r < HR; ++r) b[(NT > 0 ? NTS : 1) + SSTEPS + r] = c[j].vn[r];
        transpose_slots<NBW>(b, BT[j], lane, tscr + j * TSCR_FLOATS);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) outer_items<OT, NBT, BF>(AT[j], BT[j], w.ws);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int t = 0; t < OT; ++t) w.bs[t] += gr[j].dsp[t];
    }
    if (VO > 0) {
      // ---- dWsv = dgate (x) sp, dbsv += dgate
      f4 AT[TN][1], BT[TN][OT];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float a[VOR], b[4 * OT];
#pragma unroll
        for (int r = 0; r < VOR; ++r) a[r] = gr[j].dgate[r];
#pragma unroll
        for (int t = 0; t < OT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) b[4 * t + r] = c[j].sp[t][r];
        transpose_slots<VOR>(a, AT[j], lane, tscr + j * TSCR_FLOATS);
        transpose_slots<4 * OT>(b, BT[j], lane, tscr + j * TSCR_FLOATS);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) outer_items<1, OT, BF>(AT[j], BT[j], w.wsv);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < VOR; ++r) w.bsv[r] += gr[j].dgate[r];
      if constexpr (!PACK_V) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          f4 AV[TN][1], BH[TN][1];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float av[VOR], bh[HR];
#pragma unroll
            for (int r = 0; r < VOR; ++r) av[r] = gr[j].dvp[p][r];
#pragma unroll
            for (int r = 0; r < HR; ++r) bh[r] = c[j].vh[p][r];
            transpose_slots<VOR>(av, AV[j], lane, tscr + j * TSCR_FLOATS);
            transpose_slots<HR>(bh, BH[j], lane, tscr + j * TSCR_FLOATS);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) outer_items<1, 1, BF>(AV[j], BH[j], w.vv);
        }
      }
    }
    if constexpr (PACK_V) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        f4 A2[TN][1], B2[TN][1];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float a2[VOR + HR], b2[HR + VSTEPS];
#pragma unroll
          for (int r = 0; r < VOR; ++r) a2[r] = gr[j].dvp[p][r];
#pragma unroll
          for (int r = 0; r < HR; ++r) a2[VOR + r] = gr[j].dvh[p][r];
#pragma unroll
          for (int r = 0; r < HR; ++r) b2[r] = c[j].vh[p][r];
#pragma unroll
          for (int s_ = 0; s_ < VSTEPS; ++s_) b2[HR + s_] = bv[j][p][s_];
          transpose_slots<VOR + HR>(a2, A2[j], lane, tscr + j * TSCR_FLOATS);
          transpose_slots<HR + VSTEPS>(b2, B2[j], lane, tscr + j * TSCR_FLOATS);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) outer_items<1, 1, BF>(A2[j], B2[j], w.vv);
      }
    } else {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        f4 AH[TN][1], BI[TN][1];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float ah[HR], bin[VSTEPS];
#pragma unroll
          for (int r = 0; r < HR; ++r) ah[r] = gr[j].dvh[p][r];
#pragma unroll
          for (int s = 0; s < VSTEPS; ++s) bin[s] = bv[j][p][s];
          transpose_slots<HR>(ah, AH[j], lane, tscr + j * TSCR_FLOATS);
          transpose_slots<VSTEPS>(bin, BI[j], lane, tscr + j * TSCR_FLOATS);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) outer_items<1, 1, BF>(AH[j], BI[j], w.vh_);
      }
    }
  }
  // once per wave: the accumulated blocks -> this wave's private arena-layout gradient block `gblk` (zeroed at kernel start)
  template <class Acc>
  static __device__ __forceinline__ void wacc_flush(const WAcc& w, float* gblk, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const bool first = false;
    flush_slots<Acc, Segs<Seg<P1, 0, SO>>, WsCols, OT, NBT>(gblk + A::ws(NT), first, K, w.ws, lane);
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      float tot[4];
      float* q[4];
      bool on[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tot[r] = row_total(w.bs[t][r]);
        q[r] = gblk + A::bs(NT) + 16 * t + 4 * g + r;
        on[r] = i == 15;
      }
      add_where<Acc, 4>(q, on, tot);
    }
    if (VO > 0) {
      flush_slots<Acc, Segs<Seg<P2, 0, (VO > 0 ? VO : 1)>>, Segs<Seg<P1, 0, SO>>, 1, OT>(gblk + A::wsv(NT), first, SO, w.wsv, lane);
      float tot[VOR];
      float* q[VOR];
      bool on[VOR];
#pragma unroll
      for (int r = 0; r < VOR; ++r) {
        tot[r] = row_total(w.bsv[r]);
        q[r] = gblk + A::bsv(NT) + 4 * r + g;
        on[r] = i == 15 && 4 * r + g < VO;
      }
      add_where<Acc, VOR>(q, on, tot);
      if constexpr (!PACK_V)
        flush_slots<Acc, Segs<Seg<P2, 0, (VO > 0 ? VO : 1)>>, Segs<Seg<P2, 0, H>>, 1, 1>(gblk + A::wv(NT), first, H, w.vv, lane);
    }
    if constexpr (PACK_V) {
      flush_slots<Acc, Segs<Seg<P2, 0, (VO > 0 ? VO : 1)>, SegNone<HR>>, Segs<Seg<P2, 0, H>, SegNone<VSTEPS>>, 1, 1>(
          gblk + A::wv(NT), first, H, w.vv, lane);
      flush_slots<Acc, Segs<SegNone<VOR>, Seg<P2, 0, H>>, Segs<SegNone<HR>, VSegs>, 1, 1>(gblk, first, VI, w.vv, lane);
    } else {
      flush_slots<Acc, Segs<Seg<P2, 0, H>>, VSegs, 1, 1>(gblk, first, VI, w.vh_, lane);
    }
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

// Backward of ln_quad.  x / v: the tuple BEFORE normalisation; ds / dv: in =
// gradient of the normalised tuple, out = gradient of x / v.  The per-lane
// contributions to d(gamma), d(beta) are returned for a row reduction.
template <int S, int NV>
__device__ __forceinline__ void ln_quad_bwd(const float* ln, int lane, const f4 (&x)[S / 16],
                                            const float (&v)[3][ceil4(NV)], f4 (&ds)[S / 16],
                                            float (&dv)[3][ceil4(NV)], f4 (&dgamma)[S / 16], f4 (&dbeta)[S / 16]) {
  const int g = lane >> 4;
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < S / 16; ++t) sum += x[t][0] + x[t][1] + x[t][2] + x[t][3];
  const float mean = quad_sum(sum) * (1.0f / S);
  float var = 0.f;
#pragma unroll
  for (int t = 0; t < S / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float d = x[t][r] - mean; var = fmaf(d, d, var); }
  const float rstd = gvp::f_rsqrt(quad_sum(var) * (1.0f / S) + gvp::kLnEps);
  f4 xh[S / 16], dxh[S / 16];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int t = 0; t < S / 16; ++t) {
    const f4 ga = *reinterpret_cast<const f4*>(ln + 16 * t + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xh[t][r] = (x[t][r] - mean) * rstd;
      dgamma[t][r] = ds[t][r] * xh[t][r];
      dbeta[t][r] = ds[t][r];
      dxh[t][r] = ds[t][r] * ga[r];
      m1 += dxh[t][r];
      m2 = fmaf(dxh[t][r], xh[t][r], m2);
    }
  }
  m1 = quad_sum(m1) * (1.0f / S);
  m2 = quad_sum(m2) * (1.0f / S);
#pragma unroll
  for (int t = 0; t < S / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[t][r] = rstd * (dxh[t][r] - m1 - xh[t][r] * m2);
  // vectors: v' = v * rvn, rvn = (mean_c max(|v_c|^2, eps))^-1/2
  float n2s = 0.f, dot = 0.f;
  bool free_[ceil4(NV)];
#pragma unroll
  for (int r = 0; r < ceil4(NV); ++r) {
    free_[r] = false;
    if (4 * r + g < NV) {
      const float n2 = v[0][r] * v[0][r] + v[1][r] * v[1][r] + v[2][r] * v[2][r];
      free_[r] = n2 > gvp::kNormEps;
      n2s += gvp::f_max(n2, gvp::kNormEps);
      dot += dv[0][r] * v[0][r] + dv[1][r] * v[1][r] + dv[2][r] * v[2][r];
    }
  }
  const float rvn = gvp::f_rsqrt(quad_sum(n2s) * (1.0f / NV));
  const float k = quad_sum(dot) * rvn * rvn * rvn * (1.0f / NV);
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int r = 0; r < ceil4(NV); ++r) dv[p][r] = rvn * dv[p][r] - (free_[r] ? k * v[p][r] : 0.f);
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

// The same five GVPs of a GVPConvLayer for the tile policy ST (Io<ST>, LayerKind<ST>): image slices, arena blocks and
// gradient blocks have the SAME size and layout for every kind, only the tile arithmetic differs.
//   message_func.{0,1} / ff_func.0 : `activations`, `vector_gate` of the layer (gvp_layers.py:271-288, :355-366)
//   message_func.2 / ff_func.1     : activations (None, None), same vector_gate
template <typename ST> struct LayerKind { static constexpr bool RELU = true; static constexpr int VM = VM_GATE, VM_LAST = VM_GATE; };
template <> struct LayerKind<f32_gvpdef> { static constexpr bool RELU = true; static constexpr int VM = VM_NORM, VM_LAST = VM_NONE; };
template <> struct LayerKind<f32_linear> { static constexpr bool RELU = false; static constexpr int VM = VM_NONE, VM_LAST = VM_NONE; };
template <typename ST>
using Msg0 = GvpQ<0, MS, MV, NS, NV, MV, LayerKind<ST>::RELU, Segs<Seg<P1, 0, MS>>,
                  Segs<Seg<P2, 0, NV>, Seg<P2, NV + EV, NV>, Seg<P2, NV, EV>>, LayerKind<ST>::VM>;
template <typename ST>
using Msg1 = GvpQ<0, NS, NV, NS, NV, NV, LayerKind<ST>::RELU, Segs<Seg<P1, 0, NS>>, Segs<Seg<P2, 0, NV>>, LayerKind<ST>::VM>;
template <typename ST>
using Msg2 = GvpQ<0, NS, NV, NS, NV, NV, false, Segs<Seg<P1, 0, NS>>, Segs<Seg<P2, 0, NV>>, LayerKind<ST>::VM_LAST>;
template <typename ST>
using Ff0 = GvpQ<0, NS, NV, FS, FV, FV, LayerKind<ST>::RELU, Segs<Seg<P1, 0, NS>>, Segs<Seg<P2, 0, NV>>, LayerKind<ST>::VM>;
template <typename ST>
using Ff1 = GvpQ<0, FS, FV, NS, NV, FV, false, Segs<Seg<P1, 0, FS>>, Segs<Seg<P2, 0, FV>>, LayerKind<ST>::VM_LAST>;
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
  static __host__ __device__ int fwd_total(int num_convs) { return head(num_convs) + HD_SIZE; }
  // transposed-fragment slices for the backward kernels, appended after the forward image:
  //   embT | per layer [convT = edgeT msg0T msg1T msg2T | nodeT = ff0T ff1T] | headT
  static constexpr int TE_SIZE = QNode<NTN>::SIZE_T;
  static constexpr int TC_EDGE = 0, TC_M0 = QEdge<NTE>::SIZE_T, TC_M1 = TC_M0 + QMsg0::SIZE_T,
                       TC_M2 = TC_M1 + QMsg1::SIZE_T, TC_SIZE = TC_M2 + QMsg2::SIZE_T;
  static constexpr int TN_FF0 = 0, TN_FF1 = QFf0::SIZE_T, TN_SIZE = TN_FF1 + QFf1::SIZE_T;
  static constexpr int TH_SIZE = QHead::SIZE_T;
  static __host__ __device__ int embT(int nc) { return fwd_total(nc); }
  static __host__ __device__ int convT(int nc, int l) { return embT(nc) + TE_SIZE + l * (TC_SIZE + TN_SIZE); }
  static __host__ __device__ int nodeT(int nc, int l) { return convT(nc, l) + TC_SIZE; }
  static __host__ __device__ int headT(int nc) { return convT(nc, nc); }
  static __host__ __device__ int total(int nc) { return headT(nc) + TH_SIZE; }
};

}  // namespace gq
