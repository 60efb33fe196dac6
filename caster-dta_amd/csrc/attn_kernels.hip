// attn_kernels.hip -- varlen multi-head cross attention between the residues and the atoms of each pair
// (SURVEY 8 f-1).  Replaces, for the shipped CrossAttentionModule (embed 128, 8 heads -> head_dim 16), what
// nn.MultiheadAttention does between its input and output projections on `to_dense_batch`-padded tensors
// (joint_gnn.py:206-215, :379-380): per pair b and head h
//
//     O[q] = softmax_k( scale * Q[q] . K[k] ) V[k],   q in rows q_ptr[b]..q_ptr[b+1], k in rows k_ptr[b]..k_ptr[b+1]
//
// on the COMPACT row arrays ([N, 128] residues / [Na, 128] atoms with PyG ptr offsets): no padding to the longest
// protein, no [B, heads, Rmax, Amax] score tensor, no key-padding mask -- a pair only ever sees its own rows.
//
// Work unit: one wave per (16-query tile of one pair, head).  head_dim 16 = one v_mfma_f32_16x16x4_f32 tile:
//     S^T[key][q]  = sum_d K[key][d] Qs[q][d]    A = K rows  (lane (m = key, g): 4 dims as one float4), B = Qs rows
//     O^T[d][q]   += sum_key V[key][d] P[q][key]  A = V^T     (lane (m = d, g), k-slot r <-> key 4g + r), B = P
// The D tile of the first product (lane (n = q, g) holds keys 4g + r) IS the B operand of the second: scores never
// leave registers.  Online softmax over the key tiles (running max / sum per query; the 4 lanes of a query agree
// through two xor-shuffles), exact fp32 throughout.  The backward is two kernels without atomics: d Q with the same
// orientation (waves own query tiles), d K / d V with keys on the lanes (waves own key tiles and loop over the
// pair's query tiles); both recompute P from the saved log-sum-exp.
//
// Both directions of the module (residues -> atoms and atoms -> residues) go in ONE launch (two "problems").
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/caster_gvp.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int WAVE = 64, TILE = 16, HD = 16, WPB = 4, TPB = WAVE * WPB;

__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// exp through v_exp_f32 (2^x of x * log2 e): one multiply and one transcendental instead of expf's range reduction
// (~12 VALU); every argument here is <= 0 or a difference from a saved log-sum-exp, nowhere near overflow.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float quad_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16));
  return fmaxf(x, __shfl_xor(x, 32));
}
__device__ __forceinline__ float quad_sum(float x) {
  x += __shfl_xor(x, 16);
  return x + __shfl_xor(x, 32);
}

struct Prob {
  const float* q; const float* k; const float* v;         // [Nq][E], [Nk][E], [Nk][E]
  const int64_t* q_ptr; const int64_t* k_ptr;             // [B + 1]
  float* out; float* lse;                                  // [Nq][E], [Nq][H]
  const float* g_out; float* delta;                        // backward: d O [Nq][E]; delta [Nq][H] = sum_d dO . O
  float* g_q; float* g_k; float* g_v;
  float* w; int64_t w_lq, w_lk;                            // dense weights [B][w_lq][w_lk] (inference only)
  int32_t units_q, units_k;                                // upper bounds: (ceil(Nq/16) + B) and (ceil(Nk/16) + B)
};
struct Args { Prob p[2]; int nprob; int B; int H; int E; float scale; };

// Which pair owns tile `t` when every pair b contributes ceil((ptr[b+1] - ptr[b]) / 16) consecutive tiles?
// Wave-parallel prefix scan over the pairs, 64 at a time (B is a few dozen to a few hundred).  Returns false past
// the last tile.  All lanes return the same (b, first row of the tile, end row of the pair).
__device__ __forceinline__ bool locate(const int64_t* __restrict__ ptr, int B, int t, int lane, int& b, int64_t& row0,
                                       int64_t& row_end) {
  int base = 0;
  for (int c = 0; c < B; c += WAVE) {
    const int i = c + lane;
    const int64_t lo = i < B ? ptr[i] : 0, hi = i < B ? ptr[i + 1] : 0;
    const int tiles = (int)((hi - lo + TILE - 1) / TILE);
    int incl = tiles;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const int y = __shfl_up(incl, d);
      if (lane >= d) incl += y;
    }
    const int total = __shfl(incl, WAVE - 1);
    if (t < base + total) {
      const bool mine = t >= base + incl - tiles && t < base + incl;
      const unsigned long long vote = __ballot(mine);
      const int src = __ffsll((long long)vote) - 1;
      b = c + src;
      const int first = __shfl(base + incl - tiles, src);
      const int64_t plo = __shfl((long long)lo, src), phi = __shfl((long long)hi, src);
      row0 = plo + (int64_t)(t - first) * TILE;
      row_end = phi;
      return true;
    }
    base += total;
  }
  return false;
}

// The problem a wave works on, BY VALUE and selected with a wave-uniform index: `const Prob& P = a.p[pi]` with an index the
// compiler cannot prove uniform turned every P.k / P.v / ... inside the loops into a vector load of the pointer from the
// kernel-argument segment followed by s_waitcnt vmcnt(0) -- five dependent load pairs per key tile, 70-83 us per launch
// for 6 us of matrix work (profiles/r03/kernel_stats_davis_b64_joint_before.csv).
__device__ __forceinline__ Prob pick(const Args& a, int pi) { return pi ? a.p[1] : a.p[0]; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(TPB) void attn_fwd_kernel(Args a) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  int64_t u = (int64_t)blockIdx.x * WPB + wave_id();
  int pi = 0;
  if (u >= (int64_t)a.p[0].units_q * a.H) { u -= (int64_t)a.p[0].units_q * a.H; pi = 1; }
  if (pi >= a.nprob) return;
  const Prob P = pick(a, pi);
  if (u >= (int64_t)P.units_q * a.H) return;
  const int h = (int)(u % a.H), t = (int)(u / a.H);
  int b;
  int64_t q0_, q_end_;
  if (!locate(P.q_ptr, a.B, t, lane, b, q0_, q_end_)) return;
  // rows and element offsets in 32 bits (the host checks rows x E < 2^31): a load is then base (SGPR pair) + one 32-bit
  // VGPR offset instead of a 64-bit multiply-add chain per address
  const int q0 = (int)q0_, q_end = (int)q_end_;
  const int k0 = (int)P.k_ptr[b], k_end = (int)P.k_ptr[b + 1];
  const int qr = q0 + n;
  const bool qv = qr < q_end;
  const unsigned E = a.E, col = h * HD;
  f4 qs = {0.f, 0.f, 0.f, 0.f};
  if (qv) qs = *reinterpret_cast<const f4*>(P.q + ((unsigned)qr * E + col + 4 * g)) * a.scale;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  // The key-tile loop is a dependent chain (loads -> 4 MFMAs -> exp -> 4 MFMAs) and a pair's atoms attend to ~19 residue
  // tiles: the NEXT tile's K / V rows are loaded while the current one is computed (one global-load latency per wave
  // instead of one per key tile).
  //   S^T tile: lane (m = key, g) supplies K[key][4g..4g+3]; k-slot (s, g) <-> dim 4g + s on both operands
  //   V^T operand: lane (m = d, g), slot r <-> key kb + 4g + r
  // (loads are UNCONDITIONAL on rows clamped to the pair's last key and zeroed afterwards: with loads under lane masks
  // the compiler cannot count what is outstanding and waits for vmcnt(0), i.e. for the prefetch it has just issued)
  // element offsets of the tile's five loads, advanced by 16 rows per call and clamped to the pair's last key with one
  // 32-bit min each (offsets are monotone in the row)
  const unsigned k_step = TILE * E;
  const unsigned ko_last = (unsigned)(k_end - 1) * E + col + 4 * g, vo_last = (unsigned)(k_end - 1) * E + col + n;
  unsigned ko = (unsigned)(k0 + n) * E + col + 4 * g;
  unsigned vo[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) vo[r] = (unsigned)(k0 + 4 * g + r) * E + col + n;
  // Rows past the pair's last key are never zeroed: the clamp makes them copies of the last key's (finite) row, the tail
  // iteration sets their scores to -inf, and exp2(-inf) = 0 removes them from both sums.  Only the LAST tile of a pair can
  // be partial, so the loop over full tiles carries no masks at all (peeled tail).
  auto load_tile = [&](f4& kk, float (&va)[4]) {                   // the next 16 keys, call by call
    kk = *reinterpret_cast<const f4*>(P.k + (ko < ko_last ? ko : ko_last));
    ko += k_step;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      va[r] = P.v[vo[r] < vo_last ? vo[r] : vo_last];
      vo[r] += k_step;
    }
  };
  f4 kk_n = {0.f, 0.f, 0.f, 0.f};
  float va_n[4] = {0.f, 0.f, 0.f, 0.f};
  if (k0 < k_end) load_tile(kk_n, va_n);
  auto tile = [&](int kb, auto tail) {
    const f4 kk = kk_n;
    const float va[4] = {va_n[0], va_n[1], va_n[2], va_n[3]};
    load_tile(kk_n, va_n);                                  // past the end: copies of the last key, never used
    f4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) st = mfma(kk[s], qs[s], st);
    if constexpr (decltype(tail)::value) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kb + 4 * g + r >= k_end) st[r] = -INFINITY;
    }
    const float m_new = fmaxf(m, quad_max(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]))));   // finite: a tile holds a valid key
    const float alpha = fexp(m - m_new);                    // m = -inf before the first tile: exp2(-inf) = 0
    float p[4], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[r] = fexp(st[r] - m_new); ps += p[r]; }
    l = l * alpha + quad_sum(ps);
    acc *= alpha;
    m = m_new;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = mfma(va[r], p[r], acc);
  };
  int kb = k0;
  for (; kb + TILE <= k_end; kb += TILE) tile(kb, std::false_type{});
  if (kb < k_end) tile(kb, std::true_type{});
  if (qv) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    *reinterpret_cast<f4*>(P.out + ((unsigned)qr * E + col + 4 * g)) = acc * inv;
    if (g == 0 && P.lse) P.lse[(unsigned)qr * a.H + h] = (l > 0.f) ? m + logf(l) : -INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------ backward: d Q
// Same orientation as the forward.  Also writes delta[q][h] = sum_d dO[q][d] O[q][d] for the d K / d V kernel.
__global__ __launch_bounds__(TPB) void attn_bwd_dq_kernel(Args a) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  int64_t u = (int64_t)blockIdx.x * WPB + wave_id();
  int pi = 0;
  if (u >= (int64_t)a.p[0].units_q * a.H) { u -= (int64_t)a.p[0].units_q * a.H; pi = 1; }
  if (pi >= a.nprob) return;
  const Prob P = pick(a, pi);
  if (u >= (int64_t)P.units_q * a.H) return;
  const int h = (int)(u % a.H), t = (int)(u / a.H);
  int b;
  int64_t q0_, q_end_;
  if (!locate(P.q_ptr, a.B, t, lane, b, q0_, q_end_)) return;
  const int q0 = (int)q0_, q_end = (int)q_end_;                 // 32-bit rows / offsets, see attn_fwd_kernel
  const int k0 = (int)P.k_ptr[b], k_end = (int)P.k_ptr[b + 1];
  const int qr = q0 + n;
  const bool qv = qr < q_end;
  const unsigned E = a.E, col = h * HD;
  f4 qs = {0.f, 0.f, 0.f, 0.f}, go = {0.f, 0.f, 0.f, 0.f};
  float lse = 0.f, dl = 0.f;
  if (qv) {
    const unsigned qo = (unsigned)qr * E + col + 4 * g;
    qs = *reinterpret_cast<const f4*>(P.q + qo) * a.scale;
    go = *reinterpret_cast<const f4*>(P.g_out + qo);
    const f4 o = *reinterpret_cast<const f4*>(P.out + qo);
    lse = P.lse[(unsigned)qr * a.H + h];
    dl = go[0] * o[0] + go[1] * o[1] + go[2] * o[2] + go[3] * o[3];
  }
  dl = quad_sum(dl);
  if (qv && g == 0) P.delta[(unsigned)qr * a.H + h] = dl;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  // next key tile prefetched while the current one is computed (see attn_fwd_kernel)
  //   ka: K^T operand for d Qs: lane (m = d, g), slot r <-> key kb + 4g + r
  const unsigned k_step = TILE * E;                                // offsets advanced per call and clamped, see attn_fwd_kernel
  const unsigned ko_last = (unsigned)(k_end - 1) * E + col + 4 * g, to_last = (unsigned)(k_end - 1) * E + col + n;
  unsigned ko = (unsigned)(k0 + n) * E + col + 4 * g;
  unsigned to[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) to[r] = (unsigned)(k0 + 4 * g + r) * E + col + n;
  auto load_tile = [&](f4& kk, f4& vv, float (&ka)[4]) {           // the next 16 keys; no zeroing, tail peeled (attn_fwd_kernel)
    const unsigned o = ko < ko_last ? ko : ko_last;
    kk = *reinterpret_cast<const f4*>(P.k + o);
    vv = *reinterpret_cast<const f4*>(P.v + o);
    ko += k_step;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ka[r] = P.k[to[r] < to_last ? to[r] : to_last];
      to[r] += k_step;
    }
  };
  f4 kk_n = {0.f, 0.f, 0.f, 0.f}, vv_n = {0.f, 0.f, 0.f, 0.f};
  float ka_n[4] = {0.f, 0.f, 0.f, 0.f};
  if (k0 < k_end) load_tile(kk_n, vv_n, ka_n);
  auto tile = [&](int kb, auto tail) {
    const f4 kk = kk_n, vv = vv_n;
    const float ka[4] = {ka_n[0], ka_n[1], ka_n[2], ka_n[3]};
    load_tile(kk_n, vv_n, ka_n);
    f4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) { st = mfma(kk[s], qs[s], st); dp = mfma(vv[s], go[s], dp); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // (a lane without a query computes a finite column that is never stored; keys past the pair's end only in the tail)
      float pr = fexp(st[r] - lse);
      if constexpr (decltype(tail)::value) pr = kb + 4 * g + r < k_end ? pr : 0.f;
      acc = mfma(ka[r], pr * (dp[r] - dl), acc);
    }
  };
  int kb = k0;
  for (; kb + TILE <= k_end; kb += TILE) tile(kb, std::false_type{});
  if (kb < k_end) tile(kb, std::true_type{});
  if (qv) *reinterpret_cast<f4*>(P.g_q + ((unsigned)qr * E + col + 4 * g)) = acc * a.scale;
}

// ------------------------------------------------------------------------------------------------ backward: d K, d V
// Keys on the lanes: a wave owns a 16-key tile of one pair and loops over the pair's query tiles.
//   S[q][key]   = D[m = q][n = key]: A = Qs rows (lane (m = q, g): float4 of dims), B = K rows
//   dP[q][key]  = D[m = q][n = key]: A = dO rows, B = V rows
//   dV[key][d]  = D[m = d][n = key] += sum_q dO[q][d] P[q][key]   A: lane (m = d, g), slot r <-> query qb + 4g + r
//   dK[key][d]  = D[m = d][n = key] += sum_q Qs[q][d] dS[q][key]
__global__ __launch_bounds__(TPB) void attn_bwd_dkv_kernel(Args a) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  int64_t u = (int64_t)blockIdx.x * WPB + wave_id();
  int pi = 0;
  if (u >= (int64_t)a.p[0].units_k * a.H) { u -= (int64_t)a.p[0].units_k * a.H; pi = 1; }
  if (pi >= a.nprob) return;
  const Prob P = pick(a, pi);
  if (u >= (int64_t)P.units_k * a.H) return;
  const int h = (int)(u % a.H), t = (int)(u / a.H);
  int b;
  int64_t kt0_, k_end_;
  if (!locate(P.k_ptr, a.B, t, lane, b, kt0_, k_end_)) return;
  const int kt0 = (int)kt0_, k_end = (int)k_end_;               // 32-bit rows / offsets, see attn_fwd_kernel
  const int q0 = (int)P.q_ptr[b], q_end = (int)P.q_ptr[b + 1];
  const int kr = kt0 + n;
  const bool kv = kr < k_end;
  const unsigned E = a.E, col = h * HD, H = a.H;
  f4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
  if (kv) {
    kk = *reinterpret_cast<const f4*>(P.k + ((unsigned)kr * E + col + 4 * g));
    vv = *reinterpret_cast<const f4*>(P.v + ((unsigned)kr * E + col + 4 * g));
  }
  f4 accv = {0.f, 0.f, 0.f, 0.f}, acck = {0.f, 0.f, 0.f, 0.f};
  // next query tile prefetched while the current one is computed (an atom-key tile loops over ~19 residue tiles)
  struct QT { f4 qa, ga; float lse[4], dl[4], qt[4], gt[4]; };
  const unsigned q_step = TILE * E, h_step = TILE * H;              // offsets advanced per call and clamped, see attn_fwd_kernel
  const unsigned qo_last = (unsigned)(q_end - 1) * E + col + 4 * g, to_last = (unsigned)(q_end - 1) * E + col + n,
                 ho_last = (unsigned)(q_end - 1) * H + h;
  unsigned qo = (unsigned)(q0 + n) * E + col + 4 * g;
  unsigned to[4], ho[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { to[r] = (unsigned)(q0 + 4 * g + r) * E + col + n; ho[r] = (unsigned)(q0 + 4 * g + r) * H + h; }
  auto load_tile = [&](QT& T) {                                     // the next 16 queries; no zeroing, tail peeled (attn_fwd_kernel)
    const unsigned qmo = qo < qo_last ? qo : qo_last;               // this lane's row as the M index of the two score products
    T.qa = *reinterpret_cast<const f4*>(P.q + qmo) * a.scale;
    T.ga = *reinterpret_cast<const f4*>(P.g_out + qmo);
    qo += q_step;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                           // per k-slot r <-> query qb + 4g + r
      const unsigned tc = to[r] < to_last ? to[r] : to_last, hc = ho[r] < ho_last ? ho[r] : ho_last;
      to[r] += q_step; ho[r] += h_step;
      T.lse[r] = P.lse[hc];
      T.dl[r] = P.delta[hc];
      T.qt[r] = P.q[tc] * a.scale;                          // Qs^T operand: lane (m = d, g)
      T.gt[r] = P.g_out[tc];                                // dO^T operand
    }
  };
  QT nxt = {};
  if (q0 < q_end) load_tile(nxt);
  auto tile = [&](int qb, auto tail) {
    const QT cur = nxt;
    load_tile(nxt);
    const f4 qa = cur.qa, ga = cur.ga;
    const float (&lse)[4] = cur.lse, (&dl)[4] = cur.dl, (&qt)[4] = cur.qt, (&gt)[4] = cur.gt;
    f4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 4; ++x) { s = mfma(qa[x], kk[x], s); dp = mfma(ga[x], vv[x], dp); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // (a lane without a key computes a finite column that is never stored; queries past the pair's end only in the tail)
      float pr = fexp(s[r] - lse[r]);
      if constexpr (decltype(tail)::value) pr = qb + 4 * g + r < q_end ? pr : 0.f;
      accv = mfma(gt[r], pr, accv);
      acck = mfma(qt[r], pr * (dp[r] - dl[r]), acck);
    }
  };
  int qb = q0;
  for (; qb + TILE <= q_end; qb += TILE) tile(qb, std::false_type{});
  if (qb < q_end) tile(qb, std::true_type{});
  if (kv) {
    *reinterpret_cast<f4*>(P.g_v + ((unsigned)kr * E + col + 4 * g)) = accv;
    *reinterpret_cast<f4*>(P.g_k + ((unsigned)kr * E + col + 4 * g)) = acck;
  }
}

// ------------------------------------------------------------------------------------------------ head-averaged weights
// nn.MultiheadAttention(need_weights=True, average_attn_weights=True): w[b][q][k] = mean_h P_h[q][k], written straight
// into the reference's dense layout [B][w_lq][w_lk] (zero elsewhere; the caller zero-fills).  Inference only
// (inference/evaluation.py:43-66 slices it per pair); one wave per 16-query tile, all heads.
__global__ __launch_bounds__(TPB) void attn_weights_kernel(Args a) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  int64_t t = (int64_t)blockIdx.x * WPB + wave_id();
  int pi = 0;
  if (t >= a.p[0].units_q) { t -= a.p[0].units_q; pi = 1; }
  if (pi >= a.nprob) return;
  const Prob P = pick(a, pi);
  if (t >= P.units_q || !P.w) return;
  int b;
  int64_t q0, q_end;
  if (!locate(P.q_ptr, a.B, (int)t, lane, b, q0, q_end)) return;
  const int64_t k0 = P.k_ptr[b], k_end = P.k_ptr[b + 1], qbase = P.q_ptr[b];
  const int64_t qr = q0 + n;
  const bool qv = qr < q_end;
  const int E = a.E;
  const float invh = 1.0f / (float)a.H;
  for (int64_t kb = k0; kb < k_end; kb += TILE) {
    const int64_t kr = kb + n;
    f4 w = {0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < a.H; ++h) {
      const int col = h * HD;
      f4 qs = {0.f, 0.f, 0.f, 0.f}, kk = {0.f, 0.f, 0.f, 0.f};
      float lse = 0.f;
      if (qv) { qs = *reinterpret_cast<const f4*>(P.q + qr * E + col + 4 * g) * a.scale; lse = P.lse[qr * a.H + h]; }
      if (kr < k_end) kk = *reinterpret_cast<const f4*>(P.k + kr * E + col + 4 * g);
      f4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) st = mfma(kk[s], qs[s], st);
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] += (qv && kb + 4 * g + r < k_end) ? fexp(st[r] - lse) * invh : 0.f;
    }
    if (qv) {
      float* row = P.w + ((int64_t)b * P.w_lq + (qr - qbase)) * P.w_lk + (kb - k0) + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kb + 4 * g + r < k_end) row[r] = w[r];
    }
  }
}

int status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// `long_first`: which of two problems gets the low workgroup indices -- the one whose waves loop longest (1: over the
// keys, the forward and d Q kernels; 2: over the queries, the d K / d V kernel).  Workgroups start in index order: with
// the residues -> atoms direction first, its 10k three-tile waves filled the chip and the 1.8k nineteen-tile waves of the
// other direction started last and ran alone at the end of the launch.
int fill(Args& a, const cgvp_attn_problem* probs, int nprob, int64_t B, int H, int long_first = 0) {
  if (!probs || nprob < 1 || nprob > 2 || B < 0 || H < 1) return CGVP_ERR_BAD_ARG;
  a.nprob = nprob; a.B = (int)B; a.H = H; a.E = H * HD;
  const bool swap = nprob == 2 && ((long_first == 1 && probs[1].num_k > probs[0].num_k) ||
                                   (long_first == 2 && probs[1].num_q > probs[0].num_q));
  for (int i = 0; i < 2; ++i) {
    Prob& p = a.p[i];
    p = Prob{};
    if (i >= nprob) continue;
    const cgvp_attn_problem& s = probs[swap ? 1 - i : i];
    if (s.num_q < 0 || s.num_k < 0 || !s.q_ptr || !s.k_ptr) return CGVP_ERR_BAD_ARG;
    if ((s.num_q + TILE) * (int64_t)(H * HD) >= ((int64_t)1 << 31) || (s.num_k + TILE) * (int64_t)(H * HD) >= ((int64_t)1 << 31))
      return CGVP_ERR_BAD_ARG;                             // the kernels address rows with 32-bit element offsets
    if (s.num_q > 0 && (!s.q || !s.out || !s.lse)) return CGVP_ERR_BAD_ARG;
    if (s.num_k > 0 && (!s.k || !s.v)) return CGVP_ERR_BAD_ARG;
    const void* al[] = {s.q, s.k, s.v, s.out, s.g_out, s.g_q, s.g_k, s.g_v};
    for (const void* x : al) if ((uintptr_t)x & 15) return CGVP_ERR_BAD_ARG;
    p.q = s.q; p.k = s.k; p.v = s.v; p.q_ptr = s.q_ptr; p.k_ptr = s.k_ptr; p.out = s.out; p.lse = s.lse;
    p.g_out = s.g_out; p.delta = s.delta; p.g_q = s.g_q; p.g_k = s.g_k; p.g_v = s.g_v;
    p.w = s.weights; p.w_lq = s.weights_lq; p.w_lk = s.weights_lk;
    p.units_q = (int32_t)((s.num_q + TILE - 1) / TILE + B);
    p.units_k = (int32_t)((s.num_k + TILE - 1) / TILE + B);
  }
  return 0;
}

}  // namespace

extern "C" {

int cgvp_attn_fwd(const cgvp_attn_problem* probs, int32_t num_problems, int64_t num_pairs, int32_t heads, float scale,
                  void* stream) {
  Args a;
  if (int rc = fill(a, probs, num_problems, num_pairs, heads, 1)) return rc;
  a.scale = scale;
  if (num_pairs == 0) return 0;
  const int64_t units = ((int64_t)a.p[0].units_q + a.p[1].units_q) * heads;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)((units + WPB - 1) / WPB)), dim3(TPB), 0, (hipStream_t)stream, a);
  return status();
}

int cgvp_attn_weights(const cgvp_attn_problem* probs, int32_t num_problems, int64_t num_pairs, int32_t heads, float scale,
                      void* stream) {
  Args a;
  if (int rc = fill(a, probs, num_problems, num_pairs, heads)) return rc;
  a.scale = scale;
  for (int i = 0; i < num_problems; ++i)
    if (!probs[i].weights || probs[i].weights_lq < 0 || probs[i].weights_lk < 0) return CGVP_ERR_BAD_ARG;
  if (num_pairs == 0) return 0;
  const int64_t tiles = (int64_t)a.p[0].units_q + a.p[1].units_q;
  hipLaunchKernelGGL(attn_weights_kernel, dim3((unsigned)((tiles + WPB - 1) / WPB)), dim3(TPB), 0, (hipStream_t)stream, a);
  return status();
}

int cgvp_attn_bwd(const cgvp_attn_problem* probs, int32_t num_problems, int64_t num_pairs, int32_t heads, float scale,
                  void* stream) {
  Args a, ak;
  if (int rc = fill(a, probs, num_problems, num_pairs, heads, 1)) return rc;
  if (int rc = fill(ak, probs, num_problems, num_pairs, heads, 2)) return rc;
  a.scale = ak.scale = scale;
  for (int i = 0; i < num_problems; ++i) {
    const cgvp_attn_problem& s = probs[i];
    if ((s.num_q > 0 && (!s.g_out || !s.g_q || !s.delta)) || (s.num_k > 0 && (!s.g_k || !s.g_v))) return CGVP_ERR_BAD_ARG;
  }
  if (num_pairs == 0) return 0;
  const int64_t uq = ((int64_t)a.p[0].units_q + a.p[1].units_q) * heads;
  const int64_t uk = ((int64_t)a.p[0].units_k + a.p[1].units_k) * heads;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)((uq + WPB - 1) / WPB)), dim3(TPB), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)((uk + WPB - 1) / WPB)), dim3(TPB), 0, (hipStream_t)stream, ak);
  return status();
}

}  // extern "C"
