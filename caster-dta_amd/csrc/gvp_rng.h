// gvp_rng.h -- counter-based dropout masks generated inside the kernels (Philox4x32-10).
//
// gvp_layers.Dropout (gvp_layers.py:187-219) draws one Bernoulli(1-p) factor per (node, scalar channel) and per
// (node, vector channel) -- a vector channel's xyz share the factor -- scaled by 1/(1-p).  Which uniform a
// (node, channel) gets is a pure function of (seed, offset, stream, node, channel): the forward kernel and the
// backward kernel that recomputes the stage regenerate the same factors, nothing is stored in HBM and no mask
// launch exists.  `stream` separates the masks of one step: protein layer l uses 2l (dropout[0]) and 2l+1
// (dropout[1]); GINE layer l uses l.  Plain integer code: also compiled with g++ for the host tests.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define GVP_RNG_HD __host__ __device__ __forceinline__
#else
#define GVP_RNG_HD inline
#endif

namespace gvp {

// Kernel-argument form of cgvp_rng (include/caster_gvp.h): seed -> device {seed, offset}; NULL = no dropout.
struct RngArgs {
  const unsigned long long* seed;
  float p;
  int stream;
};

template <int ROUNDS>
GVP_RNG_HD void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// the standard 10-round generator (pinned to Random123's known answers in tests/test_host_math.py)
GVP_RNG_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
  philox4x32<10>(c0, c1, c2, c3, k0, k1, out);
}

// The masks use the 7-round variant (the fewest rounds that pass BigCrush in Salmon et al. 2011, "Parallel random
// numbers: as easy as 1, 2, 3") and 16 random bits per factor: gfx950's 32-bit integer multiplies are quarter rate,
// and a residue tile's dropout sits on the latency-bound critical path of the fused layer kernel.  One call gives
// 8 factors: keep iff u16 < round((1 - p) * 65536), i.e. the keep probability is (1 - p) to within 2^-17.
constexpr int kMaskRounds = 7;
GVP_RNG_HD void dropout8_raw(unsigned long long seed, unsigned long long offset, int stream, long long n, int blk, float p,
                             float (&f)[8]) {
  uint32_t u[4];
  philox4x32<kMaskRounds>((uint32_t)n, (uint32_t)((unsigned long long)n >> 32) ^ ((uint32_t)stream << 16) ^ (uint32_t)blk,
                          (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), u);
  const float keep = 1.0f - p, inv = 1.0f / keep;
  const uint32_t thr = (uint32_t)(keep * 65536.0f + 0.5f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = ((u[k] & 0xFFFFu) < thr) ? inv : 0.f;
    f[2 * k + 1] = ((u[k] >> 16) < thr) ? inv : 0.f;
  }
}

// Protein mask row [16 scalar-channel | 4 vector-channel factors] of node n: the quarter `g` of the row that lane
// (item, g) of the quad layout needs -- scalar channels 4g .. 4g+3 and vector channel g -- comes from ONE call.
GVP_RNG_HD void dropout_row20(unsigned long long seed, unsigned long long offset, int stream, long long n, int g, float p,
                              float (&fs)[4], float& fv) {
  float f[8];
  dropout8_raw(seed, offset, stream, n, g, p, f);
  fs[0] = f[0]; fs[1] = f[1]; fs[2] = f[2]; fs[3] = f[3];
  fv = f[4];
}

// GINE mask row of `width` channels: channels 4*blk .. 4*blk+3 of node n (two blocks share one call).
GVP_RNG_HD void dropout4(unsigned long long seed, unsigned long long offset, int stream, long long n, int blk, float p,
                         float (&f)[4]) {
  float f8[8];
  dropout8_raw(seed, offset, stream, n, blk >> 1, p, f8);
#pragma unroll
  for (int k = 0; k < 4; ++k) f[k] = f8[4 * (blk & 1) + k];
}

}  // namespace gvp
