// Test harness (never loaded by the product): the kernels' counter-based dropout generator compiled with g++.
#include "../../caster-dta_amd/csrc/gvp_rng.h"

extern "C" void host_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  uint32_t o[4];
  gvp::philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], o);
  for (int k = 0; k < 4; ++k) out[k] = o[k];
}

extern "C" void host_dropout_mask(unsigned long long seed, unsigned long long offset, int stream, long long N, int width,
                                  float p, float* out) {
  for (long long n = 0; n < N; ++n) {
    if (width == 20) {                       // protein row [16 scalar | 4 vector-channel]: quarter g from one call
      for (int g = 0; g < 4; ++g) {
        float fs[4], fv;
        gvp::dropout_row20(seed, offset, stream, n, g, p, fs, fv);
        for (int k = 0; k < 4; ++k) out[n * width + 4 * g + k] = fs[k];
        out[n * width + 16 + g] = fv;
      }
      continue;
    }
    for (int blk = 0; blk < width / 4; ++blk) {
      float f[4];
      gvp::dropout4(seed, offset, stream, n, blk, p, f);
      for (int k = 0; k < 4; ++k) out[n * width + 4 * blk + k] = f[k];
    }
  }
}
