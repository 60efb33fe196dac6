// TEST SCAFFOLDING ONLY -- not part of the product, never shipped or loaded by
// caster-dta_amd.  Compiles csrc/gvp_math.h (the per-lane arithmetic the HIP
// kernels execute) with the host compiler and drives it item by item, so the
// kernel maths can be checked against the oracle in a container without a GPU.
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../caster-dta_amd/csrc/gvp_math.h"

using namespace gvp;

template <int NTN, int NTE>
static void run(const float* P, const EncLayout& L, int num_convs, int mean, const float* x_s,
                const float* x_v, const int64_t* ntypes, const float* e_s, const float* e_v,
                const int64_t* etypes, const int64_t* ei, int64_t N, int64_t E, float* out,
                float* stage_h /* [(1+num_convs)][N][28] */, float* stage_dh /* [num_convs][N][28] */) {
  std::vector<float> h(N * ROW), h2(N * ROW), dh(N * ROW);
  for (int64_t n = 0; n < N; ++n) {
    float xs[NODE_IN_S], xv[NODE_IN_V][3], row[ROW];
    memcpy(xs, x_s + n * NODE_IN_S, sizeof xs);
    memcpy(xv, x_v + n * 3 * NODE_IN_V, sizeof xv);
    node_embed_item<NTN>(P, L, NTN ? (int)ntypes[n] : 0, xs, xv, row);
    memcpy(&h[n * ROW], row, sizeof row);
  }
  memcpy(stage_h, h.data(), sizeof(float) * N * ROW);
  for (int l = 0; l < num_convs; ++l) {
    std::fill(dh.begin(), dh.end(), 0.f);
    std::vector<int> deg(N, 0);
    for (int64_t e = 0; e < E; ++e) {   // original edge order == stable dst-sorted order per target
      const int64_t src = ei[e], dst = ei[E + e];
      float es[EDGE_IN_S], ev[EDGE_IN_V][3], xj[ROW], xi[ROW], m[ROW];
      memcpy(es, e_s + e * EDGE_IN_S, sizeof es);
      memcpy(ev, e_v + e * 3 * EDGE_IN_V, sizeof ev);
      memcpy(xj, &h[src * ROW], sizeof xj);
      memcpy(xi, &h[dst * ROW], sizeof xi);
      conv_message_item<NTE>(P, L, l, NTE ? (int)etypes[e] : 0, es, ev, xj, xi, m);
      for (int k = 0; k < ROW; ++k) dh[dst * ROW + k] += m[k];
      deg[dst]++;
    }
    if (mean)
      for (int64_t n = 0; n < N; ++n)
        for (int k = 0; k < ROW; ++k) dh[n * ROW + k] /= (float)(deg[n] > 1 ? deg[n] : 1);
    memcpy(stage_dh + (int64_t)l * N * ROW, dh.data(), sizeof(float) * N * ROW);
    const bool last = l == num_convs - 1;
    for (int64_t n = 0; n < N; ++n) {
      float x[ROW], d[ROW], row[ROW], o[OUT];
      memcpy(x, &h[n * ROW], sizeof x);
      memcpy(d, &dh[n * ROW], sizeof d);
      if (last) {
        node_update_item<true>(P, L, l, x, d, row, o);
        memcpy(out + n * OUT, o, sizeof o);
      } else {
        node_update_item<false>(P, L, l, x, d, row, o);
      }
      memcpy(&h2[n * ROW], row, sizeof row);
    }
    h.swap(h2);
    memcpy(stage_h + (int64_t)(l + 1) * N * ROW, h.data(), sizeof(float) * N * ROW);
  }
}

extern "C" int host_lba_forward(const float* P, int nt_node, int nt_edge, int num_convs, int mean,
                                const float* x_s, const float* x_v, const int64_t* ntypes,
                                const float* e_s, const float* e_v, const int64_t* etypes,
                                const int64_t* ei, int64_t N, int64_t E, float* out, float* stage_h,
                                float* stage_dh) {
  const EncLayout L = make_layout(nt_node, nt_edge, num_convs);
  if (nt_node == 20 && nt_edge == 1) run<20, 1>(P, L, num_convs, mean, x_s, x_v, ntypes, e_s, e_v, etypes, ei, N, E, out, stage_h, stage_dh);
  else if (nt_node == 0 && nt_edge == 0) run<0, 0>(P, L, num_convs, mean, x_s, x_v, ntypes, e_s, e_v, etypes, ei, N, E, out, stage_h, stage_dh);
  else return -2;
  return L.total;
}
