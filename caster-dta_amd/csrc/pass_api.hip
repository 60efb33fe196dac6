// pass_api.hip -- the WHOLE-PASS entry points of libcaster_gvp.so (include/caster_gvp.h, "WHOLE-PASS ENTRY POINTS"):
// one host call issues the complete launch sequence of an encoder pass into caller-allocated workspaces.
//
//   protein forward  (protein_gnn.py:361-388)   pass_begin [node embed + fragment image + CSR count]
//                                               -> CSR scan / fill / rank -> L x conv layer (+ head)
//   protein backward (autograd of the above)    L x [node_bwd (+ head_bwd), conv_bwd] -> embed_bwd -> edge_bwd -> reduce
//   drug forward     (molecule_gnn.py:254-268)  CSR count [+ generator advance] / scan / fill / rank -> L x GINE layer
//   drug backward                               zero d x buffers -> L x GINE layer backward -> ONE reduce over all layers
//
// Host code only: every kernel lives in the other translation units and is reached through the fine-grained C entry
// points (which validate their arguments) or the quad:: launchers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "gvp_internal.h"

namespace {

// ------------------------------------------------------------------ opt-in kernel timing (cgvp_debug_kernel_timing)
// The one piece of process state in the library (opt-in diagnostic): the flag is atomic, the list is only touched under
// its mutex, so passes issued from several host threads (one per stream) stay re-entrant with timing on.
struct TimedLaunch { hipEvent_t a, b; int kind; };
std::atomic<bool> g_timing{false};
std::mutex g_timed_mutex;
std::vector<TimedLaunch> g_timed;
struct Timed {                 // brackets one launch with events when timing is on
  hipStream_t st; bool on; TimedLaunch t;
  Timed(int kind, void* stream) : st((hipStream_t)stream), on(g_timing.load(std::memory_order_relaxed)) {
    if (!on) return;
    t.kind = kind;
    if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(t.a, st);
  }
  ~Timed() {
    if (!on) return;
    (void)hipEventRecord(t.b, st);
    std::lock_guard<std::mutex> lock(g_timed_mutex);
    g_timed.push_back(t);
  }
};

constexpr int ROW = 28;                 // merged node row [s(16) | v(4x3)]
constexpr int EROW = CGVP_EDGE_ROW;
constexpr int MROW = 20;                // dropout mask row: 16 scalar + 4 vector-channel factors

inline int64_t up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
// A/B switch (environment, read once): CGVP_GINE_RECOMPUTE=1 makes the drug backward recompute its aggregates as it did
// until round 3 instead of reading what the forward saved (the forward then saves nothing).
inline bool gine_recompute() {
  static const bool v = [] { const char* e = getenv("CGVP_GINE_RECOMPUTE"); return e && e[0] == '1'; }();
  return v;
}
inline int num_convs_of(const cgvp_layout& l) { return l.conv_stride > 0 ? (l.ln_out - l.conv0) / l.conv_stride : 0; }
inline int esize(const cgvp_dims* d) { return d->storage == CGVP_BF16 ? 2 : 4; }
inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
template <typename T>
inline T* at(void* base, int64_t off) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + off); }
template <typename T>
inline const T* at(const void* base, int64_t off) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off); }

struct Bump {          // 256-B aligned sub-buffer offsets
  int64_t off = 0;
  int64_t take(int64_t bytes) { const int64_t o = off; off = up(off + (bytes > 0 ? bytes : 0), 256); return o; }
};

inline bool has_tables(const cgvp_lba_batch* b) { return b->rowptr != nullptr; }
inline bool has_tables(const cgvp_gine_batch* b) { return b->rowptr != nullptr; }

// ------------------------------------------------------------------ protein backward workspace
struct LbaBwdWs {
  int64_t slabs, g_e, rows, total;      // per-stage partial weight-gradient slabs | d(edge embedding) per layer | 8 node-row buffers
  int64_t slab_floats, g_e_stride, row_floats;
};
int lba_bwd_layout(const cgvp_dims* dims, const cgvp_layout* layout, int64_t N, int64_t E, LbaBwdWs* w) {
  const int64_t wsz = cgvp_bwd_workspace_floats(dims, layout);
  if (wsz < 0) return (int)wsz;
  const int nc = num_convs_of(*layout);
  Bump b;
  w->slab_floats = wsz;
  w->slabs = b.take((int64_t)(2 * nc + 2) * wsz * 4);
  w->g_e_stride = up((E + 1) * EROW * 4, 256) / 4;
  w->g_e = b.take((int64_t)nc * w->g_e_stride * 4);
  w->row_floats = up(N * ROW * 4, 256) / 4;
  w->rows = b.take(8 * w->row_floats * 4);
  w->total = b.off;
  return 0;
}

// ------------------------------------------------------------------ GINE helpers
inline int gine_layer_floats(int cin, int chid, int cout, int ke) {
  return 1 + chid * cin + chid + cout * chid + cout + cin * ke + cin;
}
int check_gine_cfg(const cgvp_gine_cfg* c) {
  if (!c || c->num_layers < 1 || c->num_layers > CGVP_GINE_MAX_LAYERS) return CGVP_ERR_BAD_ARG;
  if (c->num_ntypes < 0 || c->num_etypes < 0 || c->edge_dim < 0) return CGVP_ERR_BAD_ARG;
  for (int l = 0; l <= c->num_layers; ++l)
    if (c->widths[l] < 1 || c->widths[l] > 64) return CGVP_ERR_UNSUPPORTED_DIMS;
  if (c->widths[0] <= c->num_ntypes) return CGVP_ERR_BAD_ARG;
  return 0;
}
struct GineBwdWs { int64_t slabs, gx, total, slab_floats, gx_floats; };
void gine_bwd_layout(const cgvp_gine_cfg* c, int64_t N, GineBwdWs* w) {
  Bump b;
  w->slab_floats = cgvp_gine_bwd_workspace_floats();
  w->slabs = b.take((int64_t)c->num_layers * w->slab_floats * 4);
  int maxw = 0;
  for (int l = 1; l < c->num_layers; ++l) maxw = c->widths[l] > maxw ? c->widths[l] : maxw;
  w->gx_floats = up(N * maxw * 4, 256) / 4;           // d h_l for every hidden activation, back to back: ONE zero fill
  w->gx = b.take((int64_t)(c->num_layers - 1) * w->gx_floats * 4);
  w->total = b.off;
}

// ------------------------------------------------------------------ batch staging (cgvp_stage_buffers)
struct StageTable { cgvp_stage_item it[CGVP_MAX_STAGE]; };
__global__ __launch_bounds__(256) void stage_kernel(StageTable t) {
  const cgvp_stage_item it = t.it[blockIdx.y];
  const int64_t words = it.capacity_bytes >> 2, copy = it.copy_bytes >> 2;
  uint32_t* __restrict__ d = static_cast<uint32_t*>(it.dst);
  const uint32_t* __restrict__ s = static_cast<const uint32_t*>(it.src);
  const bool vec = !(((uintptr_t)d | (uintptr_t)s) & 15);
  if (vec) {                                           // 16-byte body, word tail
    const int64_t c4 = copy >> 2, w4 = words >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < w4; i += (int64_t)gridDim.x * 256) {
      uint4 v = make_uint4(it.fill_word, it.fill_word, it.fill_word, it.fill_word);
      if (i < c4) v = reinterpret_cast<const uint4*>(s)[i];
      else if (i == c4 && (copy & 3)) {
        const int64_t b = i << 2;
        v.x = b < copy ? s[b] : it.fill_word; v.y = b + 1 < copy ? s[b + 1] : it.fill_word;
        v.z = b + 2 < copy ? s[b + 2] : it.fill_word; v.w = it.fill_word;
      }
      reinterpret_cast<uint4*>(d)[i] = v;
    }
    for (int64_t i = (w4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256)
      d[i] = i < copy ? s[i] : it.fill_word;
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256)
    d[i] = i < copy ? s[i] : it.fill_word;
}

}  // namespace

extern "C" {

int cgvp_stage_buffers(const cgvp_stage_item* items, int32_t n, void* stream) {
  if (n < 0 || n > CGVP_MAX_STAGE || (n > 0 && !items)) return CGVP_ERR_BAD_ARG;
  if (n == 0) return 0;
  StageTable t;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    const cgvp_stage_item& it = items[i];
    if (it.copy_bytes < 0 || it.capacity_bytes < it.copy_bytes || (it.copy_bytes & 3) || (it.capacity_bytes & 3)) return CGVP_ERR_BAD_ARG;
    if ((it.capacity_bytes > 0 && !it.dst) || (it.copy_bytes > 0 && !it.src)) return CGVP_ERR_BAD_ARG;
    if (((uintptr_t)it.dst & 3) || ((uintptr_t)it.src & 3)) return CGVP_ERR_BAD_ARG;
    t.it[i] = it;
    most = it.capacity_bytes > most ? it.capacity_bytes : most;
  }
  int64_t blocks = (most / 16 + 255) / 256;            // one uint4 per thread on the largest item; smaller items finish early
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(stage_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
  return launch_status();
}

// ===================================================================================== protein encoder
int cgvp_lba_fwd_workspace(const cgvp_dims* dims, const cgvp_layout* layout, int64_t N, int64_t E, int32_t save_state,
                           cgvp_lba_fwd_ws* out) {
  if (!dims || !layout || !out || N < 0 || E < 0) return CGVP_ERR_BAD_ARG;
  if (dims->layer_kind != CGVP_LAYER_GATED) return CGVP_ERR_UNSUPPORTED_DIMS;      // whole passes: CASTER-DTA's encoder only
  const int64_t img = cgvp_lba_image_floats(dims, layout);
  if (img < 0) return (int)img;
  const int nc = num_convs_of(*layout);
  Bump b;
  out->seed = b.take(16);
  out->image = b.take(img * 4);
  out->state_rows = save_state ? 2 * nc + 1 : 3;
  out->node_stride = N + (N & 1);                     // even row count: every slice stays 16-B aligned in bf16 too
  out->state = b.take(out->state_rows * out->node_stride * ROW * esize(dims));
  // the CSR build's scratch (edge ids in arrival order, E ints) lives in the edge-embedding store: it is consumed by the
  // ranking launch before conv layer 0 writes the store
  out->e_emb = b.take((E + 1) * EROW * esize(dims) > E * 4 ? (E + 1) * EROW * esize(dims) : E * 4);
  out->ids_scratch = out->e_emb;
  out->rowptr = b.take((N + 1) * 4);
  out->eperm = b.take((E > 0 ? E : 1) * 4);
  out->esrc = b.take((E > 0 ? E : 1) * 4);
  out->edst = b.take((E > 0 ? E : 1) * 4);
  out->total = b.off;
  return 0;
}

int cgvp_lba_forward_plan(int64_t N, int64_t E, int32_t num_convs, int32_t prebuilt_csr, int32_t flags, int32_t* fused_layers) {
  if (N < 0 || E < 0 || num_convs < 1) return CGVP_ERR_BAD_ARG;
  const int fuse = !(flags & CGVP_PASS_UNFUSED) && E <= 4 * N;
  if (fused_layers) *fused_layers = fuse;
  int launches = 1 + (prebuilt_csr ? 0 : (E > 0 ? 3 : 2));     // pass_begin; CSR scan (+ fill) + rank
  if (N > 0) launches += num_convs * (fuse ? 1 : 2);
  return launches;
}

int cgvp_lba_forward_pass(const cgvp_dims* dims, const cgvp_layout* layout, const float* params,
                          const cgvp_lba_batch* batch, int32_t aggr_mean, float dropout_p, uint64_t* rng_state,
                          const float* masks, int32_t* csr_counters, void* workspace, int32_t save_state,
                          int32_t flags, float* out, void* stream) {
  if (!dims || !layout || !params || !batch || !workspace) return CGVP_ERR_BAD_ARG;
  if ((uintptr_t)workspace & 255) return CGVP_ERR_BAD_ARG;
  const int64_t N = batch->num_nodes, E = batch->num_edges;
  cgvp_lba_fwd_ws ws;
  if (int rc = cgvp_lba_fwd_workspace(dims, layout, N, E, save_state, &ws)) return rc;
  const int nc = num_convs_of(*layout);
  if (nc < 1) return CGVP_ERR_UNSUPPORTED_DIMS;          // the output head is fused into the last layer
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return CGVP_ERR_BAD_ARG;
  const bool drop = dropout_p > 0.f && save_state;
  const bool draw = drop && !masks;
  if (draw && !rng_state) return CGVP_ERR_BAD_ARG;
  const bool tables = has_tables(batch);
  if (tables && (E > 0 && (!batch->eperm || !batch->esrc || !batch->edst))) return CGVP_ERR_BAD_ARG;
  if (!tables && E > 0 && (!batch->edge_index || !csr_counters)) return CGVP_ERR_BAD_ARG;
  if (N > 0 && !out) return CGVP_ERR_BAD_ARG;
  const int es = esize(dims);
  float* image = at<float>(workspace, ws.image);
  uint64_t* seed = at<uint64_t>(workspace, ws.seed);
  char* state = at<char>(workspace, ws.state);
  const int64_t srow = ws.node_stride * ROW * es;        // bytes per state row
  auto row = [&](int64_t r) { return reinterpret_cast<float*>(state + r * srow); };
  float* e_emb = at<float>(workspace, ws.e_emb);
  int32_t* rowptr = tables ? const_cast<int32_t*>(batch->rowptr) : at<int32_t>(workspace, ws.rowptr);
  int32_t* eperm = tables ? const_cast<int32_t*>(batch->eperm) : at<int32_t>(workspace, ws.eperm);
  int32_t* esrc = tables ? const_cast<int32_t*>(batch->esrc) : at<int32_t>(workspace, ws.esrc);
  int32_t* edst = tables ? const_cast<int32_t*>(batch->edst) : at<int32_t>(workspace, ws.edst);
  const bool count_here = !tables && E > 0;
  // (1) node embedding + fragment image + per-target edge counts, one launch
  if (int rc = cgvp_lba_pass_begin(dims, layout, params, image, batch->x_s, batch->x_v, batch->ntypes, N, row(0),
                                   draw ? rng_state : nullptr, draw ? seed : nullptr,
                                   count_here ? batch->edge_index : nullptr, E, count_here ? csr_counters : nullptr, stream))
    return rc;
  // (2) CSR tables (three launches after the fused count; none when the batch brings collated tables)
  if (!tables) {
    if (!csr_counters) return CGVP_ERR_BAD_ARG;
    if (int rc = cgvp_csr_from_coo(batch->edge_index, N, E, rowptr, eperm, esrc, edst, csr_counters, count_here ? 2 : 1,
                                   at<int32_t>(workspace, ws.ids_scratch), stream))
      return rc;
  }
  if (N == 0) return 0;
  // (3) the conv layers
  int32_t fuse_i = 0;
  (void)cgvp_lba_forward_plan(N, E, nc, tables ? 1 : 0, flags, &fuse_i);
  const bool fuse = fuse_i != 0;
  const int64_t mstride = N * MROW;
  for (int l = 0; l < nc; ++l) {
    const bool last = l == nc - 1;
    cgvp_rng rng{draw ? seed : nullptr, dropout_p, 2 * l};
    const cgvp_rng* rp = draw ? &rng : nullptr;
    const float* m0 = (drop && masks) ? masks + (int64_t)(2 * l) * mstride : nullptr;
    const float* m1 = (drop && masks) ? masks + (int64_t)(2 * l + 1) * mstride : nullptr;
    float *h, *dh, *h_out;
    if (save_state) { h = row(l); dh = row(nc + l); h_out = last ? row(2 * nc) : row(l + 1); }
    else { h = row(l & 1); dh = fuse ? nullptr : row(2); h_out = last ? nullptr : row((l + 1) & 1); }
    const float* e_in = l > 0 ? e_emb : nullptr;
    float* e_out = l == 0 ? e_emb : nullptr;
    if (fuse) {
      Timed timed(0, stream);
      if (int rc = cgvp_conv_layer_fwd(dims, layout, image, l, h, batch->e_s, batch->e_v, batch->etypes, rowptr, eperm, esrc,
                                       edst, N, E, aggr_mean, m0, m1, rp, last ? 1 : 0, e_in, e_out, dh, h_out, out, stream))
        return rc;
    } else {
      {
        Timed timed(0, stream);
        if (int rc = cgvp_conv_fwd(dims, layout, params, image, l, h, batch->e_s, batch->e_v, batch->etypes, rowptr, eperm,
                                   esrc, edst, N, E, aggr_mean, e_in, e_out, dh, stream))
          return rc;
      }
      if (!save_state && last) h_out = row((l + 1) & 1);      // the unfused head kernel takes an h_out buffer either way
      if (int rc = cgvp_node_update_fwd_train(dims, layout, image, l, h, dh, m0, m1, rp, N, last ? 1 : 0, h_out, out, stream))
        return rc;
    }
  }
  return launch_status();
}

int64_t cgvp_lba_bwd_workspace_bytes(const cgvp_dims* dims, const cgvp_layout* layout, int64_t N, int64_t E) {
  if (!dims || !layout || N < 0 || E < 0) return CGVP_ERR_BAD_ARG;
  if (dims->layer_kind != CGVP_LAYER_GATED) return CGVP_ERR_UNSUPPORTED_DIMS;
  LbaBwdWs w;
  if (int rc = lba_bwd_layout(dims, layout, N, E, &w)) return rc;
  return w.total;
}

int cgvp_lba_backward_pass(const cgvp_dims* dims, const cgvp_layout* layout, const cgvp_lba_batch* batch,
                           int32_t aggr_mean, float dropout_p, const float* masks, const void* fwd_workspace,
                           const float* g_out, void* bwd_workspace, float* grad_params, float* g_x_s, float* g_x_v,
                           float* g_e_s, float* g_e_v, void* stream) {
  if (!dims || !layout || !batch || !fwd_workspace || !bwd_workspace || !grad_params) return CGVP_ERR_BAD_ARG;
  if (((uintptr_t)fwd_workspace & 255) || ((uintptr_t)bwd_workspace & 255)) return CGVP_ERR_BAD_ARG;
  if ((g_x_s == nullptr) != (g_x_v == nullptr) || (g_e_s == nullptr) != (g_e_v == nullptr)) return CGVP_ERR_BAD_ARG;
  const int64_t N = batch->num_nodes, E = batch->num_edges;
  cgvp_lba_fwd_ws ws;
  if (int rc = cgvp_lba_fwd_workspace(dims, layout, N, E, 1, &ws)) return rc;
  LbaBwdWs bw;
  if (int rc = lba_bwd_layout(dims, layout, N, E, &bw)) return rc;
  const int nc = num_convs_of(*layout);
  if (nc < 1) return CGVP_ERR_UNSUPPORTED_DIMS;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {                                          // nothing contributes: the gradient arena is all zeros
    quad::zero_words(grad_params, (size_t)layout->total, st);
    return launch_status();
  }
  if (!g_out) return CGVP_ERR_BAD_ARG;
  const bool drop = dropout_p > 0.f;
  const bool draw = drop && !masks;
  const int es = esize(dims);
  const float* image = at<float>(fwd_workspace, ws.image);
  const uint64_t* seed = at<uint64_t>(fwd_workspace, ws.seed);
  const char* state = at<char>(fwd_workspace, ws.state);
  const int64_t srow = ws.node_stride * ROW * es;
  auto row = [&](int64_t r) { return reinterpret_cast<const float*>(state + r * srow); };
  const float* e_emb = at<float>(fwd_workspace, ws.e_emb);
  const bool tables = has_tables(batch);
  const int32_t* rowptr = tables ? batch->rowptr : at<int32_t>(fwd_workspace, ws.rowptr);
  const int32_t* eperm = tables ? batch->eperm : at<int32_t>(fwd_workspace, ws.eperm);
  const int32_t* esrc = tables ? batch->esrc : at<int32_t>(fwd_workspace, ws.esrc);
  const int32_t* edst = tables ? batch->edst : at<int32_t>(fwd_workspace, ws.edst);
  float* slabs = at<float>(bwd_workspace, bw.slabs);
  float* g_e = at<float>(bwd_workspace, bw.g_e);
  float* rows = at<float>(bwd_workspace, bw.rows);
  // node-row gradient buffers: two sets (even / odd layer) of {g_dh, g_h, g_src, g_dst}; a layer reads the set of the
  // layer above it while it fills its own
  auto buf = [&](int l, int k) { return rows + (int64_t)((l & 1) * 4 + k) * bw.row_floats; };
  cgvp_segment segs[CGVP_MAX_SEGS];
  int nseg = 0, stage = 0;
  int32_t cnt = 0;
  auto region = [&]() { return slabs + (int64_t)(stage++) * bw.slab_floats; };
  if (2 * nc + 2 + 1 > CGVP_MAX_SEGS) return CGVP_ERR_UNSUPPORTED_DIMS;
  const int64_t mstride = N * MROW;
  const float *up0 = nullptr, *up1 = nullptr, *up2 = nullptr;
  for (int l = nc - 1; l >= 0; --l) {
    const bool last = l == nc - 1;
    float* g_dh = buf(l, 0);
    float* g_h = drop ? buf(l, 1) : nullptr;
    float* g_src = buf(l, 2);
    float* g_dst = buf(l, 3);
    cgvp_rng rng{draw ? seed : nullptr, dropout_p, 2 * l};
    const float* m0 = (drop && masks) ? masks + (int64_t)(2 * l) * mstride : nullptr;
    const float* m1 = (drop && masks) ? masks + (int64_t)(2 * l + 1) * mstride : nullptr;
    if (int rc = cgvp_node_update_bwd(dims, layout, image, l, row(l), row(nc + l), m0, m1, draw ? &rng : nullptr,
                                      last ? row(2 * nc) : nullptr, last ? g_out : nullptr, up0, up1, up2, N, last ? 1 : 0,
                                      g_dh, g_h, g_src, grad_params, region(), segs + nseg, &cnt, stream))
      return rc;
    nseg += cnt;
    {
      Timed timed(1, stream);
      if (int rc = cgvp_conv_bwd(dims, layout, image, l, row(l), e_emb, rowptr, esrc, edst, N, E, aggr_mean, g_dh, g_src, 1,
                                 g_dst, g_e + (int64_t)l * bw.g_e_stride, grad_params, region(), segs + nseg, &cnt, stream))
        return rc;
    }
    nseg += cnt;
    up0 = g_h ? g_h : g_dh; up1 = g_src; up2 = g_dst;
  }
  if (int rc = cgvp_node_embed_bwd(dims, layout, image, batch->x_s, batch->x_v, batch->ntypes, N, up0, up1, up2, g_x_s, g_x_v,
                                   grad_params, region(), segs + nseg, &cnt, stream))
    return rc;
  nseg += cnt;
  if (E > 0) {
    const float* ge[16];
    if (nc > 16) return CGVP_ERR_UNSUPPORTED_DIMS;
    for (int l = 0; l < nc; ++l) ge[l] = g_e + (int64_t)l * bw.g_e_stride;
    if (int rc = cgvp_edge_embed_bwd(dims, layout, image, batch->e_s, batch->e_v, batch->etypes, eperm, E, ge, nc,
                                     g_e_s, g_e_v, grad_params, region(), segs + nseg, &cnt, stream))
      return rc;
    nseg += cnt;
  } else {                                               // no edge stage: gvp_edge's gradient block is covered by no segment
    quad::zero_words(grad_params + layout->edge_gvp, (size_t)(layout->conv0 - layout->edge_gvp), st);
  }
  if (int rc = cgvp_bwd_reduce(segs, nseg, grad_params, 1, stream)) return rc;
  return launch_status();
}

// ===================================================================================== drug encoder
int cgvp_gine_fwd_workspace(const cgvp_gine_cfg* cfg, int64_t N, int64_t E, int32_t save_state, cgvp_gine_fwd_ws* out) {
  if (int rc = check_gine_cfg(cfg)) return rc;
  if (!out || N < 0 || E < 0) return CGVP_ERR_BAD_ARG;
  const int L = cfg->num_layers;
  Bump b;
  out->seed = b.take(16);
  for (int l = 0; l < CGVP_GINE_MAX_LAYERS; ++l) out->hidden[l] = out->agg[l] = out->pos[l] = 0;
  if (save_state) {
    for (int l = 0; l < L - 1; ++l) out->hidden[l] = b.take(N * cfg->widths[l + 1] * 4);
    for (int l = 0; l < L; ++l) {
      out->agg[l] = b.take(N * ((cfg->widths[l] + 15) / 16 * 16) * 4);
      out->pos[l] = b.take((E > 0 ? E : 1) * 8);
    }
  } else {                                              // inference: two ping-pong buffers of the widest hidden layer
    int maxw = 1;
    for (int l = 1; l < L; ++l) maxw = cfg->widths[l] > maxw ? cfg->widths[l] : maxw;
    const int64_t a = b.take(N * maxw * 4), c = b.take(N * maxw * 4);
    for (int l = 0; l < L - 1; ++l) out->hidden[l] = (l & 1) ? c : a;
  }
  out->rowptr = b.take((N + 1) * 4);
  out->eperm = b.take((E > 0 ? E : 1) * 4);
  out->esrc = b.take((E > 0 ? E : 1) * 4);
  out->edst = b.take((E > 0 ? E : 1) * 4);
  out->saved = b.off;                                    // what the backward reads ends here
  out->ids_scratch = b.take((E > 0 ? E : 1) * 4);        // CSR build scratch: edge ids in (non-deterministic) arrival order
  out->total = b.off;
  return 0;
}

int cgvp_gine_forward_pass(const cgvp_gine_cfg* cfg, const cgvp_gine_w* w, const cgvp_gine_batch* batch, float dropout_p,
                           uint64_t* rng_state, const float* const* masks, int32_t* csr_counters, void* workspace,
                           int32_t save_state, int32_t variant, float* out, void* stream) {
  if (!cfg || !w || !batch || !workspace || ((uintptr_t)workspace & 255)) return CGVP_ERR_BAD_ARG;
  const int64_t N = batch->num_nodes, E = batch->num_edges;
  cgvp_gine_fwd_ws ws;
  if (int rc = cgvp_gine_fwd_workspace(cfg, N, E, save_state, &ws)) return rc;
  if (!(dropout_p >= 0.f && dropout_p < 1.f)) return CGVP_ERR_BAD_ARG;
  const int L = cfg->num_layers;
  const bool drop = dropout_p > 0.f && save_state && L > 1;
  const bool draw = drop && !masks;
  if (draw && !rng_state) return CGVP_ERR_BAD_ARG;
  const bool tables = has_tables(batch);
  if (!tables && (!csr_counters || (E > 0 && !batch->edge_index))) return CGVP_ERR_BAD_ARG;
  if (N > 0 && !out) return CGVP_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  uint64_t* seed = at<uint64_t>(workspace, ws.seed);
  const int32_t* rowptr = tables ? batch->rowptr : at<int32_t>(workspace, ws.rowptr);
  const int32_t* eperm = tables ? batch->eperm : at<int32_t>(workspace, ws.eperm);
  const int32_t* esrc = tables ? batch->esrc : at<int32_t>(workspace, ws.esrc);
  const int32_t* edst = tables ? batch->edst : at<int32_t>(workspace, ws.edst);
  if (!tables) {
    // the count launch also advances the dropout generator (no launch of its own)
    if (int rc = quad::csr_build(batch->edge_index, N, E, at<int32_t>(workspace, ws.rowptr), at<int32_t>(workspace, ws.eperm),
                                 at<int32_t>(workspace, ws.esrc), at<int32_t>(workspace, ws.edst), csr_counters, 1,
                                 at<int32_t>(workspace, ws.ids_scratch),
                                 draw ? reinterpret_cast<unsigned long long*>(rng_state) : nullptr,
                                 draw ? reinterpret_cast<unsigned long long*>(seed) : nullptr, st))
      return rc;
  } else if (draw) {
    if (int rc = cgvp_rng_next(rng_state, seed, stream)) return rc;
  }
  if (N == 0) return 0;
  const float* x = batch->x;
  for (int l = 0; l < L; ++l) {
    const bool first = l == 0, lastl = l == L - 1;
    float* y = lastl ? out : at<float>(workspace, ws.hidden[l]);
    cgvp_rng rng{draw ? seed : nullptr, dropout_p, l};
    const bool dl = drop && !lastl;
    bool done = false;
    if (save_state) {
      // training pass: the tile kernels (whatever `variant` says), which also SAVE the aggregated messages and their ReLU
      // patterns for the backward (cgvp_gine_fwd_ws.agg / .pos); a width they are not compiled for takes the generic
      // kernel below -- and has no backward kernel either
      if (((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return CGVP_ERR_BAD_ARG;
      gvp::RngArgs ra{(dl && draw) ? reinterpret_cast<const unsigned long long*>(seed) : nullptr, dropout_p, l};
      if (!(dl && draw)) ra = gvp::RngArgs{nullptr, 0.f, 0};
      const int rc = quad::gine_fwd(cfg->widths[l], cfg->widths[l + 1], cfg->widths[l + 1], first ? cfg->num_ntypes : 0,
                                    cfg->num_etypes, cfg->edge_dim, x, first ? batch->ntypes : nullptr, batch->eattr,
                                    batch->etypes, rowptr, eperm, esrc, edst, N, w + l, cfg->act_slope,
                                    (dl && masks) ? masks[l] : nullptr, ra, y,
                                    gine_recompute() ? nullptr : at<float>(workspace, ws.agg[l]),
                                    gine_recompute() ? nullptr : at<uint16_t>(workspace, ws.pos[l]), st);
      if (rc < 0 || rc > 1) return rc;
      done = rc == 0;
    }
    if (!done)
      if (int rc = cgvp_gine_conv_fwd(x, first ? batch->ntypes : nullptr, first ? cfg->num_ntypes : 0, batch->eattr, batch->etypes,
                                      cfg->num_etypes, cfg->edge_dim, rowptr, eperm, esrc, edst, N, E, cfg->widths[l],
                                      cfg->widths[l + 1], cfg->widths[l + 1], w + l, cfg->act_slope,
                                      (dl && masks) ? masks[l] : nullptr, (dl && draw) ? &rng : nullptr, variant, y, stream))
        return rc;
    x = y;
  }
  return launch_status();
}

int64_t cgvp_gine_bwd_workspace_bytes(const cgvp_gine_cfg* cfg, int64_t N, int64_t E) {
  if (int rc = check_gine_cfg(cfg)) return rc;
  if (N < 0 || E < 0) return CGVP_ERR_BAD_ARG;
  GineBwdWs w;
  gine_bwd_layout(cfg, N, &w);
  return w.total;
}

int cgvp_gine_backward_pass(const cgvp_gine_cfg* cfg, const cgvp_gine_w* w, const cgvp_gine_batch* batch, float dropout_p,
                            const float* const* masks, const void* fwd_workspace, const float* g_out, void* bwd_workspace,
                            float* grad_flat, float* g_x, int32_t max_workgroups, void* stream) {
  if (!cfg || !w || !batch || !fwd_workspace || !bwd_workspace || !grad_flat) return CGVP_ERR_BAD_ARG;
  if (((uintptr_t)fwd_workspace & 255) || ((uintptr_t)bwd_workspace & 255)) return CGVP_ERR_BAD_ARG;
  const int64_t N = batch->num_nodes, E = batch->num_edges;
  cgvp_gine_fwd_ws ws;
  if (int rc = cgvp_gine_fwd_workspace(cfg, N, E, 1, &ws)) return rc;
  GineBwdWs bw;
  gine_bwd_layout(cfg, N, &bw);
  const int L = cfg->num_layers;
  const int ke = cfg->num_etypes + cfg->edge_dim;
  hipStream_t st = (hipStream_t)stream;
  int64_t total = 0;
  int loff[CGVP_GINE_MAX_LAYERS];
  for (int l = 0; l < L; ++l) { loff[l] = (int)total; total += gine_layer_floats(cfg->widths[l], cfg->widths[l + 1], cfg->widths[l + 1], ke); }
  if (N == 0) {
    quad::zero_words(grad_flat, (size_t)total, st);
    return launch_status();
  }
  if (!g_out) return CGVP_ERR_BAD_ARG;
  const bool drop = dropout_p > 0.f && L > 1;
  const bool draw = drop && !masks;
  const uint64_t* seed = at<uint64_t>(fwd_workspace, ws.seed);
  const bool tables = has_tables(batch);
  const int32_t* rowptr = tables ? batch->rowptr : at<int32_t>(fwd_workspace, ws.rowptr);
  const int32_t* eperm = tables ? batch->eperm : at<int32_t>(fwd_workspace, ws.eperm);
  const int32_t* esrc = tables ? batch->esrc : at<int32_t>(fwd_workspace, ws.esrc);
  const int32_t* edst = tables ? batch->edst : at<int32_t>(fwd_workspace, ws.edst);
  float* slabs = at<float>(bwd_workspace, bw.slabs);
  float* gx = at<float>(bwd_workspace, bw.gx);
  // every d x buffer the layer kernels fill with float atomics, zeroed by ONE launch (the caller's g_x by a second)
  if (L > 1) quad::zero_words(gx, (size_t)(L - 1) * bw.gx_floats, st);
  if (g_x) quad::zero_words(g_x, (size_t)N * (cfg->widths[0] - cfg->num_ntypes), st);
  cgvp_segment segs[CGVP_GINE_MAX_LAYERS];
  const float* g = g_out;
  for (int l = L - 1; l >= 0; --l) {
    const bool first = l == 0;
    const float* x = first ? batch->x : at<float>(fwd_workspace, ws.hidden[l - 1]);
    float* gxl = first ? g_x : gx + (int64_t)(l - 1) * bw.gx_floats;
    const bool dl = drop && l < L - 1;
    gvp::RngArgs ra{(dl && draw) ? reinterpret_cast<const unsigned long long*>(seed) : nullptr, dropout_p, l};
    if (!(dl && draw)) ra = gvp::RngArgs{nullptr, 0.f, 0};
    const float* mask = (dl && masks) ? masks[l] : nullptr;
    if (((uintptr_t)g & 15) || ((uintptr_t)mask & 15)) return CGVP_ERR_BAD_ARG;
    int rows = 0, row_len = 0;
    float* slab = slabs + (int64_t)l * bw.slab_floats;
    if (int rc = quad::gine_bwd(cfg->widths[l], cfg->widths[l + 1], cfg->widths[l + 1], first ? cfg->num_ntypes : 0,
                                cfg->num_etypes, cfg->edge_dim, x, first ? batch->ntypes : nullptr, batch->eattr, batch->etypes,
                                rowptr, eperm, esrc, edst, N, w + l, cfg->act_slope, mask, ra, g, gxl, slab, max_workgroups,
                                &rows, &row_len, st, gine_recompute() ? nullptr : at<float>(fwd_workspace, ws.agg[l]),
                                gine_recompute() ? nullptr : at<uint16_t>(fwd_workspace, ws.pos[l])))
      return rc;
    segs[l] = cgvp_segment{slab, rows, row_len, 0,
                           gine_layer_floats(cfg->widths[l], cfg->widths[l + 1], cfg->widths[l + 1], ke), loff[l]};
    g = gxl;
  }
  // ONE reduce for all layers; the segments are disjoint and cover grad_flat: stored, no zero fill
  quad::reduce_segments(segs, L, grad_flat, st, 1);
  return launch_status();
}

// ===================================================================================== diagnostics
int cgvp_debug_kernel_timing(int32_t enable) {
  g_timing.store(enable != 0, std::memory_order_relaxed);
  return 0;
}

int cgvp_debug_kernel_times(float* ms, int32_t* kinds, int32_t capacity) {
  int n = 0;
  std::vector<TimedLaunch> taken;
  {
    std::lock_guard<std::mutex> lock(g_timed_mutex);
    taken.swap(g_timed);
  }
  for (const TimedLaunch& t : taken) {
    float v = 0.f;
    (void)hipEventSynchronize(t.b);
    (void)hipEventElapsedTime(&v, t.a, t.b);
    if (n < capacity && ms && kinds) { ms[n] = v; kinds[n] = t.kind; }
    ++n;
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  return n;
}

}  // extern "C"
