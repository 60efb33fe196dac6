// torch_bridge.cpp -- the EAGER fast path between PyTorch and libcaster_gvp.so (host code, compiled with g++).
//
// Why it exists: the reference's training loop (train_model.py:548-587) calls the model eagerly with a NEW batch every
// step; the hot path's device time is ~0.26 ms per 64-pair step, and driving it through Python -- a torch.library custom
// op with a 60-tensor parameter list, ctypes marshalling, a Python backward that hands 60 gradient views back to the
// engine -- costs 0.7 ms of host time per encoder pass pair (measured, tools/host_profile_encoders.py).  Here the same
// two things happen in C++:
//   * ONE call per pass into the C ABI (cgvp_*_forward_pass / cgvp_*_backward_pass, include/caster_gvp.h), workspaces
//     from PyTorch's caching allocator on PyTorch's current stream;
//   * the autograd node of the pass (torch::autograd::Node): its apply() runs the backward pass and returns the
//     parameter gradients as views of ONE gradient arena.
// The torch.library custom ops (gvp_hip/autograd_ops.py) stay the path torch.compile traces; both call the same C
// entry points, so the launch sequence lives in one place.  No arithmetic happens here.
#include <torch/extension.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/functions/accumulate_grad.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <torch/csrc/autograd/graph_task.h>
#include <torch/csrc/autograd/variable.h>

#include <c10/hip/HIPStream.h>

#include <chrono>
#include <cstdlib>
#include <map>

#include "../../include/caster_gvp.h"

namespace {

using torch::autograd::Node;
using torch::autograd::variable_list;

// opt-in host timing (CGVP_BRIDGE_TIMING=1): wall time per named section, reported by timing_report()
struct Section { double total = 0; long n = 0; };
std::map<std::string, Section>& sections() { static std::map<std::string, Section> m; return m; }
const bool g_time = std::getenv("CGVP_BRIDGE_TIMING") != nullptr;
struct Tic {
  const char* name; std::chrono::steady_clock::time_point t0;
  explicit Tic(const char* n) : name(n) { if (g_time) t0 = std::chrono::steady_clock::now(); }
  ~Tic() {
    if (!g_time) return;
    Section& s = sections()[name];
    s.total += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    ++s.n;
  }
};

struct NotImplemented : public std::runtime_error { using std::runtime_error::runtime_error; };     // -> Python NotImplementedError

void check(int rc, const char* what) {
  if (rc == 0) return;
  if (rc == CGVP_ERR_UNSUPPORTED_DIMS)
    throw NotImplemented(std::string(what) + ": dimensions outside the compiled CASTER-DTA(s,v) configuration");
  if (rc < 0) throw py::value_error(std::string(what) + ": bad argument (code " + std::to_string(rc) + ")");
  throw std::runtime_error(std::string(what) + ": HIP launch failed with hipError_t " + std::to_string(rc));
}

void* current_stream(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// ------------------------------------------------------------------ leaf gradients without one engine task per leaf
// The encoders have 50 + 24 parameter leaves; the autograd engine spends ~3-4 us of host time per leaf (a queue hand-off
// and an AccumulateGrad task each) whoever produces the gradients: ~0.3 ms per step for 0.25 ms of device work.  When a
// backward pass is a plain `loss.backward()` -- ACCUMULATE mode (no `inputs=`, not torch.autograd.grad) -- and no leaf
// carries a hook, a pass's node does what the 60 AccumulateGrad tasks would do, itself: p.grad = view of the gradient
// arena (or p.grad += view when a gradient is already there), and takes its edges to those AccumulateGrad nodes out of
// THIS backward (the engine skips invalid edges); an end-of-backward callback puts them back, so the graph is what it was:
// torch.autograd.grad, backward(inputs=...), hooks (DDP, register_hook, post-accumulate hooks) and retain_graph see the
// stock structure and take the stock path.  CGVP_EXACT_LEAVES=1 (or set_exact_leaves(True)) forces the stock path.
bool g_exact_leaves = std::getenv("CGVP_EXACT_LEAVES") != nullptr && std::getenv("CGVP_EXACT_LEAVES")[0] == '1';
long g_fast_leaf_passes = 0;                // diagnostics (tests): passes that took the fast path

struct LeafScatter {
  // `views[i]`: the gradient of parameter i (a view of the pass's gradient arena `garena`, arena order), for i < np.
  // Returns true when the leaves' .grad hold the result: the caller then returns UNDEFINED outputs for [0, np).
  static bool run(Node& node, size_t np, const std::vector<at::Tensor>& views, const at::Tensor& garena) {
    if (g_exact_leaves || at::GradMode::is_enabled()) return false;
    const auto* exec_info = torch::autograd::get_current_graph_task_exec_info();
    if (exec_info && !exec_info->empty()) return false;          // autograd.grad / backward(inputs=...): captures needed
    std::vector<torch::autograd::AccumulateGrad*> acc(np, nullptr);
    for (size_t i = 0; i < np; ++i) {
      const torch::autograd::Edge& e = node.next_edge(i);
      if (!e.is_valid()) continue;                                // a frozen parameter
      auto* a = dynamic_cast<torch::autograd::AccumulateGrad*>(e.function.get());
      if (!a || e.input_nr != 0) return false;                    // not a leaf (a view / clone of one): stock path
      if (!a->pre_hooks().empty() || !a->post_hooks().empty() || !a->tensor_pre_hooks().empty() ||
          !a->retains_grad_hooks().empty() || a->tensor_post_acc_grad_hooks() != nullptr)
        return false;                                             // somebody listens at this leaf: stock path
      if (!views[i].defined() || a->variable.sizes() != views[i].sizes()) return false;
      acc[i] = a;
    }
    // one fused add when every leaf already holds a gradient and those are consecutive views of one arena themselves
    bool all_defined = true, none_defined = true;
    for (size_t i = 0; i < np; ++i) {
      if (!acc[i]) continue;
      const bool d = acc[i]->variable.grad().defined();
      all_defined = all_defined && d;
      none_defined = none_defined && !d;
    }
    bool fused = false;
    if (all_defined && !none_defined) {
      const char* at_ = nullptr;
      bool ok = true;
      int64_t total = 0;
      for (size_t i = 0; i < np && ok; ++i) {
        if (!acc[i]) { ok = false; break; }
        const at::Tensor& g = acc[i]->variable.grad();
        ok = g.scalar_type() == at::kFloat && g.is_contiguous() && g.device() == garena.device() &&
             (at_ == nullptr || static_cast<const char*>(g.data_ptr()) == at_);
        at_ = static_cast<const char*>(g.data_ptr()) + g.numel() * 4;
        total += g.numel();
      }
      if (ok && total == garena.numel()) {
        at::Tensor old = at::from_blob(acc[0]->variable.grad().data_ptr(), {total}, garena.options());
        old.add_(garena);
        fused = true;
      }
    }
    if (!fused) {
      for (size_t i = 0; i < np; ++i) {
        if (!acc[i]) continue;
        at::Tensor& g = acc[i]->variable.mutable_grad();
        if (!g.defined()) g = views[i];
        else g.add_(views[i]);                                    // (in place, as AccumulateGrad does outside create_graph)
      }
    }
    // Take the leaf edges out of this backward, and put them back when it ends.  Also what the engine does for leaf
    // streams: the caller's stream waits for the stream this node ran on, if that is another one.
    auto self = node.getptr();
    auto saved = std::make_shared<torch::autograd::edge_list>(node.next_edges().begin(), node.next_edges().begin() + np);
    for (size_t i = 0; i < np; ++i) node.next_edges()[i] = torch::autograd::Edge();
    const c10::hip::HIPStream ran_on = c10::hip::getCurrentHIPStream(garena.device().index());
    torch::autograd::Engine::get_default_engine().queue_callback([self, saved, np, ran_on]() {
      for (size_t i = 0; i < np; ++i) self->next_edges()[i] = (*saved)[i];
      // (the callback runs on the thread that called backward(), with the caller's streams current)
      const c10::hip::HIPStream now = c10::hip::getCurrentHIPStream(ran_on.device_index());
      if (now != ran_on) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
          (void)hipEventRecord(ev, ran_on.stream());
          (void)hipStreamWaitEvent(now.stream(), ev, 0);
          (void)hipEventDestroy(ev);
        }
      }
    });
    ++g_fast_leaf_passes;
    return true;
  }
};

// an activation / index buffer as the kernels want it: on the GPU, contiguous, 16-B aligned
// What a backward node re-reads from LIVE storage (inputs; for the drug encoder also the weights -- the protein pass
// snapshots its weights into the forward workspace) must not have been written in place since the forward: stock autograd
// raises through SavedVariable's version check, these nodes hold plain tensors, so they keep the version counters themselves.
struct Guarded { at::Tensor t; int64_t version; const char* what; };
inline void guard_add(std::vector<Guarded>& g, const at::Tensor& t, const char* what) {
  if (t.defined() && !t.is_inference()) g.push_back({t, t._version(), what});
}
inline void guard_check(const std::vector<Guarded>& g, const char* node) {
  for (const Guarded& e : g)
    TORCH_CHECK(e.t._version() == e.version, "caster_gvp: ", e.what, " needed by ", node, " was modified by an in-place "
                "operation after the forward pass (version ", e.version, " -> ", e.t._version(), "): the gradients would be "
                "computed from the new values");
}
// The kernels' backward is not itself differentiable: create_graph=True must fail loudly instead of returning constants.
inline void once_differentiable(const char* node) {
  TORCH_CHECK(!at::GradMode::is_enabled(), "caster_gvp: ", node, " is not differentiable twice (backward called with "
              "create_graph=True); the encoder kernels provide first derivatives only");
}

at::Tensor ready(const at::Tensor& t, const char* name, c10::optional<at::ScalarType> dtype = c10::nullopt) {
  TORCH_CHECK(t.is_cuda(), name, ": caster-dta_amd runs on MI355X only (got a ", t.device(), " tensor); there is no CPU path");
  at::Tensor r = t;
  if (dtype && r.scalar_type() != *dtype) r = r.to(*dtype);
  if (!r.is_contiguous()) r = r.contiguous();
  if (reinterpret_cast<uintptr_t>(r.data_ptr()) % 16) r = r.clone();
  return r;
}
void* ptr(const at::Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr() : nullptr; }

// A contiguous [sizes] view of the flat gradient arena at element `offset`, built directly (a TensorImpl on the arena's
// storage): at::as_strided through the dispatcher costs ~1.3 us per call, and a backward pass hands out 50 + 24 of them.
inline at::Tensor arena_view(const at::Tensor& arena, c10::IntArrayRef sizes, c10::IntArrayRef strides, int64_t offset) {
  auto impl = c10::make_intrusive<c10::TensorImpl>(c10::TensorImpl::VIEW, c10::Storage(arena.storage()), arena.key_set(),
                                                   arena.dtype());
  impl->set_sizes_and_strides(sizes, strides);
  impl->set_storage_offset(arena.storage_offset() + offset);
  return at::Tensor(std::move(impl));
}
void* ptr(const c10::optional<at::Tensor>& t) { return t ? ptr(*t) : nullptr; }

struct LbaCfg {
  cgvp_dims dims;
  cgvp_layout layout;
  int nc;
  int mean;
};
LbaCfg make_cfg(const std::vector<int64_t>& c, bool bf16) {
  TORCH_CHECK(c.size() == 13, "cfg must hold 13 integers");
  LbaCfg r;
  r.dims = cgvp_dims{(int32_t)c[0], (int32_t)c[1], (int32_t)c[2], (int32_t)c[3], (int32_t)c[4], (int32_t)c[5],
                     (int32_t)c[6], (int32_t)c[7], (int32_t)c[8], bf16 ? CGVP_BF16 : CGVP_F32, CGVP_LAYER_GATED};
  check(cgvp_lba_layout(&r.dims, (int32_t)c[9], (int32_t)c[10], (int32_t)c[11], &r.layout), "cgvp_lba_layout");
  r.nc = (int)c[11];
  r.mean = c[12] != 0;
  return r;
}

// The arena-ordered parameter list as one flat fp32 buffer: zero-copy when the tensors are consecutive views of one
// storage (what gvp_hip.arena.ParamArena maintains), one cat otherwise.
at::Tensor flat_arena(const std::vector<at::Tensor>& params, bool* zero_copy) {
  const at::Tensor& p0 = params[0];
  const char* base = static_cast<const char*>(p0.data_ptr());
  const char* at_ = base;
  bool ok = p0.scalar_type() == at::kFloat;
  for (const at::Tensor& p : params) {
    if (!ok) break;
    ok = p.scalar_type() == at::kFloat && p.is_contiguous() && static_cast<const char*>(p.data_ptr()) == at_ &&
         p.storage().is_alias_of(p0.storage());
    at_ += p.numel() * 4;
  }
  *zero_copy = ok;
  if (ok) {
    int64_t n = (at_ - base) / 4;
    return at::from_blob(const_cast<char*>(base), {n}, p0.options().requires_grad(false));
  }
  std::vector<at::Tensor> flat;
  flat.reserve(params.size());
  for (const at::Tensor& p : params) flat.push_back(p.detach().reshape({-1}).to(at::kFloat));
  return at::cat(flat);
}

struct Tables { const int32_t* rowptr = nullptr; const int32_t* eperm = nullptr; const int32_t* esrc = nullptr; const int32_t* edst = nullptr; };
Tables tables_of(const std::vector<at::Tensor>& csr) {
  Tables t;
  if (csr.empty()) return t;
  TORCH_CHECK(csr.size() == 4, "csr must be [] or [rowptr, eperm, esrc, edst]");
  for (const at::Tensor& c : csr) TORCH_CHECK(c.is_cuda() && c.scalar_type() == at::kInt && c.is_contiguous(), "csr tables must be contiguous int32 CUDA tensors");
  t.rowptr = csr[0].data_ptr<int32_t>(); t.eperm = csr[1].data_ptr<int32_t>();
  t.esrc = csr[2].data_ptr<int32_t>(); t.edst = csr[3].data_ptr<int32_t>();
  return t;
}

// ===================================================================================== protein encoder
struct LbaBackward : public Node {
  LbaCfg cfg;
  at::Tensor x_s, x_v, ntypes, e_s, e_v, etypes, ws, masks;
  std::vector<at::Tensor> csr;
  std::vector<std::vector<int64_t>> shapes, strides;
  std::vector<int64_t> numels;
  at::ScalarType x_dtype;
  int64_t N = 0, E = 0;
  double dropout_p = 0;
  bool need_x = false;
  std::vector<Guarded> guarded;

  variable_list apply(variable_list&& grads) override {
    Tic tic_all("lba_bwd.apply");
    once_differentiable("CasterGvpLbaEncoderBackward");
    const size_t np = shapes.size();
    variable_list out(np + 2);
    if (grads.empty() || !grads[0].defined()) return out;
    TORCH_CHECK(ws.defined(), "caster_gvp: backward through a protein encoder pass whose saved state was already released "
                              "(call backward once, or pass retain_graph=True)");
    guard_check(guarded, "the protein encoder's backward");
    c10::DeviceGuard guard(x_s.device());
    at::Tensor g_out = ready(grads[0], "grad_output", at::kFloat);
    auto f32 = x_s.options().dtype(at::kFloat).requires_grad(false);
    at::Tensor gparams = at::empty({(int64_t)cfg.layout.total}, f32);
    const int64_t nbytes = cgvp_lba_bwd_workspace_bytes(&cfg.dims, &cfg.layout, N, E);
    if (nbytes < 0) check((int)nbytes, "cgvp_lba_bwd_workspace_bytes");
    at::Tensor bws = at::empty({nbytes}, f32.dtype(at::kByte));
    at::Tensor g_x_s, g_x_v;
    if (need_x) {
      g_x_s = at::empty({N, (int64_t)cfg.dims.node_in_s}, f32);
      g_x_v = at::empty({N, (int64_t)cfg.dims.node_in_v, 3}, f32);
    }
    const Tables t = tables_of(csr);
    cgvp_lba_batch b{N, E, (const float*)ptr(x_s), (const float*)ptr(x_v), (const int64_t*)ptr(ntypes),
                     (const float*)ptr(e_s), (const float*)ptr(e_v), (const int64_t*)ptr(etypes), nullptr,
                     t.rowptr, t.eperm, t.esrc, t.edst};
    {
      Tic tic_c("lba_bwd.c_call");
      check(cgvp_lba_backward_pass(&cfg.dims, &cfg.layout, &b, cfg.mean, (float)dropout_p, (const float*)ptr(masks), ptr(ws),
                                   (const float*)ptr(g_out), ptr(bws), (float*)ptr(gparams), (float*)ptr(g_x_s),
                                   (float*)ptr(g_x_v), nullptr, nullptr, current_stream(x_s)),
            "cgvp_lba_backward_pass");
    }
    Tic tic_v("lba_bwd.views");
    int64_t off = 0;
    for (size_t i = 0; i < np; ++i) {      // one view op per parameter (as_strided), not narrow + view
      if (task_should_compute_output(i)) out[i] = arena_view(gparams, shapes[i], strides[i], off);
      off += numels[i];
    }
    if (LeafScatter::run(*this, np, out, gparams))
      for (size_t i = 0; i < np; ++i) out[i] = at::Tensor();
    if (need_x) {
      out[np] = x_dtype == at::kFloat ? g_x_s : g_x_s.to(x_dtype);
      out[np + 1] = x_dtype == at::kFloat ? g_x_v : g_x_v.to(x_dtype);
    }
    return out;
  }
  void release_variables() override {
    ws.reset(); masks.reset(); x_s.reset(); x_v.reset(); e_s.reset(); e_v.reset(); ntypes.reset(); etypes.reset();
    csr.clear(); guarded.clear();
  }
  std::string name() const override { return "CasterGvpLbaEncoderBackward"; }
};

// -> (out [N, out_s], ws: the forward workspace (uint8; undefined when the pass saved nothing), whether the parameters
//     were consecutive views of one arena (else they were concatenated: the caller re-seats them for the next pass))
std::tuple<at::Tensor, at::Tensor, bool> lba_encoder(std::vector<at::Tensor> params, at::Tensor x_s_in, at::Tensor x_v_in,
                                               at::Tensor ntypes_in, at::Tensor e_s_in, at::Tensor e_v_in,
                                               at::Tensor etypes_in, at::Tensor edge_index_in, std::vector<at::Tensor> csr,
                                               std::vector<int64_t> cfgv, double dropout_p, bool save_state,
                                               c10::optional<at::Tensor> masks, c10::optional<at::Tensor> rng_state,
                                               c10::optional<at::Tensor> counters, bool fuse) {
  TORCH_CHECK(!params.empty(), "no parameters");
  const bool bf16 = x_s_in.scalar_type() == at::kBFloat16;
  const at::ScalarType sdt = bf16 ? at::kBFloat16 : at::kFloat;
  TORCH_CHECK(x_s_in.scalar_type() == at::kFloat || bf16, "x_s: expected float32 or bfloat16, got ", x_s_in.scalar_type());
  const LbaCfg cfg = make_cfg(cfgv, bf16);
  at::Tensor x_s = ready(x_s_in, "x_s", sdt), x_v = ready(x_v_in, "x_v", sdt);
  at::Tensor e_s = ready(e_s_in, "eattr_s", sdt), e_v = ready(e_v_in, "eattr_v", sdt);
  TORCH_CHECK(edge_index_in.dim() == 2 && edge_index_in.size(0) == 2, "edge_index must be [2, E]");
  const int64_t N = x_s.size(0), E = edge_index_in.size(1);
  const bool tables = !csr.empty();
  if (x_s.dim() != 2 || x_s.size(1) != cfg.dims.node_in_s || x_v.dim() != 3 || x_v.size(0) != N ||
      x_v.size(1) != cfg.dims.node_in_v || e_s.dim() != 2 || e_s.size(1) != cfg.dims.edge_in_s || e_v.dim() != 3 ||
      e_v.size(1) != cfg.dims.edge_in_v || e_s.size(0) != e_v.size(0) || (!tables && e_s.size(0) != E))
    throw NotImplemented("feature shapes do not match the compiled CASTER-DTA configuration");
  at::Tensor ntypes, etypes, edge_index;
  if (cfg.layout.nt_node > 0) ntypes = ready(ntypes_in, "ntypes", at::kLong);
  if (cfg.layout.nt_edge > 0) etypes = ready(etypes_in, "etypes", at::kLong);
  if (!tables) edge_index = ready(edge_index_in, "edge_index", at::kLong);
  bool zero_copy = false;
  at::Tensor flat = flat_arena(params, &zero_copy);
  TORCH_CHECK(flat.numel() == cfg.layout.total, "parameter arena has ", flat.numel(), " floats, kernels expect ", cfg.layout.total);
  const bool drop = dropout_p > 0 && save_state;
  const bool draw = drop && !(masks && masks->numel() > 0);
  TORCH_CHECK(!draw || rng_state, "training-mode dropout needs the persistent generator state");
  TORCH_CHECK(tables || counters, "the CSR build needs the persistent counters");
  c10::DeviceGuard guard(x_s.device());
  cgvp_lba_fwd_ws wsd;
  check(cgvp_lba_fwd_workspace(&cfg.dims, &cfg.layout, N, E, save_state ? 1 : 0, &wsd), "cgvp_lba_fwd_workspace");
  auto opts = x_s.options().requires_grad(false);
  at::Tensor ws = at::empty({wsd.total}, opts.dtype(at::kByte));
  at::Tensor out = at::empty({N, (int64_t)cfg.dims.out_s}, opts.dtype(sdt));
  const Tables t = tables_of(csr);
  cgvp_lba_batch b{N, E, (const float*)ptr(x_s), (const float*)ptr(x_v), (const int64_t*)ptr(ntypes),
                   (const float*)ptr(e_s), (const float*)ptr(e_v), (const int64_t*)ptr(etypes),
                   (const int64_t*)ptr(edge_index), t.rowptr, t.eperm, t.esrc, t.edst};
  Tic tic_f("lba_fwd.c_call");
  const int rc = cgvp_lba_forward_pass(&cfg.dims, &cfg.layout, (const float*)ptr(flat), &b, cfg.mean,
                                       save_state ? (float)dropout_p : 0.f, draw ? (uint64_t*)ptr(rng_state) : nullptr,
                                       drop ? (const float*)ptr(masks) : nullptr, tables ? nullptr : (int32_t*)ptr(counters),
                                       ptr(ws), save_state ? 1 : 0, fuse ? 0 : CGVP_PASS_UNFUSED, (float*)ptr(out),
                                       current_stream(x_s));
  if (rc != 0 && counters) counters->zero_();     // the persistent counters must not stay half-used
  check(rc, "cgvp_lba_forward_pass");
  if (!save_state) return {out, at::Tensor(), zero_copy};
  // autograd: one node for the whole pass
  bool any = false;
  for (const at::Tensor& p : params) any = any || p.requires_grad();
  const bool need_x = x_s_in.requires_grad() || x_v_in.requires_grad();
  if (at::GradMode::is_enabled() && (any || need_x)) {
    auto node = std::shared_ptr<LbaBackward>(new LbaBackward(), torch::autograd::deleteNode);
    torch::autograd::edge_list edges = torch::autograd::collect_next_edges(params, x_s_in, x_v_in);
    node->set_next_edges(std::move(edges));
    node->cfg = cfg;
    node->x_s = x_s; node->x_v = x_v; node->ntypes = ntypes; node->e_s = e_s; node->e_v = e_v; node->etypes = etypes;
    node->ws = ws;
    if (drop && masks && masks->numel() > 0) node->masks = *masks;
    node->csr = csr;
    node->N = N; node->E = E; node->dropout_p = drop ? dropout_p : 0.0; node->need_x = need_x;
    node->x_dtype = x_s_in.scalar_type();
    guard_add(node->guarded, x_s, "x_s"); guard_add(node->guarded, x_v, "x_v");
    guard_add(node->guarded, e_s, "eattr_s"); guard_add(node->guarded, e_v, "eattr_v");
    node->shapes.reserve(params.size());
    node->strides.reserve(params.size());
    for (const at::Tensor& p : params) {
      node->shapes.push_back(p.sizes().vec());
      { auto st = c10::contiguous_strides(p.sizes()); node->strides.emplace_back(st.begin(), st.end()); }
      node->numels.push_back(p.numel());
    }
    torch::autograd::set_history(out, node);
  }
  return {out, ws, zero_copy};
}

// ===================================================================================== drug encoder
struct GineMeta {
  cgvp_gine_cfg cfg;
  int L;
};
GineMeta gine_meta(const std::vector<int64_t>& widths, int64_t nt, int64_t net, int64_t edge_dim, double slope) {
  GineMeta m;
  m.L = (int)widths.size() - 1;
  TORCH_CHECK(m.L >= 1 && m.L <= CGVP_GINE_MAX_LAYERS, "the GINE pass supports 1..", CGVP_GINE_MAX_LAYERS, " layers");
  m.cfg.num_layers = m.L;
  for (int i = 0; i <= CGVP_GINE_MAX_LAYERS; ++i) m.cfg.widths[i] = i <= m.L ? (int32_t)widths[i] : 0;
  m.cfg.num_ntypes = (int32_t)nt; m.cfg.num_etypes = (int32_t)net; m.cfg.edge_dim = (int32_t)edge_dim; m.cfg.act_slope = (float)slope;
  return m;
}
// ONE-LEAF MODE (HomoMoleculeGNN_GINE.fuse_parameters): `params` is a single flat tensor holding the 7 L tensors end to
// end in the order below; it is split into views here (shapes follow from the configuration) and the backward node
// returns ONE gradient, the flat buffer the kernels' gradients are written to anyway.
std::vector<at::Tensor> gine_split_arena(const at::Tensor& arena, const GineMeta& m) {
  std::vector<at::Tensor> v;
  const at::Tensor flat = arena.detach();
  TORCH_CHECK(flat.dim() == 1 && flat.is_contiguous(), "the fused GINE parameter arena must be a contiguous 1-D tensor");
  int64_t off = 0;
  auto take = [&](std::initializer_list<int64_t> shape) {
    int64_t n = 1;
    for (int64_t d : shape) n *= d;
    TORCH_CHECK(off + n <= flat.numel(), "the fused GINE parameter arena is shorter than the configuration needs");
    v.push_back(flat.narrow(0, off, n).view(at::IntArrayRef(shape.begin(), shape.size())));
    off += n;
  };
  const int64_t ew = m.cfg.num_etypes + m.cfg.edge_dim;
  for (int l = 0; l < m.L; ++l) {
    const int64_t a = m.cfg.widths[l], b = m.cfg.widths[l + 1];
    take({1}); take({b, a}); take({b}); take({b, b}); take({b}); take({a, ew}); take({a});
  }
  TORCH_CHECK(off == flat.numel(), "the fused GINE parameter arena has ", flat.numel(), " floats, the configuration needs ", off);
  return v;
}

// params: 7 tensors per layer in slab / state_dict order  eps | w0 | b0 | w1 | b1 | we | be
void gine_weights(const std::vector<at::Tensor>& params, int L, std::vector<at::Tensor>& keep, std::vector<cgvp_gine_w>& w) {
  TORCH_CHECK((int)params.size() == 7 * L, "expected 7 parameter tensors per GINE layer");
  keep.reserve(params.size());
  for (const at::Tensor& p : params) keep.push_back(ready(p.detach(), "weight", at::kFloat));
  w.resize(L);
  for (int l = 0; l < L; ++l) {
    const at::Tensor* k = &keep[7 * l];
    w[l] = cgvp_gine_w{(const float*)k[0].data_ptr(), (const float*)k[5].data_ptr(), (const float*)k[6].data_ptr(),
                       (const float*)k[1].data_ptr(), (const float*)k[2].data_ptr(), (const float*)k[3].data_ptr(),
                       (const float*)k[4].data_ptr()};
  }
}

struct GineBackward : public Node {
  GineMeta meta;
  std::vector<at::Tensor> params, csr, masks;
  at::Tensor x, ntypes, eattr, etypes, ws;
  int64_t N = 0, E = 0;
  double dropout_p = 0;
  bool need_x = false;
  bool one_leaf = false;          // the caller's `params` was the single fused arena: return one gradient
  int max_workgroups = 0;
  std::vector<Guarded> guarded;

  variable_list apply(variable_list&& grads) override {
    Tic tic_all("gine_bwd.apply");
    once_differentiable("CasterGvpGineEncoderBackward");
    const size_t np = one_leaf ? 1 : params.size();
    variable_list out(np + 1);
    if (grads.empty() || !grads[0].defined()) return out;
    TORCH_CHECK(ws.defined(), "caster_gvp: backward through a drug encoder pass whose saved state was already released "
                              "(call backward once, or pass retain_graph=True)");
    guard_check(guarded, "the drug encoder's backward");
    c10::DeviceGuard guard(x.device());
    at::Tensor g_out = ready(grads[0], "grad_output", at::kFloat);
    auto f32 = x.options().dtype(at::kFloat).requires_grad(false);
    int64_t total = 0;
    for (const at::Tensor& p : params) total += p.numel();
    at::Tensor gflat = at::empty({total}, f32);
    const int64_t nbytes = cgvp_gine_bwd_workspace_bytes(&meta.cfg, N, E);
    if (nbytes < 0) check((int)nbytes, "cgvp_gine_bwd_workspace_bytes");
    at::Tensor bws = at::empty({nbytes}, f32.dtype(at::kByte));
    at::Tensor g_x;
    if (need_x) g_x = at::empty({N, (int64_t)(meta.cfg.widths[0] - meta.cfg.num_ntypes)}, f32);
    std::vector<at::Tensor> keep;
    std::vector<cgvp_gine_w> w;
    gine_weights(params, meta.L, keep, w);
    const Tables t = tables_of(csr);
    cgvp_gine_batch b{N, E, (const float*)ptr(x), (const int64_t*)ptr(ntypes), (const float*)ptr(eattr),
                      (const int64_t*)ptr(etypes), nullptr, t.rowptr, t.eperm, t.esrc, t.edst};
    std::vector<const float*> mp;
    for (const at::Tensor& m : masks) mp.push_back((const float*)ptr(m));
    {
      Tic tic_c("gine_bwd.c_call");
      check(cgvp_gine_backward_pass(&meta.cfg, w.data(), &b, (float)dropout_p, mp.empty() ? nullptr : mp.data(), ptr(ws),
                                    (const float*)ptr(g_out), ptr(bws), (float*)ptr(gflat), (float*)ptr(g_x), max_workgroups,
                                    current_stream(x)),
            "cgvp_gine_backward_pass");
    }
    if (one_leaf) {
      if (task_should_compute_output(0)) out[0] = gflat;
    } else {
      int64_t off = 0;
      for (size_t i = 0; i < np; ++i) {
        const int64_t n = params[i].numel();
        if (task_should_compute_output(i)) out[i] = arena_view(gflat, params[i].sizes(), c10::contiguous_strides(params[i].sizes()), off);
        off += n;
      }
      if (LeafScatter::run(*this, np, out, gflat))
        for (size_t i = 0; i < np; ++i) out[i] = at::Tensor();
    }
    if (need_x) out[np] = g_x;
    return out;
  }
  void release_variables() override {
    ws.reset(); x.reset(); eattr.reset(); ntypes.reset(); etypes.reset(); params.clear(); csr.clear(); masks.clear();
    guarded.clear();
  }
  std::string name() const override { return "CasterGvpGineEncoderBackward"; }
};

std::tuple<at::Tensor, at::Tensor> gine_encoder(std::vector<at::Tensor> params_in, at::Tensor x_in, at::Tensor ntypes_in,
                                                at::Tensor eattr_in, at::Tensor etypes_in, at::Tensor edge_index_in,
                                                std::vector<at::Tensor> csr, std::vector<int64_t> widths, int64_t num_ntypes,
                                                int64_t num_etypes, double slope, double dropout_p, bool save_state,
                                                std::vector<at::Tensor> masks, c10::optional<at::Tensor> rng_state,
                                                c10::optional<at::Tensor> counters, int64_t variant, int64_t bwd_workgroups) {
  TORCH_CHECK(x_in.scalar_type() == at::kFloat, "x: expected float32, got ", x_in.scalar_type());
  TORCH_CHECK(eattr_in.scalar_type() == at::kFloat, "eattr: expected float32, got ", eattr_in.scalar_type());
  at::Tensor x = ready(x_in, "x"), eattr = ready(eattr_in, "eattr");
  TORCH_CHECK(edge_index_in.dim() == 2 && edge_index_in.size(0) == 2, "edge_index must be [2, E]");
  TORCH_CHECK(x.dim() == 2 && eattr.dim() == 2, "x must be [N, F], eattr [E, D]");
  const int64_t N = x.size(0), E = edge_index_in.size(1);
  const GineMeta meta = gine_meta(widths, num_ntypes, num_etypes, eattr.size(1), slope);
  if (x.size(1) != widths[0] - num_ntypes) throw py::value_error("x has " + std::to_string(x.size(1)) + " columns, expected " + std::to_string(widths[0] - num_ntypes));
  const bool tables = !csr.empty();
  at::Tensor ntypes, etypes, edge_index;
  if (num_ntypes > 0) ntypes = ready(ntypes_in, "ntypes", at::kLong);
  if (num_etypes > 0) etypes = ready(etypes_in, "etypes", at::kLong);
  if (!tables) edge_index = ready(edge_index_in, "edge_index", at::kLong);
  const bool one_leaf = params_in.size() == 1;
  const std::vector<at::Tensor> params = one_leaf ? gine_split_arena(params_in[0], meta) : params_in;
  std::vector<at::Tensor> keep;
  std::vector<cgvp_gine_w> w;
  gine_weights(params, meta.L, keep, w);
  const bool drop = dropout_p > 0 && save_state && meta.L > 1;
  const bool pinned = drop && !masks.empty();
  const bool draw = drop && !pinned;
  TORCH_CHECK(!draw || rng_state, "training-mode dropout needs the persistent generator state");
  TORCH_CHECK(tables || counters, "the CSR build needs the persistent counters");
  c10::DeviceGuard guard(x.device());
  cgvp_gine_fwd_ws wsd;
  check(cgvp_gine_fwd_workspace(&meta.cfg, N, E, save_state ? 1 : 0, &wsd), "cgvp_gine_fwd_workspace");
  auto opts = x.options().requires_grad(false);
  at::Tensor ws = at::empty({wsd.total}, opts.dtype(at::kByte));
  at::Tensor out = at::empty({N, widths.back()}, opts);
  const Tables t = tables_of(csr);
  cgvp_gine_batch b{N, E, (const float*)ptr(x), (const int64_t*)ptr(ntypes), (const float*)ptr(eattr),
                    (const int64_t*)ptr(etypes), (const int64_t*)ptr(edge_index), t.rowptr, t.eperm, t.esrc, t.edst};
  std::vector<const float*> mp;
  if (pinned) {
    TORCH_CHECK((int)masks.size() == meta.L - 1, "expected one mask per layer but the last");
    for (at::Tensor& m : masks) { m = ready(m, "mask", at::kFloat); mp.push_back((const float*)ptr(m)); }
  }
  const int rc = cgvp_gine_forward_pass(&meta.cfg, w.data(), &b, save_state ? (float)dropout_p : 0.f,
                                        draw ? (uint64_t*)ptr(rng_state) : nullptr, pinned ? mp.data() : nullptr,
                                        tables ? nullptr : (int32_t*)ptr(counters), ptr(ws), save_state ? 1 : 0, (int32_t)variant,
                                        (float*)ptr(out), current_stream(x));
  if (rc != 0 && counters) counters->zero_();
  check(rc, "cgvp_gine_forward_pass");
  if (!save_state) return {out, at::Tensor()};
  bool any = false;
  for (const at::Tensor& p : params_in) any = any || p.requires_grad();
  const bool need_x = x_in.requires_grad();
  if (at::GradMode::is_enabled() && (any || need_x)) {
    auto node = std::shared_ptr<GineBackward>(new GineBackward(), torch::autograd::deleteNode);
    node->set_next_edges(torch::autograd::collect_next_edges(params_in, x_in));
    node->meta = meta;
    node->one_leaf = one_leaf;
    node->params.reserve(params.size());
    for (const at::Tensor& p : params) {
      node->params.push_back(p.detach());
      guard_add(node->guarded, node->params.back(), "a drug-encoder weight");     // (detach() shares the version counter)
    }
    node->csr = csr;
    if (pinned) node->masks = masks;
    node->x = x; node->ntypes = ntypes; node->eattr = eattr; node->etypes = etypes; node->ws = ws;
    node->N = N; node->E = E; node->dropout_p = drop ? dropout_p : 0.0; node->need_x = need_x;
    node->max_workgroups = (int)bwd_workgroups;
    guard_add(node->guarded, x, "x"); guard_add(node->guarded, eattr, "eattr");
    torch::autograd::set_history(out, node);
  }
  return {out, ws};
}

// =================================================================================== the joint head's row-wise ops
// The same entry points gvp_hip/head_ops.py reaches through torch.library custom ops (which stay the path torch.compile
// traces), as C++ autograd functions for the EAGER loop: a Python custom op costs 60-70 us of host time per call each
// way (dispatcher -> Python -> ctypes, a Python autograd.Function around it), the head makes ~35 such calls per step
// and an eager whole-model step was host-bound at 4.3-5.3 ms for 1.6 ms of device work (tools/host_profile_joint.py).
// No arithmetic here: shapes, allocation from the caching allocator, one C call each.
using torch::autograd::AutogradContext;
using torch::autograd::Function;

at::Tensor rows_f32(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat, name, ": expected an fp32 tensor on the GPU");
  return t.is_contiguous() ? t : t.contiguous();
}
inline cgvp_rng rng_of(const at::Tensor& pair, double p, int64_t site) {
  return cgvp_rng{reinterpret_cast<const uint64_t*>(pair.data_ptr()), (float)p, (int32_t)site};
}
inline void rows_dim(const at::Tensor& t, int64_t* R, int32_t* D) {
  *D = (int32_t)t.size(-1);
  *R = t.numel() / (*D > 0 ? *D : 1);
}

at::Tensor head_rng_next(at::Tensor state) {
  at::Tensor out = at::empty({2}, state.options());
  check(cgvp_rng_next(reinterpret_cast<uint64_t*>(state.data_ptr()), reinterpret_cast<uint64_t*>(out.data_ptr()),
                      current_stream(state)), "cgvp_rng_next");
  return out;
}

// dropout_p(LeakyReLU_slope(t)); slope 0 = ReLU   (joint_gnn.py:188-198, :388)
struct ActDropout : public Function<ActDropout> {
  static at::Tensor forward(AutogradContext* ctx, at::Tensor t_in, at::Tensor pair, int64_t site, double p, double slope) {
    at::Tensor t = rows_f32(t_in, "act_dropout");
    int64_t R; int32_t D;
    rows_dim(t, &R, &D);
    at::Tensor y = at::empty_like(t);
    const cgvp_rng rng = rng_of(pair, p, site);
    check(cgvp_act_dropout_fwd((const float*)t.data_ptr(), p > 0 ? &rng : nullptr, (float)slope, R, D, (float*)y.data_ptr(),
                               current_stream(t)), "cgvp_act_dropout_fwd");
    ctx->save_for_backward({y, pair});
    ctx->saved_data["site"] = site; ctx->saved_data["p"] = p; ctx->saved_data["slope"] = slope;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list go) {
    auto saved = ctx->get_saved_variables();
    at::Tensor g = rows_f32(go[0], "act_dropout backward");
    const at::Tensor &y = saved[0], &pair = saved[1];
    const double p = ctx->saved_data["p"].toDouble();
    int64_t R; int32_t D;
    rows_dim(g, &R, &D);
    at::Tensor gt = at::empty_like(g);
    const cgvp_rng rng = rng_of(pair, p, ctx->saved_data["site"].toInt());
    check(cgvp_act_dropout_bwd((const float*)g.data_ptr(), (const float*)y.data_ptr(), p > 0 ? &rng : nullptr,
                               (float)ctx->saved_data["slope"].toDouble(), R, D, (float*)gt.data_ptr(), current_stream(g)),
          "cgvp_act_dropout_bwd");
    return {gt, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

// x + dropout_p(a)   (x undefined: dropout_p(a) alone)
struct DropoutAdd : public Function<DropoutAdd> {
  static at::Tensor forward(AutogradContext* ctx, at::Tensor a_in, at::Tensor x_in, at::Tensor pair, int64_t site, double p) {
    at::Tensor a = rows_f32(a_in, "dropout_add");
    const bool has_x = x_in.defined() && x_in.numel() > 0;
    at::Tensor x = has_x ? rows_f32(x_in, "dropout_add") : at::Tensor();
    int64_t R; int32_t D;
    rows_dim(a, &R, &D);
    at::Tensor y = at::empty_like(a);
    const cgvp_rng rng = rng_of(pair, p, site);
    check(cgvp_dropout_add((const float*)a.data_ptr(), has_x ? (const float*)x.data_ptr() : nullptr, p > 0 ? &rng : nullptr, R, D,
                           (float*)y.data_ptr(), current_stream(a)), "cgvp_dropout_add");
    ctx->save_for_backward({pair});
    ctx->saved_data["site"] = site; ctx->saved_data["p"] = p; ctx->saved_data["has_x"] = has_x;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list go) {
    const at::Tensor pair = ctx->get_saved_variables()[0];
    at::Tensor g = rows_f32(go[0], "dropout_add backward");
    at::Tensor ga;
    if (ctx->needs_input_grad(0)) {
      const double p = ctx->saved_data["p"].toDouble();
      int64_t R; int32_t D;
      rows_dim(g, &R, &D);
      ga = at::empty_like(g);
      const cgvp_rng rng = rng_of(pair, p, ctx->saved_data["site"].toInt());
      check(cgvp_dropout_scale((const float*)g.data_ptr(), p > 0 ? &rng : nullptr, R, D, (float*)ga.data_ptr(), current_stream(g)),
            "cgvp_dropout_scale");
    }
    return {ga, (ctx->saved_data["has_x"].toBool() && ctx->needs_input_grad(1)) ? g : at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor()};
  }
};

// nn.LayerNorm over the last dim of compact rows [R, D]   (joint_gnn.py:376-389)
struct RowLayerNorm : public Function<RowLayerNorm> {
  static at::Tensor forward(AutogradContext* ctx, at::Tensor x_in, at::Tensor w, at::Tensor b, double eps) {
    at::Tensor x = rows_f32(x_in, "layer_norm");
    const int64_t R = x.size(0);
    const int32_t D = (int32_t)x.size(1);
    at::Tensor y = at::empty_like(x), mean = at::empty({R}, x.options()), rstd = at::empty({R}, x.options());
    check(cgvp_layer_norm_fwd((const float*)x.data_ptr(), (const float*)w.data_ptr(), (const float*)b.data_ptr(), R, D, (float)eps,
                              (float*)y.data_ptr(), (float*)mean.data_ptr(), (float*)rstd.data_ptr(), current_stream(x)),
          "cgvp_layer_norm_fwd");
    ctx->save_for_backward({x, mean, rstd, w});
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list go) {
    auto sv = ctx->get_saved_variables();
    const at::Tensor &x = sv[0], &mean = sv[1], &rstd = sv[2], &w = sv[3];
    at::Tensor gy = rows_f32(go[0], "layer_norm backward");
    const int64_t R = x.size(0);
    const int32_t D = (int32_t)x.size(1);
    const int64_t n = cgvp_layer_norm_bwd_workspace_floats(R, D);
    if (n < 0) check((int)n, "cgvp_layer_norm_bwd_workspace_floats");
    at::Tensor ws = at::empty({n > 0 ? n : 1}, x.options()), gx = at::empty_like(x), gwb = at::empty({2 * (int64_t)D}, x.options());
    check(cgvp_layer_norm_bwd((const float*)gy.data_ptr(), (const float*)x.data_ptr(), (const float*)mean.data_ptr(),
                              (const float*)rstd.data_ptr(), (const float*)w.data_ptr(), R, D, (float*)gx.data_ptr(),
                              (float*)ws.data_ptr(), (float*)gwb.data_ptr(), current_stream(x)), "cgvp_layer_norm_bwd");
    return {gx, gwb.narrow(0, 0, D), gwb.narrow(0, D, D), at::Tensor()};
  }
};

// F.linear whose weight / bias gradients come from the split-row kernel   (joint_gnn.py:188-198, :376-389)
struct FastLinear : public Function<FastLinear> {
  static at::Tensor forward(AutogradContext* ctx, at::Tensor x, at::Tensor w, at::Tensor b) {
    ctx->save_for_backward({x, w});
    return at::linear(x, w, b);
  }
  static variable_list backward(AutogradContext* ctx, variable_list go) {
    auto sv = ctx->get_saved_variables();
    const at::Tensor &x = sv[0], &w = sv[1];
    at::Tensor gy = rows_f32(go[0], "linear backward");
    at::Tensor gx = ctx->needs_input_grad(0) ? at::matmul(gy, w) : at::Tensor();
    const int64_t R = x.size(0);
    const int32_t I = (int32_t)w.size(1), O = (int32_t)w.size(0);
    const int64_t n = cgvp_linear_wgrad_workspace_floats(R, I, O);
    if (n < 0) check((int)n, "cgvp_linear_wgrad_workspace_floats");
    at::Tensor xs = rows_f32(x, "linear backward");
    at::Tensor ws = at::empty({n > 0 ? n : 1}, gy.options()), flat = at::empty({(int64_t)O * I + O}, gy.options());
    check(cgvp_linear_wgrad((const float*)xs.data_ptr(), (const float*)gy.data_ptr(), R, I, O, (float*)ws.data_ptr(),
                            (float*)flat.data_ptr(), current_stream(gy)), "cgvp_linear_wgrad");
    return {gx, flat.narrow(0, 0, (int64_t)O * I).view({O, I}), flat.narrow(0, (int64_t)O * I, O)};
  }
};

// The varlen cross-attention core, both directions in one launch (gvp_hip/attention_ops.py caster_gvp::cross_attention)
struct CrossAttention : public Function<CrossAttention> {
  static void fill(cgvp_attn_problem* P, const at::Tensor& q_r, const at::Tensor& k_a, const at::Tensor& v_a, const at::Tensor& q_a,
                   const at::Tensor& k_r, const at::Tensor& v_r, const at::Tensor& rptr, const at::Tensor& aptr) {
    P[0] = cgvp_attn_problem{};
    P[1] = cgvp_attn_problem{};
    P[0].q = (const float*)q_r.data_ptr(); P[0].k = (const float*)k_a.data_ptr(); P[0].v = (const float*)v_a.data_ptr();
    P[0].q_ptr = (const int64_t*)rptr.data_ptr(); P[0].k_ptr = (const int64_t*)aptr.data_ptr();
    P[0].num_q = q_r.size(0); P[0].num_k = k_a.size(0);
    P[1].q = (const float*)q_a.data_ptr(); P[1].k = (const float*)k_r.data_ptr(); P[1].v = (const float*)v_r.data_ptr();
    P[1].q_ptr = (const int64_t*)aptr.data_ptr(); P[1].k_ptr = (const int64_t*)rptr.data_ptr();
    P[1].num_q = q_a.size(0); P[1].num_k = k_r.size(0);
  }
  static variable_list forward(AutogradContext* ctx, at::Tensor q_r_, at::Tensor k_a_, at::Tensor v_a_, at::Tensor q_a_, at::Tensor k_r_,
                               at::Tensor v_r_, at::Tensor rptr_, at::Tensor aptr_, int64_t heads) {
    const int64_t E = heads * 16;
    at::Tensor t[6] = {rows_f32(q_r_, "q_r"), rows_f32(k_a_, "k_a"), rows_f32(v_a_, "v_a"), rows_f32(q_a_, "q_a"),
                       rows_f32(k_r_, "k_r"), rows_f32(v_r_, "v_r")};
    for (const at::Tensor& x : t)
      if (x.dim() != 2 || x.size(1) != E) throw NotImplemented("cross_attention: the kernels are compiled for head_dim 16 (embed = 16 * heads)");
    TORCH_CHECK(t[0].size(0) == t[4].size(0) && t[0].size(0) == t[5].size(0) && t[3].size(0) == t[1].size(0) && t[3].size(0) == t[2].size(0),
                "row counts of the residue / atom tensors disagree");
    at::Tensor rptr = rptr_.to(at::kLong).contiguous(), aptr = aptr_.to(at::kLong).contiguous();
    TORCH_CHECK(rptr.is_cuda() && aptr.is_cuda() && rptr.dim() == 1 && rptr.sizes() == aptr.sizes() && rptr.numel() >= 1,
                "rptr / aptr must both be [num_pairs + 1] on the GPU");
    const int64_t N = t[0].size(0), Na = t[3].size(0), B = rptr.numel() - 1;
    at::Tensor o_r = at::empty({N, E}, t[0].options()), o_a = at::empty({Na, E}, t[0].options());
    at::Tensor lse_r = at::empty({N, heads}, t[0].options()), lse_a = at::empty({Na, heads}, t[0].options());
    cgvp_attn_problem P[2];
    fill(P, t[0], t[1], t[2], t[3], t[4], t[5], rptr, aptr);
    P[0].out = (float*)ptr(o_r); P[0].lse = (float*)ptr(lse_r);
    P[1].out = (float*)ptr(o_a); P[1].lse = (float*)ptr(lse_a);
    check(cgvp_attn_fwd(P, 2, B, (int32_t)heads, 0.25f, current_stream(t[0])), "cgvp_attn_fwd");
    ctx->save_for_backward({t[0], t[1], t[2], t[3], t[4], t[5], o_r, o_a, lse_r, lse_a, rptr, aptr});
    ctx->saved_data["heads"] = heads;
    ctx->mark_non_differentiable({lse_r, lse_a});
    return {o_r, o_a, lse_r, lse_a};
  }
  static variable_list backward(AutogradContext* ctx, variable_list go) {
    auto sv = ctx->get_saved_variables();
    const int64_t heads = ctx->saved_data["heads"].toInt();
    at::Tensor g_o_r = go[0].defined() ? rows_f32(go[0], "d o_r") : at::zeros_like(sv[6]);
    at::Tensor g_o_a = go[1].defined() ? rows_f32(go[1], "d o_a") : at::zeros_like(sv[7]);
    at::Tensor outs[6];
    for (int i = 0; i < 6; ++i) outs[i] = at::empty_like(sv[i]);
    at::Tensor d_r = at::empty_like(sv[8]), d_a = at::empty_like(sv[9]);
    cgvp_attn_problem P[2];
    fill(P, sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[10], sv[11]);
    P[0].out = (float*)ptr(sv[6]); P[0].lse = (float*)ptr(sv[8]); P[0].g_out = (const float*)ptr(g_o_r); P[0].delta = (float*)ptr(d_r);
    P[0].g_q = (float*)ptr(outs[0]); P[0].g_k = (float*)ptr(outs[1]); P[0].g_v = (float*)ptr(outs[2]);
    P[1].out = (float*)ptr(sv[7]); P[1].lse = (float*)ptr(sv[9]); P[1].g_out = (const float*)ptr(g_o_a); P[1].delta = (float*)ptr(d_a);
    P[1].g_q = (float*)ptr(outs[3]); P[1].g_k = (float*)ptr(outs[4]); P[1].g_v = (float*)ptr(outs[5]);
    check(cgvp_attn_bwd(P, 2, sv[10].numel() - 1, (int32_t)heads, 0.25f, current_stream(sv[0])), "cgvp_attn_bwd");
    // the two directions share no operand: a residue-side tensor gets its gradient from exactly one of them
    return {outs[0], outs[1], outs[2], outs[3], outs[4], outs[5], at::Tensor(), at::Tensor(), at::Tensor()};
  }
};
std::vector<at::Tensor> head_cross_attention(at::Tensor q_r, at::Tensor k_a, at::Tensor v_a, at::Tensor q_a, at::Tensor k_r, at::Tensor v_r,
                                             at::Tensor rptr, at::Tensor aptr, int64_t heads) {
  return CrossAttention::apply(q_r, k_a, v_a, q_a, k_r, v_r, rptr, aptr, heads);
}

at::Tensor head_act_dropout(at::Tensor t, at::Tensor pair, int64_t site, double p, double slope) {
  return ActDropout::apply(t, pair, site, p, slope);
}
at::Tensor head_dropout_add(at::Tensor a, c10::optional<at::Tensor> x, at::Tensor pair, int64_t site, double p) {
  return DropoutAdd::apply(a, x ? *x : at::Tensor(), pair, site, p);
}
at::Tensor head_layer_norm(at::Tensor x, at::Tensor w, at::Tensor b, double eps) { return RowLayerNorm::apply(x, w, b, eps); }
at::Tensor head_linear(at::Tensor x, at::Tensor w, at::Tensor b) { return FastLinear::apply(x, w, b); }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  py::register_exception_translator([](std::exception_ptr p) {
    try { if (p) std::rethrow_exception(p); }
    catch (const NotImplemented& e) { PyErr_SetString(PyExc_NotImplementedError, e.what()); }
  });
  m.doc() = "eager fast path of caster-dta_amd: C++ autograd nodes over the whole-pass C ABI of libcaster_gvp.so";
  m.def("lba_encoder", &lba_encoder, "VectorProteinGNN_LBAModel.forward (protein_gnn.py:361-388) with autograd");
  m.def("gine_encoder", &gine_encoder, "HomoMoleculeGNN_GINE.forward (molecule_gnn.py:254-268) with autograd");
  // the joint head's row-wise ops (gvp_hip/head_ops.py uses them in eager mode, the torch.library ops under torch.compile)
  m.def("head_rng_next", &head_rng_next, "advance the persistent {seed, offset} state, return this step's pair");
  m.def("head_act_dropout", &head_act_dropout, "dropout_p(LeakyReLU_slope(t)) with autograd");
  m.def("head_dropout_add", &head_dropout_add, "x + dropout_p(a) with autograd (x may be None)");
  m.def("head_layer_norm", &head_layer_norm, "row-wise LayerNorm with autograd");
  m.def("head_linear", &head_linear, "F.linear with the split-row weight / bias gradient kernel");
  m.def("head_cross_attention", &head_cross_attention, "varlen residue <-> atom cross attention (both directions) with autograd");
  // the version of include/caster_gvp.h this bridge was COMPILED against (struct layouts, argument lists); the library
  // loaded at run time reports its own through cgvp_abi_version(): _lib.bridge() requires all three to agree
  m.def("abi_version", []() { return (int)CGVP_ABI_VERSION; });
  m.def("library_abi_version", []() { return (int)cgvp_abi_version(); });
  m.def("stale_grad_accumulators", [](std::vector<at::Tensor> params) {
          // A leaf's AccumulateGrad node remembers the stream that was current when it was CREATED, and lives as long as any
          // autograd graph points at it.  A backward pass inside a stream capture then makes that other stream wait for an
          // event of the capturing stream: it joins the capture, is never joined back, and hipStreamEndCapture crashes
          // (segmentation fault on this ROCm build).  Count the leaves whose live accumulator belongs to another stream.
          int64_t n = 0;
          for (const at::Tensor& p : params) {
            if (!p.defined() || !p.requires_grad() || !p.is_leaf() || !p.is_cuda()) continue;
            auto acc = torch::autograd::impl::try_get_grad_accumulator(p);
            if (!acc) continue;
            const auto st = acc->stream();
            if (st && *st != c10::Stream(c10::hip::getCurrentHIPStream(p.device().index()))) ++n;
          }
          return n;
        }, "leaves whose live AccumulateGrad node was created under another stream than the current one");
  m.def("set_exact_leaves", [](bool on) { const bool was = g_exact_leaves; g_exact_leaves = on; return was; },
        "True: every backward hands the engine one gradient per parameter leaf (the stock autograd path); False (default): "
        "plain loss.backward() passes write the leaves' .grad themselves (LeafScatter)");
  m.def("fast_leaf_passes", []() { return g_fast_leaf_passes; });
  m.def("timing_report", []() {
    std::string r;
    for (auto& kv : sections())
      r += kv.first + ": " + std::to_string(kv.second.total / (kv.second.n ? kv.second.n : 1)) + " us x " + std::to_string(kv.second.n) + "\n";
    return r;
  }, "CGVP_BRIDGE_TIMING=1: average host microseconds per instrumented section");
}
