/* caster_gvp.h -- C ABI of libcaster_gvp.so: the MI355X (gfx950) implementation
 * of CASTER-DTA's GVP / GINE message-passing hot path.
 *
 * The reference (stelleg/caster-dta) is 100 % Python and has no FFI; this is the
 * boundary a maintainer binds (ctypes stub in INTEGRATION.md) underneath the
 * reference's nn.Module API.  Each entry point names the reference code it
 * replaces (file:line under the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors);
 *     nothing is allocated, freed or retained; outputs are preallocated;
 *   - float tensors are contiguous fp32, row-major; nn.Linear weights are
 *     [out][in]; index tensors are int64 as PyG delivers them, CSR tables int32;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream);
 *     launches are asynchronous, no call synchronises;
 *   - return value: 0 = launched, >0 = hipError_t from the launch,
 *     <0 = CGVP_ERR_* argument error.  Re-entrant, no global state (except the opt-in DIAGNOSTICS at the end).
 *   - node state between stages is the merged row [s(16) | v(4x3)] = 28 floats
 *     per residue, exactly gvp_layers.py:101 `_merge` of the (16,4) hidden tuple.
 */
#ifndef CASTER_GVP_H
#define CASTER_GVP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGVP_ABI_VERSION 29
#define CGVP_ERR_BAD_ARG (-1)
#define CGVP_ERR_UNSUPPORTED_DIMS (-2)

/* PARAMETER ARENA.  All weights of VectorProteinGNN_LBAModel live in ONE
 * contiguous fp32 buffer, in the reference's state_dict order with the
 * zero-size dummy_params skipped (protein_gnn.py:325-358; key list in
 * the .pt checkpoint in pretrained_model_downstream/):
 *
 *   gvp_node.0 {wh.weight, ws.weight, ws.bias, wv.weight, wsv.weight, wsv.bias}
 *   gvp_node.1 {scalar_norm.weight, scalar_norm.bias}
 *   gvp_edge.0 {...}, gvp_edge.1 {...}
 *   for l in range(num_convs):
 *     conv_list.l.conv.message_func.{0,1,2} {...}
 *     conv_list.l.norm.{0,1}.scalar_norm {weight, bias}
 *     conv_list.l.ff_func.{0,1} {...}
 *   gvp_norm_before_scalar.scalar_norm {weight, bias}
 *   gvp_to_scalar {wh.weight, ws.weight, ws.bias}
 *
 * Every tensor is row-major nn.Linear layout [out][in].  ws.weight has
 * (num_types + si + h) input columns: one-hot type columns FIRST
 * (protein_gnn.py:139-152), then scalar features, then vector norms.
 * 15,117 floats for the shipped CASTER-DTA(2,2) model.  The host keeps the
 * nn.Parameters as views into this buffer, so optimizers update it in place and
 * no per-step packing exists.  Gradients use a second arena of the same layout. */
typedef struct {
  int32_t nt_node, nt_edge;   /* one-hot widths (num_ntypes, num_etypes)            */
  int32_t node_gvp, node_ln;  /* float offsets of gvp_node.0 / gvp_node.1           */
  int32_t edge_gvp, edge_ln;  /* gvp_edge.0 / gvp_edge.1                            */
  int32_t conv0;              /* conv_list.0 block                                  */
  int32_t conv_stride;        /* floats per conv_list.l block                       */
  int32_t ln_out, head;       /* gvp_norm_before_scalar / gvp_to_scalar             */
  int32_t total;              /* arena length in floats                             */
} cgvp_layout;

/* Encoder dimensions; the kernels are compiled for the CASTER-DTA(s,v)
 * configuration and return CGVP_ERR_UNSUPPORTED_DIMS for anything else. */
typedef struct {
  int32_t node_in_s, node_in_v;     /* 17, 3 */
  int32_t edge_in_s, edge_in_v;     /* 32, 1 */
  int32_t hidden_s, hidden_v;       /* 16, 4 */
  int32_t edge_hidden_s, edge_hidden_v; /* 32, 1 */
  int32_t out_s;                    /* 64 */
  int32_t storage;                  /* CGVP_F32 (0) or CGVP_BF16 (1): element type of the ACTIVATION buffers, see below */
  int32_t layer_kind;               /* CGVP_LAYER_*: which GVPConvLayer the conv / node-update entry points compute      */
} cgvp_dims;
/* LAYER KIND.  gvp_layers.GVPConvLayer(activations, vector_gate) (gvp_layers.py:340-366) comes in three forms in the
 * reference; message_func.{0,1} and ff_func.0 use the layer's arguments, message_func.2 and ff_func.1 always have
 * activations (None, None) (gvp_layers.py:279-288, :357-366):
 *   CGVP_LAYER_GATED   (relu, None), vector_gate=True : v * sigmoid(wsv(s))   protein_gnn.py:349-353 (CASTER-DTA / LBA)
 *   CGVP_LAYER_GVPDEF  (relu, sigmoid), no gate       : v * sigmoid(|v|)      protein_gnn.py:565-573 (CPD encoder/decoder)
 *   CGVP_LAYER_LINEAR  (None, None), no gate          : v                     protein_gnn.py:468-473 (PocketMiner-style)
 * The arena / image / gradient-block layout is the SAME for all three (the wsv slots of the two un-gated kinds are
 * ignored by the forward and receive zero gradient).  Kinds other than GATED are accepted by cgvp_lba_layout /
 * cgvp_lba_image_floats / cgvp_lba_prepare (layout only), cgvp_conv_fwd (stored edge embedding `e_in`, nt_edge = 0),
 * cgvp_node_update_fwd[_train] / cgvp_node_update_bwd (without the head), cgvp_conv_bwd, cgvp_bwd_workspace_floats
 * and cgvp_bwd_reduce -- fp32 storage only; every other entry point returns CGVP_ERR_UNSUPPORTED_DIMS for them. */
#define CGVP_LAYER_GATED 0
#define CGVP_LAYER_GVPDEF 1
#define CGVP_LAYER_LINEAR 2
/* ACTIVATION STORAGE ("bf16 storage / fp32 accumulate", BASELINE config 5; MFMA kernels only).  With
 * dims->storage == CGVP_BF16 every activation buffer of the protein entry points -- x_s, x_v, e_s, e_v, the node rows
 * h / dh / h_out, the edge-embedding store e_in / e_out / e_emb and the residue embeddings out -- holds bfloat16
 * (2-byte) elements with the same shapes and element strides, passed through the `float*` parameters below.  Loads
 * widen to fp32; all arithmetic, LayerNorm statistics, MFMA accumulation, the parameter arena / fragment image and
 * EVERY gradient buffer (g_*, grad_params, workspace) and dropout mask stay fp32; stores round to nearest even.
 * This halves the activation bytes of SURVEY 8(d)'s model: (300 + 112 L) N + (148 + 86 L) E per pass. */
#define CGVP_F32 0
#define CGVP_BF16 1

/* Destination-sorted CSR of a batched graph.  Replaces the gather/scatter
 * bookkeeping inside PyG MessagePassing.propagate (called at
 * gvp_layers.py:298-300; torch_geometric is third party, not vendored):
 * messages are reduced over edge_index[1].
 *   rowptr[N+1]  first sorted-edge position of every target node
 *   eperm[E]     original edge id of every sorted position (stable in edge id)
 *   esrc[E], edst[E]  source / target node of every sorted position
 *   work         per-target counters: N+1 ints rounded UP to a multiple of 64, 16-B aligned.  They must be zero when
 *                the kernels start: pass work_is_zero = 0 and the call zero-fills them first (one more launch), or
 *                keep a buffer that was zero-filled once and pass work_is_zero = 1 -- every call leaves the
 *                counters it used zeroed again.  work_is_zero = 2: the counters already hold the per-target counts
 *                (cgvp_lba_pass_begin produced them on this stream): the count launch is skipped.
 *   ids_scratch  E ints (edge ids in arrival order, consumed by the ranking launch)
 * Four launches on `stream` (count, scan, fill, rank; three after cgvp_lba_pass_begin), stable: a target's edges stay
 * in edge-id order.  Edges with an endpoint outside [0, N) are dropped; the unused positions at the end of the sorted
 * tables hold eperm = -1, esrc = edst = 0. */
int cgvp_csr_from_coo(const int64_t* edge_index, int64_t num_nodes, int64_t num_edges,
                      int32_t* rowptr, int32_t* eperm, int32_t* esrc, int32_t* edst,
                      int32_t* work, int32_t work_is_zero, int32_t* ids_scratch, void* stream);

/* The same tables for a batch assembled from graphs whose CSR is already known (the
 * reference's datasets reuse a few hundred unique protein / drug graphs across tens of
 * thousands of pairs, dataset/dual_dataset.py:123-125): a STORE holds the per-graph
 * tables back to back with LOCAL node / edge indices (graph g: st_node_off[g+1] -
 * st_node_off[g] nodes and as many + 1 row pointers at st_rowptr + st_node_off[g] + g;
 * its edges at st_edge_off[g]); batch slot b takes graph sel[b] and places it at node
 * offset b_node_off[b], edge offset b_edge_off[b] (all offset arrays are device int64
 * prefix sums).  Equivalent to cgvp_csr_from_coo on the concatenated edge_index
 * (PyG Batch.from_data_list order), in ONE small launch.
 * table_mode != 0 (feature-table wire format, dataset/dual_dataset.py:526-547 made unnecessary on the device): the
 * store ALSO keeps every unique graph's raw edge features with the edges already in dst-sorted order, back to back
 * ([sum of E_g][32], [sum of E_g][1][3], [sum of E_g] types); eperm then holds positions in THOSE tables, and the
 * encoder entry points are simply given the store's tables as e_s / e_v / etypes: a batch's edge features are never
 * copied or re-gathered, each graph's rows are read sequentially in place. */
int cgvp_csr_collate(const int32_t* st_rowptr, const int32_t* st_eperm, const int32_t* st_esrc,
                     const int32_t* st_edst, const int64_t* st_node_off, const int64_t* st_edge_off,
                     const int64_t* sel, const int64_t* b_node_off, const int64_t* b_edge_off,
                     int64_t batch_graphs, int32_t table_mode, int32_t* rowptr, int32_t* eperm, int32_t* esrc,
                     int32_t* edst, void* stream);

/* Fill `out` with the arena layout for the given one-hot widths and depth.
 * Host-only, no GPU work. */
int cgvp_lba_layout(const cgvp_dims* dims, int32_t num_ntypes, int32_t num_etypes,
                    int32_t num_convs, cgvp_layout* out);

/* FRAGMENT IMAGE.  The MFMA kernels do not read the arena directly: a prep
 * kernel re-lays every nn.Linear out as pre-permuted v_mfma_f32_16x16x4_f32
 * A-operand fragments plus bias / LayerNorm / type-column tables, one slice per
 * kernel (csrc/gvp_quad.h).  Rebuild it after every parameter update (one tiny
 * launch); inference can keep it.  `cgvp_lba_image_floats` returns its length
 * (<0 on error), `cgvp_lba_prepare` fills it on `stream`. */
int64_t cgvp_lba_image_floats(const cgvp_dims* dims, const cgvp_layout* layout);
int cgvp_lba_prepare(const cgvp_dims* dims, const cgvp_layout* layout, const float* params,
                     float* image, void* stream);

/* The three forward entry points below take BOTH the arena (`params`) and the
 * fragment image (`image`).  image != NULL selects the MFMA kernels (16 items per
 * wave tile, 4 lanes per item, the production path); image == NULL runs the
 * scalar one-item-per-lane kernels straight from the arena (kept as an
 * independent second implementation for cross-checks and A/B timing). */

/* First launch of a protein TRAINING pass: three independent pieces of work in ONE launch (each of them is launch
 * latency at CASTER-DTA batch sizes) --
 *   (1) gvp_node on all residues: exactly cgvp_node_embed_fwd (MFMA kernels), incl. the generator hand-off;
 *   (2) the fragment image of the current weights, written to `image` (exactly cgvp_lba_prepare); the embedding blocks
 *       take their slice straight from `params`, so (1) does not wait for (2);
 *   (3) optional (edge_index and csr_counters both non-NULL): the per-target edge counts of the CSR build into
 *       `csr_counters` (int32 [>= N + 1 rounded up to 64], all zero on entry) -- then call cgvp_csr_from_coo on the same
 *       stream with the same buffer as `work` and work_is_zero = 2 ("already counted").
 * Replaces protein_gnn.py:368-375 plus the per-step weight / batch bookkeeping in front of it. */
int cgvp_lba_pass_begin(const cgvp_dims* dims, const cgvp_layout* layout, const float* params, float* image,
                        const float* x_s, const float* x_v, const int64_t* ntypes, int64_t num_nodes, float* h,
                        uint64_t* rng_state, uint64_t* rng_out, const int64_t* edge_index, int64_t num_edges,
                        int32_t* csr_counters, void* stream);

/* gvp_node = Sequential(GVP, LayerNorm) on one-hot(ntypes) ++ x_s, x_v
 * (protein_gnn.py:368-375).  x_s [N][17], x_v [N][3][3], ntypes [N] -> h [N][28]. */
/* rng_state / rng_out (both or neither; MFMA kernels): as the FIRST kernel of an encoder pass this call can advance the
 * persistent dropout generator without a launch of its own -- see cgvp_rng_next. */
int cgvp_node_embed_fwd(const cgvp_dims* dims, const cgvp_layout* layout, const float* params,
                        const float* image, const float* x_s, const float* x_v,
                        const int64_t* ntypes, int64_t num_nodes, float* h, uint64_t* rng_state,
                        uint64_t* rng_out, void* stream);

/* GVPConv.forward (gvp_layers.py:291-308) of conv layer `layer` for every edge
 * plus the reduction over target nodes (aggr 'sum'/'add' or 'mean'), with
 * gvp_edge (protein_gnn.py:376) fused in.
 * h [N][28]; e_s [E][32], e_v [E][1][3], etypes [E] in ORIGINAL edge order; CSR
 * tables from cgvp_csr_from_coo.  -> dh [N][28] (every row written, zero for
 * isolated nodes).
 * EDGE EMBEDDING STORE (MFMA kernels; both optional).  gvp_edge + LayerNorm does not depend on the conv layer.
 * With e_out != NULL the call also writes it, in SORTED-edge order (the CSR position p, not the original edge
 * id): CGVP_EDGE_ROW floats per edge = [e_s 32 | e_v 3 | pad].  With e_in != NULL the call reads that store
 * (the buffer must hold num_edges + 1 rows: row num_edges is a spare that masked-off lanes write zeros to)
 * sequentially instead of gathering and re-embedding the raw features (e_s / e_v / etypes / eperm are then
 * unused and may be NULL).  The host code lets layer 0 write it and every later layer -- and every conv
 * backward -- read it: this is exactly SURVEY 8(d)'s byte model (embed writes 140 B/edge, each conv reads them). */
#define CGVP_EDGE_ROW 36
int cgvp_conv_fwd(const cgvp_dims* dims, const cgvp_layout* layout, const float* params,
                  const float* image, int32_t layer, const float* h, const float* e_s,
                  const float* e_v, const int64_t* etypes, const int32_t* rowptr,
                  const int32_t* eperm, const int32_t* esrc, const int32_t* edst,
                  int64_t num_nodes, int64_t num_edges, int32_t aggr_mean, const float* e_in,
                  float* e_out, float* dh, void* stream);

/* Rest of GVPConvLayer.forward in eval mode (gvp_layers.py:407-410):
 * h_out = LN1(y + FF(y)), y = LN0(h + dh).  When `with_head` != 0 also applies
 * gvp_norm_before_scalar + gvp_to_scalar (protein_gnn.py:385-386) and writes
 * out [N][64]; h_out may then be NULL. */
int cgvp_node_update_fwd(const cgvp_dims* dims, const cgvp_layout* layout, const float* params,
                         const float* image, int32_t layer, const float* h, const float* dh,
                         int64_t num_nodes, int32_t with_head, float* h_out, float* out,
                         void* stream);

/* IN-KERNEL DROPOUT (training).  gvp_layers.Dropout (gvp_layers.py:187-219) multiplies by Bernoulli(1-p)/(1-p)
 * factors, one per (node, scalar channel) and one per (node, vector channel) shared by its xyz.  The kernels
 * generate them with a counter-based generator (Philox4x32-10, csrc/gvp_rng.h) keyed by
 * (seed, offset, stream, node, channel), so the backward pass regenerates exactly the factors of the forward
 * pass and no mask ever exists in HBM.  `seed` points at TWO uint64 on the device {seed, offset} (so a
 * HIP-graph replay sees fresh values when the host refreshes them on the stream); `stream` is the id of the
 * first mask of the call: conv layer l passes 2*l (its dropout[0] uses stream, dropout[1] stream + 1), GINE
 * layer l passes l.  rng == NULL or rng->seed == NULL: no dropout.  Explicit mask pointers, where an entry
 * point has them, take precedence (tests pin masks that way). */
typedef struct {
  const uint64_t* seed;   /* device pointer to {seed, offset}            */
  float p;                /* drop probability, 0 <= p < 1                */
  int32_t stream;         /* id of the first mask this call draws        */
} cgvp_rng;

/* Advance a persistent generator state: state = {seed, offset} (device, 2 x uint64) gets offset + 1 and the new pair
 * is copied to `out` (device, 2 x uint64), which is what this pass's cgvp_rng.seed points at -- so a forward, its
 * backward and a HIP-graph replay of both agree on the factors while successive passes differ.  One 1-thread launch;
 * cgvp_node_embed_fwd does the same inside its own launch. */
int cgvp_rng_next(uint64_t* state, uint64_t* out, void* stream);

/* The factors the kernels would apply, written out for inspection (statistics tests; checking a training
 * step against the CPU oracle run with the same masks): out [num_masks][N][width], mask m of the call =
 * stream rng->stream + m.  width 20 = the protein row [16 scalar | 4 vector-channel factors]; any other
 * multiple of 4 = the GINE row of that many channels. */
int cgvp_dropout_masks(const cgvp_rng* rng, int32_t num_masks, int64_t num_nodes, int32_t width,
                       float* out, void* stream);

/* ONE launch for a whole GVPConvLayer.forward (gvp_layers.py:400-415), MFMA kernels
 * only: cgvp_conv_fwd followed, on each wave's own target nodes and straight from
 * its LDS rows, by cgvp_node_update_fwd[_train].  `dh` is optional (the aggregated
 * messages, needed by the backward pass only); mask0 / mask1 as in
 * cgvp_node_update_fwd_train (NULL in eval mode); with_head as in
 * cgvp_node_update_fwd (h_out then optional).  Pays off when a conv wave owns many
 * targets (about 30 / average in-degree of them share one 16-lane node tile): the
 * host code uses it for num_edges <= 4 * num_nodes and the two-launch form otherwise. */
int cgvp_conv_layer_fwd(const cgvp_dims* dims, const cgvp_layout* layout, const float* image,
                        int32_t layer, const float* h, const float* e_s, const float* e_v,
                        const int64_t* etypes, const int32_t* rowptr, const int32_t* eperm,
                        const int32_t* esrc, const int32_t* edst, int64_t num_nodes,
                        int64_t num_edges, int32_t aggr_mean, const float* mask0,
                        const float* mask1, const cgvp_rng* rng, int32_t with_head, const float* e_in,
                        float* e_out, float* dh, float* h_out, float* out, void* stream);

/* Training-mode variant of cgvp_node_update_fwd (MFMA kernels only): `mask0` /
 * `mask1` are the dropout masks of gvp_layers.Dropout (gvp_layers.py:187-219) for
 * dropout[0] (on dh) and dropout[1] (on the feed-forward output), one row
 * [16 scalar factors | 4 vector-channel factors] per node, each 0 or 1/(1-p);
 * either may be NULL (= no dropout). */
int cgvp_node_update_fwd_train(const cgvp_dims* dims, const cgvp_layout* layout, const float* image,
                               int32_t layer, const float* h, const float* dh, const float* mask0,
                               const float* mask1, const cgvp_rng* rng, int64_t num_nodes,
                               int32_t with_head, float* h_out, float* out, void* stream);

/* ---------------------------------------------------------------- BACKWARD
 * Autograd of the three forward stages (what torch.autograd derives from
 * gvp_layers.py / protein_gnn.py in the reference).  Each call recomputes its
 * stage from the saved stage INPUTS (h_l, dh_l, raw features) and
 *   - writes data gradients to the given buffers,
 *   - ADDS weight gradients into `grad_params`, an arena of the parameter layout
 *     (zero it once per backward pass).  `workspace` is scratch of
 *     cgvp_bwd_workspace_floats() floats (per-workgroup partial blocks, summed in a
 *     fixed order by a reduce kernel: weight gradients are run-to-run reproducible).
 * MFMA kernels only (image required). */
int64_t cgvp_bwd_workspace_floats(const cgvp_dims* dims, const cgvp_layout* layout);

/* DEFERRED REDUCTION.  By default every backward entry point reduces its own
 * partial blocks.  Pass a non-NULL `segs` (room for CGVP_MAX_SEGS_PER_CALL
 * entries) and `nsegs` to skip that: the call then only describes WHERE its
 * partials sit inside `workspace` (which must stay untouched -- give each call its
 * own region of cgvp_bwd_workspace_floats() floats), and ONE
 * cgvp_bwd_reduce(all segments) at the end of the backward pass adds everything
 * into grad_params (same fixed summation order, one launch instead of ten). */
typedef struct {
  const float* slab;   /* first partial row                                  */
  int32_t rows;        /* number of partial rows                             */
  int32_t stride;      /* floats between rows                                */
  int32_t col0, len;   /* columns [col0, col0+len) of every row ...          */
  int32_t dst;         /* ... are summed into grad_params[dst, dst+len)      */
} cgvp_segment;
#define CGVP_MAX_SEGS_PER_CALL 2
#define CGVP_MAX_SEGS 32
/* overwrite != 0: STORE the sums instead of adding them (for segment lists whose destinations are disjoint and cover
 * every element the caller reads: grad_params then needs no zero-fill launch). */
int cgvp_bwd_reduce(const cgvp_segment* segs, int32_t nsegs, float* grad_params, int32_t overwrite, void* stream);

/* d/d(h, dh, weights) of cgvp_node_update_fwd[_train].  Upstream gradient: with
 * the head, g_out [N][64] together with h_out [N][28], the h_out the forward call
 * wrote (the head's input; pass a buffer to the forward even when with_head);
 * otherwise the SUM of up to three [N][28] buffers g_up0..2 (NULL entries
 * skipped; h_out / g_out ignored).  Writes g_dh [N][28] (= mask0 * d h) and, when
 * g_h != NULL, g_h [N][28] (the residual path; equals g_dh without dropout).
 * With the head this is two launches (head, then the layer).  `zero_out` (optional,
 * [N][28]) is filled with zeros on the way: pass the g_src of the cgvp_conv_bwd call
 * that follows and set its g_src_zeroed to save that call's memset launch. */
int cgvp_node_update_bwd(const cgvp_dims* dims, const cgvp_layout* layout, const float* image,
                         int32_t layer, const float* h, const float* dh, const float* mask0,
                         const float* mask1, const cgvp_rng* rng, const float* h_out, const float* g_out,
                         const float* g_up0, const float* g_up1, const float* g_up2,
                         int64_t num_nodes, int32_t with_head, float* g_dh, float* g_h,
                         float* zero_out, float* grad_params, float* workspace, cgvp_segment* segs,
                         int32_t* nsegs, void* stream);

/* d/d(h, edge embedding, message weights) of cgvp_conv_fwd given g_dh = d(loss)/d(dh).  `e_emb` is the edge
 * embedding store the forward wrote ([E][CGVP_EDGE_ROW], sorted-edge order).  Gradients w.r.t. the node rows
 * arrive in two buffers that the consumer sums: g_src [N][28] (scatter over the unsorted sources; float atomics;
 * zeroed by this call unless g_src_zeroed != 0) and g_dst [N][28] (segmented sums over the sorted targets; every
 * row written).  g_e [E][CGVP_EDGE_ROW] receives this layer's d(edge embedding) (plain stores, every row, in the
 * storage type of dims->storage like e_emb -- the only gradient buffer that is not fp32 under CGVP_BF16: it is the
 * largest one and cgvp_edge_embed_bwd rounds it to bf16 matrix operands anyway); the gradients of gvp_edge's own
 * weights come from ONE cgvp_edge_embed_bwd over all layers' g_e. */
int cgvp_conv_bwd(const cgvp_dims* dims, const cgvp_layout* layout, const float* image, int32_t layer,
                  const float* h, const float* e_emb, const int32_t* rowptr, const int32_t* esrc,
                  const int32_t* edst, int64_t num_nodes, int64_t num_edges, int32_t aggr_mean,
                  const float* g_dh, float* g_src, int32_t g_src_zeroed, float* g_dst, float* g_e,
                  float* grad_params, float* workspace, cgvp_segment* segs, int32_t* nsegs,
                  void* stream);

/* Backward of gvp_edge = Sequential(GVP, LayerNorm) (protein_gnn.py:331-335, :376), once per step: the upstream
 * gradient of edge p is the SUM over g_e[0 .. num_g) of row p (the buffers the conv backward calls wrote).
 * gvp_edge.0 / gvp_edge.1 weight gradients are ADDED into grad_params (or described in `segs` for the deferred
 * reduction).  g_e_s [E][32] / g_e_v [E][1][3] (fp32, ORIGINAL edge order; both or neither): when given, also the
 * gradient w.r.t. the raw edge features (what autograd returns for `eattr` in the reference); rows of edges the CSR
 * build dropped are zero (two fill launches in front).  `g_e` is a HOST array of device pointers. */
int cgvp_edge_embed_bwd(const cgvp_dims* dims, const cgvp_layout* layout, const float* image,
                        const float* e_s, const float* e_v, const int64_t* etypes, const int32_t* eperm,
                        int64_t num_edges, const float* const* g_e, int32_t num_g, float* g_e_s, float* g_e_v,
                        float* grad_params, float* workspace, cgvp_segment* segs, int32_t* nsegs, void* stream);

/* d/d(x_s, x_v, weights) of cgvp_node_embed_fwd; upstream = sum of g_up0..2.
 * g_x_s [N][17] / g_x_v [N][3][3] may both be NULL (inputs without gradient). */
int cgvp_node_embed_bwd(const cgvp_dims* dims, const cgvp_layout* layout, const float* image,
                        const float* x_s, const float* x_v, const int64_t* ntypes,
                        int64_t num_nodes, const float* g_up0, const float* g_up1,
                        const float* g_up2, float* g_x_s, float* g_x_v, float* grad_params,
                        float* workspace, cgvp_segment* segs, int32_t* nsegs, void* stream);

/* One GINEConv + activation of HomoMoleculeGNN_GINE (molecule_gnn.py:254-268,
 * :271-280; PyG GINEConv / MLP restated):
 *   x'_i = LeakyReLU_slope( Lin1( LeakyReLU_slope( Lin0( (1+eps) x_i +
 *            sum_{j->i} ReLU(x_j + W_e [onehot(etype) ++ eattr] + b_e) ) ) ) )
 * `x` is [N][cin] when `ntypes` is NULL; for the first layer pass the raw atom
 * features [N][cin - num_ntypes] and `ntypes`, and the one-hot columns are
 * synthesised on the fly (molecule_gnn.py:127-140).  cin <= 64, chid <= 64,
 * cout <= 64, edge_dim + num_etypes <= 16.
 * variant 0 (production): 16-atom MFMA tiles (csrc/gine_quad_kernels.hip) for the
 * compiled CASTER-DTA layer shapes -- needs `edst` as well -- and the generic kernel for
 * any other shape; variant 1: always the generic one-wave-per-atom kernel (`edst` may
 * be NULL), kept as the independent second implementation for cross-checks. */
typedef struct {
  const float* eps;     /* [1]                conv_list.l.eps            */
  const float* we;      /* [cin][net+edge_dim] conv_list.l.lin.weight     */
  const float* be;      /* [cin]              conv_list.l.lin.bias       */
  const float* w0;      /* [chid][cin]        conv_list.l.nn.lins.0.weight */
  const float* b0;      /* [chid]                                         */
  const float* w1;      /* [cout][chid]       conv_list.l.nn.lins.1.weight */
  const float* b1;      /* [cout]                                         */
} cgvp_gine_w;

int cgvp_gine_conv_fwd(const float* x, const int64_t* ntypes, int32_t num_ntypes,
                       const float* eattr, const int64_t* etypes, int32_t num_etypes,
                       int32_t edge_dim, const int32_t* rowptr, const int32_t* eperm,
                       const int32_t* esrc, const int32_t* edst, int64_t num_nodes,
                       int64_t num_edges, int32_t cin, int32_t chid, int32_t cout,
                       const cgvp_gine_w* w, float act_slope, const float* mask, const cgvp_rng* rng,
                       int32_t variant, float* out, void* stream);

/* Backward of cgvp_gine_conv_fwd.  `mask` (optional, [N][cout]) is the dropout
 * mask molecule_gnn.py:262 applies to the layer output during training (also an
 * optional argument of the forward).  g_out [N][cout] -> g_x [N][cin - num_ntypes]
 * (NULL to skip; zeroed here, filled with float atomics) and the layer's weight
 * gradients ADDED into `grad_layer` in state_dict order
 *   eps | nn.lins.0.weight | nn.lins.0.bias | nn.lins.1.weight | nn.lins.1.bias | lin.weight | lin.bias.
 * Takes all four CSR tables of cgvp_csr_from_coo (the forward needs no edst).
 * Runs on 16-atom MFMA tiles (csrc/gine_quad_kernels.hip).  Compiled for the
 * CASTER-DTA layer shapes (52,16,16), (16,64,64), (16,16,16) with 11 / 0 atom
 * types, 5 bond types, 9 bond features. */
int64_t cgvp_gine_bwd_workspace_floats(void);
/* `max_workgroups`: cap on the workgroups (= CUs) the backward may occupy; <= 0 selects the
 * default of 16: inside CASTER-DTA the drug backward runs beside the protein backward, whose
 * kernels own 240 of the 256 CUs.  A caller that trains the molecule encoder alone passes up
 * to 256.  (A call argument, not library state: the library keeps no global state.) */
int cgvp_gine_conv_bwd(const float* x, const int64_t* ntypes, int32_t num_ntypes, const float* eattr,
                       const int64_t* etypes, int32_t num_etypes, int32_t edge_dim,
                       const int32_t* rowptr, const int32_t* eperm, const int32_t* esrc,
                       const int32_t* edst, int64_t num_nodes, int64_t num_edges, int32_t cin,
                       int32_t chid, int32_t cout, const cgvp_gine_w* w, float act_slope,
                       const float* mask, const cgvp_rng* rng, const float* g_out, float* g_x,
                       float* grad_layer, float* workspace, int32_t max_workgroups, void* stream);

/* ------------------------------------------------------------ EDGE FEATURISATION (opt-in)
 * The protein edge features of utils/create_protein_features.py:225-273 (+ calc_pos_encoding :368-386) computed on the
 * device from C-alpha coordinates and sequence indices, for the edges i -> j of edge_index ([2][E], row 0 = i):
 *   e_s[e][ 0:16] = exp(-((|CA_i - CA_j| - mu_k) / 1.25)^2), mu_k = linspace(0, 20, 16)
 *   e_s[e][16:24] = cos((seq_j - seq_i) f_k), e_s[e][24:32] = sin(...), f_k = exp(-2 k ln(10000) / 8), k = 0..7
 *   e_v[e]        = (CA_i - CA_j) / |CA_i - CA_j|, zero for coincident positions (self loops)
 * ca_xyz [N][3] fp32 (Angstrom), seq_index [N] int64 (position of the residue in its own chain), outputs fp32 in
 * ORIGINAL edge order: exactly the eattr tensors the encoder entry points take.  A dataset can then keep 12 B per
 * residue + the edge list instead of 140 B per edge. */
int cgvp_edge_featurise(const float* ca_xyz, const int64_t* seq_index, const int64_t* edge_index,
                        int64_t num_nodes, int64_t num_edges, float* e_s, float* e_v, void* stream);

/* ------------------------------------------------------------ CROSS-ATTENTION CORE
 * The part of nn.MultiheadAttention between its input and output projections, for the residue <-> atom cross
 * attention of CrossAttentionModule (joint_gnn.py:321-409; called on to_dense_batch-padded tensors at
 * joint_gnn.py:206-215 in the reference), on COMPACT row arrays with PyG ptr offsets instead of padded batches:
 * for every pair b and head h
 *     out[q] = softmax_k( scale * q[q] . k[k] ) v[k],   q in rows q_ptr[b] .. q_ptr[b+1], k in k_ptr[b] .. k_ptr[b+1]
 * head_dim is 16 (embed 128 / 8 heads in CASTER-DTA; embed = 16 * heads in general); q / k / v / out are
 * [rows][16 * heads] fp32 row-major (the projected tensors), lse [num_q][heads] is the log-sum-exp the backward
 * needs.  A pair without keys yields zero rows.  One call takes up to TWO problems (the module's two directions run
 * in one launch).  No attention dropout (the reference trains with attention_dropout = 0, train_model.py:317). */
typedef struct {
  const float* q; const float* k; const float* v;
  const int64_t* q_ptr; const int64_t* k_ptr;   /* [num_pairs + 1] row offsets                            */
  int64_t num_q, num_k;                          /* total rows of q and of k / v                           */
  float* out; float* lse;                        /* forward outputs (inputs of the backward / weights)     */
  const float* g_out;                            /* backward: d out [num_q][16 * heads]                    */
  float* delta;                                  /* backward scratch [num_q][heads]                        */
  float* g_q; float* g_k; float* g_v;            /* backward outputs, every row written                    */
  float* weights; int64_t weights_lq, weights_lk; /* cgvp_attn_weights: dense [num_pairs][lq][lk], zero-filled by the caller */
} cgvp_attn_problem;
int cgvp_attn_fwd(const cgvp_attn_problem* problems, int32_t num_problems, int64_t num_pairs, int32_t heads,
                  float scale, void* stream);
/* d q, d k, d v from d out (two launches, no atomics: run-to-run reproducible). */
int cgvp_attn_bwd(const cgvp_attn_problem* problems, int32_t num_problems, int64_t num_pairs, int32_t heads,
                  float scale, void* stream);
/* nn.MultiheadAttention's returned weights (need_weights=True, averaged over heads) in the reference's dense
 * layout [num_pairs][weights_lq][weights_lk] (what inference/evaluation.py:43-66 slices per pair); needs the lse
 * of cgvp_attn_fwd.  Inference only. */
int cgvp_attn_weights(const cgvp_attn_problem* problems, int32_t num_problems, int64_t num_pairs, int32_t heads,
                      float scale, void* stream);

/* ------------------------------------------------------------ WHOLE-PASS ENTRY POINTS (the production path)
 * ONE call per encoder pass: each of the four functions below issues the complete launch sequence of
 * VectorProteinGNN_LBAModel.forward (protein_gnn.py:361-388) / its autograd, resp. HomoMoleculeGNN_GINE.forward
 * (molecule_gnn.py:254-268) / its autograd, on `stream`.  This is how the training loop of the reference
 * (train_model.py:548-587: a NEW batch -- new N, E, edge_index -- every step) drives the library: per step 4 host
 * calls and 4 caller-allocated buffers instead of ~25 launches' worth of argument marshalling and ~30 small tensors.
 * Everything a pass produces for its backward lives in ONE caller-allocated forward workspace whose sub-buffer byte
 * offsets the *_fwd_workspace functions report (all offsets are multiples of 256); the backward passes read it and
 * use a second scratch workspace of *_bwd_workspace_bytes.  The library keeps no state and allocates nothing: both
 * workspaces, the dropout generator state (`rng_state`, 2 x uint64, persistent across passes) and the CSR counters
 * (`csr_counters`, int32 [>= N + 1 rounded up to 64], zero on entry, left zero on exit, persistent; one per stream) are
 * the caller's, so a caller that captures a step into a HIP graph decides their lifetime (hand out graph-owned or
 * never-freed buffers).  The fine-grained entry points above remain (tests of single stages, A/B timing). */
typedef struct {
  int64_t num_nodes, num_edges;
  const float* x_s;  const float* x_v;  const int64_t* ntypes;     /* [N][17], [N][3][3], [N] (NULL when num_ntypes == 0)   */
  const float* e_s;  const float* e_v;  const int64_t* etypes;     /* [E or table rows][32], [..][1][3], [..]                */
  const int64_t* edge_index;   /* [2][E] as PyG delivers it; may be NULL when the four tables below are given               */
  const int32_t* rowptr; const int32_t* eperm; const int32_t* esrc; const int32_t* edst;  /* prebuilt CSR (cgvp_csr_collate) or all NULL */
} cgvp_lba_batch;

typedef struct {              /* byte offsets inside the forward workspace                                                   */
  int64_t seed;               /* uint64[2] {seed, offset} of this pass's dropout (written by its first kernel)               */
  int64_t image;              /* fragment image of the weights the pass ran with (float[cgvp_lba_image_floats])              */
  int64_t state;              /* [state_rows][node_stride][28] activations: h_0..h_{L-1}, dh_0..dh_{L-1}, head input          */
  int64_t e_emb;              /* [E + 1][CGVP_EDGE_ROW] edge embedding store, sorted-edge order                               */
  int64_t rowptr, eperm, esrc, edst, ids_scratch;   /* CSR tables int32 [N+1], [E], [E], [E], [E] (built unless the batch brings them) */
  int64_t total;              /* bytes to allocate                                                                            */
  int64_t state_rows, node_stride;
} cgvp_lba_fwd_ws;
/* save_state != 0: training pass (everything the backward needs is kept); 0: inference (ping-pong rows only). */
int cgvp_lba_fwd_workspace(const cgvp_dims* dims, const cgvp_layout* layout, int64_t num_nodes, int64_t num_edges,
                           int32_t save_state, cgvp_lba_fwd_ws* out);

/* flags for cgvp_lba_forward_pass */
#define CGVP_PASS_UNFUSED 1   /* never use the one-launch-per-layer kernel (A/B timing; the default picks it for E <= 4 N) */
/* dropout: dropout_p > 0 selects training-mode dropout (gvp_layers.py:187-219): from explicit `masks` ([2 L][N][20],
 * test hook) when given, else drawn in-kernel from `rng_state` (required then), which the pass advances.
 * out: [N][64] residue embeddings (element type = dims->storage). */
int cgvp_lba_forward_pass(const cgvp_dims* dims, const cgvp_layout* layout, const float* params,
                          const cgvp_lba_batch* batch, int32_t aggr_mean, float dropout_p, uint64_t* rng_state,
                          const float* masks, int32_t* csr_counters, void* workspace, int32_t save_state,
                          int32_t flags, float* out, void* stream);

/* The launch plan cgvp_lba_forward_pass follows for these sizes (host-only, no GPU work): returns the number of
 * kernel launches of the pass and sets *fused_layers to 1 when every GVPConvLayer is ONE launch (conv + node update
 * (+ head); chosen for E <= 4 N unless CGVP_PASS_UNFUSED), 0 when it is two. */
int cgvp_lba_forward_plan(int64_t num_nodes, int64_t num_edges, int32_t num_convs, int32_t prebuilt_csr, int32_t flags,
                          int32_t* fused_layers);

int64_t cgvp_lba_bwd_workspace_bytes(const cgvp_dims* dims, const cgvp_layout* layout, int64_t num_nodes,
                                     int64_t num_edges);
/* Autograd of cgvp_lba_forward_pass(save_state = 1): g_out [N][64] fp32 -> grad_params (arena layout, every element
 * STORED: no zero fill needed) and, when both of a pair are non-NULL, the gradients w.r.t. the raw features: g_x_s
 * [N][17] / g_x_v [N][3][3] and g_e_s [E][32] / g_e_v [E][1][3] (original edge order).  `batch`, aggr_mean,
 * dropout_p and masks must be the forward's; `fwd_workspace` is the buffer the forward filled (read only). */
int cgvp_lba_backward_pass(const cgvp_dims* dims, const cgvp_layout* layout, const cgvp_lba_batch* batch,
                           int32_t aggr_mean, float dropout_p, const float* masks, const void* fwd_workspace,
                           const float* g_out, void* bwd_workspace, float* grad_params, float* g_x_s, float* g_x_v,
                           float* g_e_s, float* g_e_v, void* stream);

#define CGVP_GINE_MAX_LAYERS 8
typedef struct {
  int32_t num_layers;                          /* L = num_convs                                                       */
  int32_t widths[CGVP_GINE_MAX_LAYERS + 1];    /* channel widths: widths[0] = in + num_ntypes (52), ..., widths[L] = out */
  int32_t num_ntypes, num_etypes, edge_dim;    /* 11, 5, 9                                                             */
  float act_slope;                             /* LeakyReLU slope (0 = ReLU, 1 = identity)                              */
} cgvp_gine_cfg;
typedef struct {
  int64_t num_nodes, num_edges;
  const float* x; const int64_t* ntypes;       /* [N][widths[0] - num_ntypes], [N]                                     */
  const float* eattr; const int64_t* etypes;   /* [E][edge_dim], [E]                                                   */
  const int64_t* edge_index;
  const int32_t* rowptr; const int32_t* eperm; const int32_t* esrc; const int32_t* edst;
} cgvp_gine_batch;
typedef struct {
  int64_t seed;                                /* uint64[2]                                                            */
  int64_t hidden[CGVP_GINE_MAX_LAYERS];        /* hidden[l]: output of layer l, l < L - 1 ([N][widths[l + 1]] fp32)     */
  /* training passes only (save_state): what the backward of layer l would otherwise recompute from the gathered source
   * rows -- agg[l]: the aggregated messages sum_j relu(x_j + W_e e_ji + b) per atom, [N][ceil16(widths[l])] fp32;
   * pos[l]: the ReLU pattern of every message, 4 x uint16 per SORTED edge position (bit 4 mt + r of word g <-> channel
   * 16 mt + 4 g + r); 0 when not saved */
  int64_t agg[CGVP_GINE_MAX_LAYERS];
  int64_t pos[CGVP_GINE_MAX_LAYERS];
  int64_t rowptr, eperm, esrc, edst;
  int64_t saved;                               /* the backward pass reads bytes [0, saved) only                        */
  int64_t ids_scratch;                         /* forward-only scratch behind them                                      */
  int64_t total;
} cgvp_gine_fwd_ws;
int cgvp_gine_fwd_workspace(const cgvp_gine_cfg* cfg, int64_t num_nodes, int64_t num_edges, int32_t save_state,
                            cgvp_gine_fwd_ws* out);
/* w: HOST array of num_layers cgvp_gine_w; masks: NULL or HOST array of num_layers - 1 device pointers ([N][widths[l+1]],
 * explicit inter-layer dropout factors, test hook); otherwise dropout_p > 0 draws in-kernel from rng_state. */
int cgvp_gine_forward_pass(const cgvp_gine_cfg* cfg, const cgvp_gine_w* w, const cgvp_gine_batch* batch,
                           float dropout_p, uint64_t* rng_state, const float* const* masks, int32_t* csr_counters,
                           void* workspace, int32_t save_state, int32_t variant, float* out, void* stream);
int64_t cgvp_gine_bwd_workspace_bytes(const cgvp_gine_cfg* cfg, int64_t num_nodes, int64_t num_edges);
/* g_out [N][widths[L]] -> grad_flat: the layers' weight gradients back to back, each in state_dict order
 * (eps | nn.lins.0.weight | nn.lins.0.bias | nn.lins.1.weight | nn.lins.1.bias | lin.weight | lin.bias), every element
 * STORED; g_x [N][widths[0] - num_ntypes] or NULL. */
int cgvp_gine_backward_pass(const cgvp_gine_cfg* cfg, const cgvp_gine_w* w, const cgvp_gine_batch* batch,
                            float dropout_p, const float* const* masks, const void* fwd_workspace, const float* g_out,
                            void* bwd_workspace, float* grad_flat, float* g_x, int32_t max_workgroups, void* stream);

/* ------------------------------------------------------------ ROW-WISE LINEAR LAYERS OF THE HEAD: weight gradients
 * The joint head applies nn.Linear layers to every residue / atom row (joint_gnn.py:188-198, :376-389).  Their weight and
 * bias gradients reduce over ALL rows of the batch into a tiny output:
 *     out[o * I + i] = sum_r gy[r][o] x[r][i]   (o < O, i < I)        out[O * I + o] = sum_r gy[r][o]
 * x [R][I], gy [R][O] fp32 row-major; out [O * I + O] (every element stored); workspace of
 * cgvp_linear_wgrad_workspace_floats(R, I, O) floats (per-split partial sums; < 0 = unsupported shape).
 * I a multiple of 16 up to 256, O a multiple of 64.  Two launches (split-row MFMA kernel, fixed-order reduce):
 * deterministic, exact fp32. */
/* Row-wise nn.LayerNorm of the joint head (joint_gnn.py:376-389, the pre-attention and feed-forward norms on compact
 * residue / atom rows): y = (x - mean) * rstd * gamma + beta per row of `dim` floats (dim in {64, 128, 256, 512}), biased
 * variance, rstd = 1 / sqrt(var + eps); mean / rstd [rows] are saved for the backward.  gamma / beta may be NULL
 * (elementwise_affine=False).  Backward: gx [rows][dim], g_gamma_beta [2 * dim] = [d gamma | d beta] (STORED; per-workgroup
 * partials in `workspace` of cgvp_layer_norm_bwd_workspace_floats() floats are added in a fixed order). */
int cgvp_layer_norm_fwd(const float* x, const float* gamma, const float* beta, int64_t rows, int32_t dim, float eps,
                        float* y, float* mean, float* rstd, void* stream);
int64_t cgvp_layer_norm_bwd_workspace_floats(int64_t rows, int32_t dim);
int cgvp_layer_norm_bwd(const float* gy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        int64_t rows, int32_t dim, float* gx, float* workspace, float* g_gamma_beta, void* stream);

/* The dropout sites of the joint head fused with their neighbours (joint_gnn.py:188-198, :376-389), on compact fp32
 * rows [rows][dim] (dim a multiple of 8, 16-B aligned buffers):
 *   cgvp_dropout_add      y = x + a * f   (x may be NULL: y = a * f)     residual + nn.Dropout
 *   cgvp_dropout_scale    out = g * f                                    its backward w.r.t. a (d x = g)
 *   cgvp_act_dropout_fwd  y = LeakyReLU_slope(t) * f  (slope 0: ReLU)    activation + nn.Dropout
 *   cgvp_act_dropout_bwd  gt = g * f * (y > 0 ? 1 : slope)               from the saved OUTPUT y
 * f = inverted-dropout factor of (row, column) regenerated from `rng` ({seed, offset} pair on the device, p, stream id =
 * the call site; rng NULL or p = 0: f = 1); same generator as the encoders' in-kernel dropout (cgvp_rng above). */
int cgvp_dropout_add(const float* a, const float* x, const cgvp_rng* rng, int64_t rows, int32_t dim, float* y, void* stream);
int cgvp_dropout_scale(const float* g, const cgvp_rng* rng, int64_t rows, int32_t dim, float* out, void* stream);
int cgvp_act_dropout_fwd(const float* t, const cgvp_rng* rng, float slope, int64_t rows, int32_t dim, float* y, void* stream);
int cgvp_act_dropout_bwd(const float* g, const float* y, const cgvp_rng* rng, float slope, int64_t rows, int32_t dim, float* gt,
                         void* stream);

int64_t cgvp_linear_wgrad_workspace_floats(int64_t num_rows, int32_t in_features, int32_t out_features);
int cgvp_linear_wgrad(const float* x, const float* gy, int64_t num_rows, int32_t in_features, int32_t out_features,
                      float* workspace, float* out, void* stream);

/* ------------------------------------------------------------ BATCH STAGING (shape-bucketed HIP-graph replay)
 * A training loop that replays a captured step on a DIFFERENT batch every step (train_model.py:548-587 feeds new N, E
 * each step) keeps static input buffers per shape bucket and copies the batch into them.  This does every copy of a
 * step in ONE launch: item k copies `copy_bytes` from src to dst and fills the rest of dst's `capacity_bytes` with the
 * 32-bit pattern `fill_word` (0xFFFFFFFF = index -1 for the padded tail of edge_index: the CSR build drops such edges;
 * 0 for features and upstream gradients: padded nodes are isolated and receive no gradient).  All sizes are multiples
 * of 4, pointers 4-byte aligned; src may be NULL when copy_bytes == 0.  Up to CGVP_MAX_STAGE items per call. */
#define CGVP_MAX_STAGE 24
typedef struct {
  void* dst; const void* src;
  int64_t copy_bytes, capacity_bytes;
  uint32_t fill_word;
} cgvp_stage_item;
int cgvp_stage_buffers(const cgvp_stage_item* items, int32_t num_items, void* stream);

/* ------------------------------------------------------------ DIAGNOSTICS
 * The ONE piece of process-global state in the library, off by default, not thread-safe; bench.py's roofline leg uses
 * it to time the dominant kernel in situ: while enabled, the whole-pass entry points bracket every conv-layer launch
 * (kind 0 = forward conv layer, 1 = conv backward) with a pair of HIP events on the pass's stream.
 * cgvp_debug_kernel_times synchronises the pairs recorded so far, writes their elapsed milliseconds and kinds (up to
 * `capacity`), forgets them and returns how many there were.  Not for use during stream capture. */
int cgvp_debug_kernel_timing(int32_t enable);
int cgvp_debug_kernel_times(float* ms, int32_t* kinds, int32_t capacity);

/* Library self-description (checked by the loader and the CPU test-suite). */
int cgvp_abi_version(void);
const char* cgvp_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* CASTER_GVP_H */
