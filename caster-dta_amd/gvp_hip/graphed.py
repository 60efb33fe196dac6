"""Shape-bucketed HIP-graph replay of an encoder training step for loops that feed a DIFFERENT batch every step.

The reference's loop (train_model.py:548-587) draws a new batch -- new N, E, edge_index -- every step.  Launched
eagerly, a step of both encoders costs ~0.6 ms of host time (PyTorch's autograd engine alone spends ~0.3 ms on the 74
parameter leaves) for ~0.25 ms of device work.  A captured step replays in one host call, but a HIP graph freezes its
shapes.  `GraphedEncoderStep` keeps one captured step per SHAPE BUCKET:

  * sizes (residues, protein edges, atoms, drug edges) are rounded up to the next multiple of 1/8 of their power of two
    (at most 12.5 % padding), a bucket = the four rounded sizes;
  * each bucket owns static input buffers; a step copies the batch into them with ONE launch (cgvp_stage_buffers) and
    replays the bucket's graph;
  * padding is inert by construction: padded residues / atoms are isolated nodes (their upstream gradient rows are zero,
    and every backward stage is linear in its upstream gradient), padded edges carry endpoint -1 and are dropped by the
    CSR build inside the captured step (tests/test_hip_random_graphs.py covers out-of-range endpoints); dropout is keyed
    by (seed, offset, layer, row, channel), so the real rows draw the same factors as in an unpadded eager step.

The outputs (embeddings of the real rows, every weight gradient) are the graph's static tensors: consume them (optimizer
step, all-reduce) before the next `run` of the same bucket.  Opt-in: the nn.Module API is untouched.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, ops


def bucket_size(n):
    """n rounded up to a multiple of 2^(floor(log2 n) - 3): at most 12.5 % larger, 8 buckets per octave."""
    n = int(n)
    if n <= 64:
        return 64
    q = 1 << max(n.bit_length() - 4, 0)
    return (n + q - 1) // q * q


class _Bucket:
    def __init__(self, owner, key, pdata, mdata, sdt):
        N, E, Na, Ea = key
        dev = owner.device
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        ni = pdata["x"][0].shape[1], pdata["x"][1].shape[1], pdata["eattr"][0].shape[1], pdata["eattr"][1].shape[1]
        self.p = dict(x=(z(N, ni[0], dt=sdt), z(N, ni[1], 3, dt=sdt)), edge_index=torch.full((2, E), -1, dtype=torch.int64, device=dev),
                      ntypes=z(N, dt=torch.int64), etypes=z(E, dt=torch.int64),
                      eattr=(z(E, ni[2], dt=sdt), z(E, ni[3], 3, dt=sdt)))
        self.m = dict(x=z(Na, mdata["x"].shape[1]), edge_index=torch.full((2, Ea), -1, dtype=torch.int64, device=dev),
                      ntypes=z(Na, dt=torch.int64), etypes=z(Ea, dt=torch.int64), eattr=z(Ea, mdata["eattr"].shape[1]))
        self.g_res, self.g_atm = z(N, owner.out_res, dt=sdt), z(Na, owner.out_atm)
        self.graph, self.out = None, None
        self.steps = 0


class GraphedEncoderStep:
    """step = forward of both encoders + backward with every weight gradient, protein chain on the launch stream, drug
    chain on a side stream (what bench.py times).  `run(pdata, mdata, g_res, g_atm)` -> (residues [N, 64],
    atoms [Na, 64], protein grads, drug grads) -- views of static tensors of the batch's bucket."""

    def __init__(self, protein_gnn, molecule_gnn, train=True, quantum=bucket_size):
        self.prot, self.mol = protein_gnn, molecule_gnn
        self.pp = [p for p in protein_gnn.parameters() if p.numel()]
        self.mp = [p for p in molecule_gnn.parameters() if p.numel()]
        self.device = self.pp[0].device
        self.out_res = protein_gnn.out_channels[0] if isinstance(protein_gnn.out_channels, (tuple, list)) else protein_gnn.out_channels
        self.out_atm = molecule_gnn.out_channels
        self.train, self.quantum = train, quantum
        self.buckets = {}
        self.stream = torch.cuda.Stream(device=self.device)          # capture + replay stream (its CSR counters persist)
        self.side = torch.cuda.Stream(device=self.device)
        self.pool = None

    def _step(self, b):
        main_s = torch.cuda.current_stream()
        self.side.wait_stream(main_s)
        res = self.prot(**b.p)
        with torch.cuda.stream(self.side):
            atm = self.mol(**b.m)
        if not self.train:
            main_s.wait_stream(self.side)
            return res, atm, (), ()
        gp = torch.autograd.grad([res], self.pp, [b.g_res])
        with torch.cuda.stream(self.side):
            gd = torch.autograd.grad([atm], self.mp, [b.g_atm])
        main_s.wait_stream(self.side)
        return res, atm, gp, gd

    def _stage(self, b, pdata, mdata, g_res, g_atm):
        pairs = [(b.p["x"][0], pdata["x"][0]), (b.p["x"][1], pdata["x"][1]), (b.p["ntypes"], pdata["ntypes"]),
                 (b.p["etypes"], pdata["etypes"]), (b.p["eattr"][0], pdata["eattr"][0]), (b.p["eattr"][1], pdata["eattr"][1]),
                 (b.m["x"], mdata["x"]), (b.m["ntypes"], mdata["ntypes"]), (b.m["etypes"], mdata["etypes"]),
                 (b.m["eattr"], mdata["eattr"])]
        if self.train:
            pairs += [(b.g_res, g_res), (b.g_atm, g_atm)]
        items = (_lib.StageItem * _lib.MAX_STAGE)()
        keep, k = [], 0
        for dst, src in pairs:
            if src.dtype != dst.dtype or not src.is_contiguous():
                src = src.to(dst.dtype).contiguous()
                keep.append(src)
            items[k] = _lib.StageItem(dst.data_ptr(), src.data_ptr() if src.numel() else 0, src.numel() * src.element_size(),
                                      dst.numel() * dst.element_size(), 0)
            k += 1
        for dst, src in ((b.p["edge_index"], pdata["edge_index"]), (b.m["edge_index"], mdata["edge_index"])):
            src = src.contiguous()
            keep.append(src)
            E, Ecap = src.shape[1], dst.shape[1]
            for row in range(2):                                      # each row of [2, E] lands in its row of [2, Ecap], tail = -1
                items[k] = _lib.StageItem(dst[row].data_ptr(), src[row].data_ptr() if E else 0, E * 8, Ecap * 8, 0xFFFFFFFF)
                k += 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cgvp_stage_buffers(items, k, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "cgvp_stage_buffers")
        return keep

    def run(self, pdata, mdata, g_res=None, g_atm=None):
        N, E = pdata["x"][0].shape[0], pdata["edge_index"].shape[1]
        Na, Ea = mdata["x"].shape[0], mdata["edge_index"].shape[1]
        key = (self.quantum(N), self.quantum(E), self.quantum(Na), self.quantum(Ea))
        sdt = torch.bfloat16 if pdata["x"][0].dtype == torch.bfloat16 else torch.float32
        b = self.buckets.get((key, sdt))
        if b is not None and b.graph is not None:
            # steady state: stage + replay on the caller's stream (a graph replays on whatever stream it is launched on;
            # no cross-stream dependency, which costs ~10 us of idle device per hop)
            keep = self._stage(b, pdata, mdata, g_res, g_atm)
            b.graph.replay()
            b.steps += 1
            del keep                                         # same stream: the allocator's stream ordering covers the reuse
            res, atm, gp, gd = b.out
            return res[:N], atm[:Na], gp, gd
        caller = torch.cuda.current_stream()
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream), torch.set_grad_enabled(self.train):
            if b is None:
                b = self.buckets[(key, sdt)] = _Bucket(self, key, pdata, mdata, sdt)
            keep = self._stage(b, pdata, mdata, g_res, g_atm)
            old = ops.CSR_CACHE_ENABLED
            ops.CSR_CACHE_ENABLED = False                    # the captured step rebuilds its CSR tables from the staged edge_index
            try:
                for _ in range(2):                           # warm-up on the capture stream (generator state, counters exist)
                    self._step(b)
                torch.cuda.current_stream().synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream, pool=self.pool):
                    b.out = self._step(b)
                if self.pool is None:
                    self.pool = g.pool()                     # every bucket's graph shares one memory pool
                b.graph = g
            finally:
                ops.CSR_CACHE_ENABLED = old
            b.graph.replay()
            b.steps += 1
        caller.wait_stream(self.stream)
        for t in keep:
            t.record_stream(self.stream)
        res, atm, gp, gd = b.out
        return res[:N], atm[:Na], gp, gd


def _check_no_foreign_graph(params):
    """Observed on this ROCm / PyTorch build (tools/debug_graphed.py, 15 runs): capturing the WHOLE-MODEL backward while an
    autograd graph of an earlier eager step on another stream is still alive (its `pred` / `loss` held somewhere) ends in
    a segmentation fault inside hipStreamEndCapture, every time; with that graph dropped, or no eager step before, never.
    What such a graph keeps alive are the leaves' AccumulateGrad nodes, which remember the stream they were created under
    (csrc/torch_bridge.cpp, stale_grad_accumulators); the engine synchronises gradient hand-offs with that stream.  The
    encoder-only captures of GraphedEncoderStep never showed it.  Refuse with a message instead of crashing."""
    br = _lib.bridge()                     # (GraphedTrainStep refuses to exist without it)
    n = int(br.stale_grad_accumulators(list(params)))
    if n:
        raise RuntimeError(
            f"{n} parameters are still referenced by an autograd graph that was built on another stream (an output or loss "
            "of an earlier eager step is alive): capturing a backward pass now would pull that stream into the capture. "
            "Drop those tensors first (`del loss, pred`), or run the eager step after the captured one.")


class _TrainBucket:
    def __init__(self, owner, key, pdata, mdata, target, sdt):
        N, E, Na, Ea, B = key
        dev = owner.device
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        ni = pdata["x"][0].shape[1], pdata["x"][1].shape[1], pdata["eattr"][0].shape[1], pdata["eattr"][1].shape[1]
        # B real pairs + ONE padding pair that owns every padded residue / atom (never empty: the sizes are rounded up from
        # n + 1), so that the head's per-pair pooling and attention see B + 1 well-formed pairs and the loss reads the first B
        pptr = torch.full((B + 2,), N, dtype=torch.int64, device=dev)
        mptr = torch.full((B + 2,), Na, dtype=torch.int64, device=dev)
        self.p = dict(x=(z(N, ni[0], dt=sdt), z(N, ni[1], 3, dt=sdt)), edge_index=torch.full((2, E), -1, dtype=torch.int64, device=dev),
                      ntypes=z(N, dt=torch.int64), etypes=z(E, dt=torch.int64),
                      eattr=(z(E, ni[2], dt=sdt), z(E, ni[3], 3, dt=sdt)), ptr=pptr)
        self.m = dict(x=z(Na, mdata["x"].shape[1]), edge_index=torch.full((2, Ea), -1, dtype=torch.int64, device=dev),
                      ntypes=z(Na, dt=torch.int64), etypes=z(Ea, dt=torch.int64), eattr=z(Ea, mdata["eattr"].shape[1]), ptr=mptr)
        self.target = torch.zeros((B,) + tuple(target.shape[1:]), dtype=target.dtype, device=dev)
        self.rows_p = torch.arange(N, device=dev)
        self.rows_m = torch.arange(Na, device=dev)
        self.graph, self.out, self.steps = None, None, 0


class GraphedTrainStep:
    """The WHOLE JointGNN training step -- forward of both encoders and the head, loss, backward with all 764k gradients --
    replayed from shape-bucketed HIP graphs, for the loop of train_model.py:548-587 that feeds a different batch every step:

        step = GraphedTrainStep(model, torch.nn.functional.mse_loss)        # optional: scaler=GradScaler, autocast=dtype
        for pdata, mdata, y in loader:            # the dicts carry 'ptr' (PyG Batch objects do)
            loss, pred = step.run(pdata, mdata, y)                           # p.grad of every parameter is set
            optimizer.step()

    One launch stages the batch into the bucket's padded static buffers, one replay runs the step.  Padding: padded
    residues / atoms are isolated nodes that belong to an extra (B+1)-th pair whose prediction the loss never reads, so
    every real pair's numbers and every gradient equal those of the eager step on the unpadded batch (padded rows have
    zero upstream gradient; tests/test_hip_models.py::test_graphed_train_step_equals_eager).  The gradients are the graph's
    static tensors: consume them (optimizer step) before the next `run` of the same bucket."""

    def __init__(self, model, loss_fn, scaler=None, autocast=None, quantum=bucket_size):
        if _lib.bridge() is None:
            # the capture-safety check below (_check_no_foreign_graph) lives in the C++ bridge: without it a forgotten eager
            # `loss` / `pred` turns the next capture into a segmentation fault inside hipStreamEndCapture
            raise RuntimeError("GraphedTrainStep needs the C++ eager bridge (lib/caster_gvp_torch.so, built by "
                               "__graft_entry__.build(); CGVP_BRIDGE=0 / CGVP_LIB_PATH turn it off)")
        self.model, self.loss_fn, self.scaler, self.autocast, self.quantum = model, loss_fn, scaler, autocast, quantum
        self.params = [p for p in model.parameters() if p.numel() and p.requires_grad]
        self.device = self.params[0].device
        self.buckets = {}
        self.stream = None                      # capture + replay stream, created at the first capture
        self.pool = None

    def _step(self, b, B):
        import contextlib
        ctx = torch.autocast("cuda", dtype=self.autocast) if self.autocast is not None else contextlib.nullcontext()
        # pair index of every row from the staged offsets (no host sync): rows behind the last real one -> the padding pair B
        pd = dict(b.p, batch=self._pair_of_rows(b.p["ptr"], b.rows_p))
        md = dict(b.m, batch=self._pair_of_rows(b.m["ptr"], b.rows_m))
        aw = getattr(self.model, "attention_weights", None)
        if aw is not None:
            self.model.attention_weights = "never"      # (the padded weight tensors need a host-side maximum: not capturable)
        try:
            with ctx:
                pred, _ = self.model(pd, md)
                loss = self.loss_fn(pred[:B], b.target)
        finally:
            if aw is not None:
                self.model.attention_weights = aw
        out = self.scaler.scale(loss) if self.scaler is not None else loss
        grads = torch.autograd.grad(out, self.params, allow_unused=True)
        return loss.detach(), pred[:B].detach(), grads

    @staticmethod
    def _pair_of_rows(ptr, rows):
        """pair index of every row = number of pair END offsets <= row (plain compare + sum: capturable, no host sync)"""
        if os.environ.get("CGVP_GRAPHED_BATCH") == "searchsorted":
            return torch.searchsorted(ptr[1:].contiguous(), rows, right=True)
        return (rows.unsqueeze(1) >= ptr[1:].unsqueeze(0)).sum(dim=1)

    def _stage(self, b, pdata, mdata, target):
        pairs = [(b.p["x"][0], pdata["x"][0]), (b.p["x"][1], pdata["x"][1]), (b.p["ntypes"], pdata["ntypes"]),
                 (b.p["etypes"], pdata["etypes"]), (b.p["eattr"][0], pdata["eattr"][0]), (b.p["eattr"][1], pdata["eattr"][1]),
                 (b.m["x"], mdata["x"]), (b.m["ntypes"], mdata["ntypes"]), (b.m["etypes"], mdata["etypes"]),
                 (b.m["eattr"], mdata["eattr"]), (b.target, target),
                 (b.p["ptr"][:-1], pdata["ptr"]), (b.m["ptr"][:-1], mdata["ptr"])]      # (the last offset stays the padded size)
        items = (_lib.StageItem * _lib.MAX_STAGE)()
        keep, k = [], 0
        for dst, src in pairs:
            if src.dtype != dst.dtype or not src.is_contiguous():
                src = src.to(dst.dtype).contiguous()
                keep.append(src)
            items[k] = _lib.StageItem(dst.data_ptr(), src.data_ptr() if src.numel() else 0, src.numel() * src.element_size(),
                                      dst.numel() * dst.element_size(), 0)
            k += 1
        for dst, src in ((b.p["edge_index"], pdata["edge_index"]), (b.m["edge_index"], mdata["edge_index"])):
            src = src.contiguous()
            keep.append(src)
            E, Ecap = src.shape[1], dst.shape[1]
            for row in range(2):
                items[k] = _lib.StageItem(dst[row].data_ptr(), src[row].data_ptr() if E else 0, E * 8, Ecap * 8, 0xFFFFFFFF)
                k += 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cgvp_stage_buffers(items, k, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "cgvp_stage_buffers")
        return keep

    def _publish(self, b):
        loss, pred, grads = b.out
        for p, g in zip(self.params, grads):
            if g is not None and p.grad is not g:
                p.grad = g
        return loss, pred

    def run(self, pdata, mdata, target):
        if "ptr" not in pdata or "ptr" not in mdata:
            raise ValueError("GraphedTrainStep needs the graph offsets 'ptr' in both dicts (PyG Batch objects carry them)")
        N, E = pdata["x"][0].shape[0], pdata["edge_index"].shape[1]
        Na, Ea = mdata["x"].shape[0], mdata["edge_index"].shape[1]
        B = int(pdata["ptr"].numel()) - 1
        key = (self.quantum(N + 1), self.quantum(E), self.quantum(Na + 1), self.quantum(Ea), B)
        sdt = torch.bfloat16 if pdata["x"][0].dtype == torch.bfloat16 else torch.float32
        b = self.buckets.get((key, sdt))
        if b is not None and b.graph is not None:
            keep = self._stage(b, pdata, mdata, target)
            b.graph.replay()
            b.steps += 1
            del keep
            return self._publish(b)
        caller = torch.cuda.current_stream()
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        self.stream.wait_stream(caller)
        was_training = self.model.training
        with torch.cuda.stream(self.stream), torch.enable_grad():
            _check_no_foreign_graph(self.params)
            if b is None:
                b = self.buckets[(key, sdt)] = _TrainBucket(self, key, pdata, mdata, target, sdt)
            keep = self._stage(b, pdata, mdata, target)
            old = ops.CSR_CACHE_ENABLED
            ops.CSR_CACHE_ENABLED = False
            try:
                for _ in range(2):
                    self._step(b, B)
                torch.cuda.current_stream().synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream, pool=self.pool):
                    b.out = self._step(b, B)
                if self.pool is None:
                    self.pool = g.pool()
                b.graph = g
            finally:
                ops.CSR_CACHE_ENABLED = old
            b.graph.replay()
            b.steps += 1
        caller.wait_stream(self.stream)
        for t in keep:
            t.record_stream(self.stream)
        self.model.train(was_training)
        return self._publish(b)
