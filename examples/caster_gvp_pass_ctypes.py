# models/_caster_gvp_pass.py -- ctypes binding of the WHOLE-PASS entry points of libcaster_gvp.so (C ABI v29): one call
# runs VectorProteinGNN_LBAModel.forward (protein_gnn.py:361-388), one call its backward -- the production path.
# Nothing but torch + ctypes; every buffer is the caller's.
import ctypes as C
import torch                                           # import torch first: it provides the HIP runtime


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("node_in_s", "node_in_v", "edge_in_s", "edge_in_v", "hidden_s", "hidden_v",
                                         "edge_hidden_s", "edge_hidden_v", "out_s", "storage", "layer_kind")]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nt_node", "nt_edge", "node_gvp", "node_ln", "edge_gvp", "edge_ln", "conv0",
                                         "conv_stride", "ln_out", "head", "total")]


class Batch(C.Structure):                              # cgvp_lba_batch
    _fields_ = [("num_nodes", C.c_int64), ("num_edges", C.c_int64)] + [(n, C.c_void_p) for n in (
        "x_s", "x_v", "ntypes", "e_s", "e_v", "etypes", "edge_index", "rowptr", "eperm", "esrc", "edst")]


class FwdWs(C.Structure):                              # cgvp_lba_fwd_ws: byte offsets inside the forward workspace
    _fields_ = [(n, C.c_int64) for n in ("seed", "image", "state", "e_emb", "rowptr", "eperm", "esrc", "edst",
                                         "ids_scratch", "total", "state_rows", "node_stride")]


P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


class ProteinEncoder:
    def __init__(self, lib_path, state_dict, num_convs=2, num_ntypes=20, num_etypes=1):
        """state_dict: the `protein_gnn.gnn_model.*` slice of a CASTER-DTA checkpoint (keys without that prefix)."""
        self.lib = lib = C.CDLL(lib_path)
        lib.cgvp_lba_bwd_workspace_bytes.restype = C.c_int64
        lib.cgvp_lba_bwd_workspace_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        assert lib.cgvp_abi_version() == 29
        self.dims, self.lay = Dims(17, 3, 32, 1, 16, 4, 32, 1, 64, 0, 0), Layout()
        check(lib.cgvp_lba_layout(C.byref(self.dims), num_ntypes, num_etypes, num_convs, C.byref(self.lay)), "cgvp_lba_layout")
        # ONE fp32 arena with the weights in state_dict order (zero-size dummy_params skipped); gradients come back the same way
        self.params = torch.cat([v.reshape(-1).float() for v in state_dict.values() if v.numel()]).cuda()
        assert self.params.numel() == self.lay.total
        self.counters = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")      # persistent, zero between calls
        self.rng = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")        # persistent dropout {seed, offset}

    def _batch(self, x_s, x_v, ntypes, edge_index, e_s, e_v, etypes):
        return Batch(x_s.shape[0], edge_index.shape[1], x_s.data_ptr(), x_v.data_ptr(), ntypes.data_ptr(), e_s.data_ptr(),
                     e_v.data_ptr(), etypes.data_ptr(), edge_index.data_ptr(), 0, 0, 0, 0)

    def forward(self, x_s, x_v, ntypes, edge_index, e_s, e_v, etypes, dropout_p=0.0):
        """-> (out [N, 64], workspace): keep `workspace` for backward()."""
        N, E = x_s.shape[0], edge_index.shape[1]
        assert N + 1 <= self.counters.numel()
        w = FwdWs()
        check(self.lib.cgvp_lba_fwd_workspace(C.byref(self.dims), C.byref(self.lay), C.c_int64(N), C.c_int64(E), 1, C.byref(w)),
              "cgvp_lba_fwd_workspace")
        ws = torch.empty(w.total, dtype=torch.uint8, device="cuda")
        out = torch.empty(N, 64, device="cuda")
        b = self._batch(x_s, x_v, ntypes, edge_index, e_s, e_v, etypes)
        check(self.lib.cgvp_lba_forward_pass(C.byref(self.dims), C.byref(self.lay), P(self.params), C.byref(b), 0,
                                             C.c_float(dropout_p), P(self.rng), None, P(self.counters), P(ws), 1, 0, P(out), S()),
              "cgvp_lba_forward_pass")
        return out, ws

    def backward(self, g_out, ws, x_s, x_v, ntypes, edge_index, e_s, e_v, etypes, dropout_p=0.0):
        """-> (gradient arena [15117 for CASTER-DTA(2,2)] in the layout of `params`, d x_s, d x_v)."""
        N, E = x_s.shape[0], edge_index.shape[1]
        nbytes = self.lib.cgvp_lba_bwd_workspace_bytes(C.byref(self.dims), C.byref(self.lay), N, E)
        bws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        gparams = torch.empty(self.lay.total, device="cuda")
        g_xs, g_xv = torch.empty(N, 17, device="cuda"), torch.empty(N, 3, 3, device="cuda")
        b = self._batch(x_s, x_v, ntypes, edge_index, e_s, e_v, etypes)
        check(self.lib.cgvp_lba_backward_pass(C.byref(self.dims), C.byref(self.lay), C.byref(b), 0, C.c_float(dropout_p), None,
                                              P(ws), P(g_out.contiguous()), P(bws), P(gparams), P(g_xs), P(g_xv), None, None, S()),
              "cgvp_lba_backward_pass")
        return gparams, g_xs, g_xv
