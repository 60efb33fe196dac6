# models/_caster_gvp.py -- ctypes binding of libcaster_gvp.so (C ABI v29): the protein encoder forward
# (VectorProteinGNN_LBAModel.forward, protein_gnn.py:361-388) on the MFMA kernels, nothing but torch + ctypes.
import ctypes as C
import torch                                           # import torch first: it provides the HIP runtime


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("node_in_s", "node_in_v", "edge_in_s", "edge_in_v", "hidden_s", "hidden_v",
                                         "edge_hidden_s", "edge_hidden_v", "out_s", "storage", "layer_kind")]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nt_node", "nt_edge", "node_gvp", "node_ln", "edge_gvp", "edge_ln", "conv0",
                                         "conv_stride", "ln_out", "head", "total")]


P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
I64 = C.c_int64


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


class ProteinEncoder:
    def __init__(self, lib_path, state_dict, num_convs=2, num_ntypes=20, num_etypes=1):
        """state_dict: the `protein_gnn.gnn_model.*` slice of a CASTER-DTA checkpoint (keys without that prefix)."""
        self.lib = lib = C.CDLL(lib_path)
        lib.cgvp_lba_image_floats.restype = C.c_int64
        assert lib.cgvp_abi_version() == 29
        self.num_convs = num_convs
        self.dims, self.lay = Dims(17, 3, 32, 1, 16, 4, 32, 1, 64, 0, 0), Layout()     # storage 0 = fp32 activations, 1 = bf16
        check(lib.cgvp_lba_layout(C.byref(self.dims), num_ntypes, num_etypes, num_convs, C.byref(self.lay)), "cgvp_lba_layout")
        # params: ONE fp32 tensor with the weights in state_dict order (the zero-size dummy_params are skipped)
        self.params = torch.cat([v.reshape(-1).float() for v in state_dict.values() if v.numel()]).cuda()
        assert self.params.numel() == self.lay.total
        self.image = torch.empty(lib.cgvp_lba_image_floats(C.byref(self.dims), C.byref(self.lay)), device="cuda")

    def forward(self, x_s, x_v, ntypes, edge_index, e_s, e_v, etypes):
        lib, d, lay = self.lib, C.byref(self.dims), C.byref(self.lay)
        N, E = x_s.shape[0], e_s.shape[0]
        i32 = dict(dtype=torch.int32, device="cuda")
        rowptr, work = torch.empty(N + 1, **i32), torch.zeros((N + 64) // 64 * 64, **i32)
        eperm, esrc, edst, ids = (torch.empty(max(E, 1), **i32) for _ in range(4))
        h, h2, dh = (torch.empty(N, 28, device="cuda") for _ in range(3))
        out = torch.empty(N, 64, device="cuda")
        # one launch: node embedding + fragment image of the current weights + per-target edge counts of the CSR build
        check(lib.cgvp_lba_pass_begin(d, lay, P(self.params), P(self.image), P(x_s), P(x_v), P(ntypes), I64(N), P(h), None,
                                      None, P(edge_index), I64(E), P(work), S()), "cgvp_lba_pass_begin")
        check(lib.cgvp_csr_from_coo(P(edge_index), I64(N), I64(E), P(rowptr), P(eperm), P(esrc), P(edst), P(work), 2,
                                    P(ids), S()), "cgvp_csr_from_coo")       # work_is_zero = 2: already counted
        for layer in range(self.num_convs):
            last = layer == self.num_convs - 1
            check(lib.cgvp_conv_fwd(d, lay, P(self.params), P(self.image), layer, P(h), P(e_s), P(e_v), P(etypes),
                                    P(rowptr), P(eperm), P(esrc), P(edst), I64(N), I64(E), 0, None, None, P(dh), S()),
                  "cgvp_conv_fwd")
            check(lib.cgvp_node_update_fwd(d, lay, P(self.params), P(self.image), layer, P(h), P(dh), I64(N), int(last),
                                           P(h2), P(out), S()), "cgvp_node_update_fwd")
            h, h2 = h2, h
        return out
