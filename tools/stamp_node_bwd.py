"""Diagnostic: cycles per segment of a node_bwd wave (-DCGVP_STAMPS build).  The conv_bwd stamps share the
buffer, so only the LAST backward kernel that ran (node_bwd of layer 0, no head) is read unless HEAD=1."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "caster-dta_amd"), REPO]
from gvp_hip import _lib
_lib.LIB_PATH = os.path.join(REPO, "caster-dta_amd", "lib", "_stamps", "libcaster_gvp_stamps.so")
import davis_synth as ds
from gvp_hip import ops
import ctypes as C
import __graft_entry__ as entry
dev = torch.device("cuda:0")
model, state = entry._load_model(dev)
pb = ds.protein_batch(64, 0)
d = {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in ds.to_torch(pb).items()}
h = ctypes.CDLL(_lib.LIB_PATH)
enc = model.protein_gnn.gnn_model
from gvp_hip import autograd_ops
params = enc._arena_buffer(); dims, layout = enc._hip_config(); image = ops.prepare_image(params, layout, dims)
N = pb.num_nodes
L = _lib.lib()
f32 = dict(dtype=torch.float32, device=dev)
hh, dh, g = torch.randn(N, 28, **f32), torch.randn(N, 28, **f32) * 0.1, torch.randn(N, 28, **f32)
gout = torch.randn(N, 64, **f32)
gdh = torch.empty(N, 28, **f32); gp = torch.zeros(layout.total, **f32)
ws = torch.empty(int(L.cgvp_bwd_workspace_floats(C.byref(dims), C.byref(layout))), **f32)
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
for head in (0, 1):
    buf = torch.zeros(4096 * 16 + 8 * 4096 * 2, dtype=torch.int64, device=dev)
    h.cgvp_debug_set_stamp_buffer_bwd(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        rc = L.cgvp_node_update_bwd(C.byref(dims), C.byref(layout), P(image), 1 if head else 0, P(hh), P(dh), P(None), P(None), None,
                                    P(hh if head else None), P(gout if head else None), P(None if head else g), P(None), P(None), N, head, P(gdh), P(None), P(None),
                                    P(gp), P(ws), P(None), P(None), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    torch.cuda.synchronize()
    b = buf.cpu().numpy()[:4096 * 16].reshape(4096, 16); b = b[b[:, 10] > 0]
    names = ["stage images", "barrier", "fwd recompute", "head fwd+bwd+wgrads+LN" if head else "load upstream", "LN1 bwd",
             "ff1 backward", "ff1 weight_grads", "ff0 backward", "ff0 weight_grads", "LN0 bwd + store (to loop end)", "barrier + slab write"]
    print(f"--- node_bwd head={head}: waves {len(b)}")
    seq = [(0,1),(1,2),(2,3),(3,4),(4,5),(5,6),(6,7),(7,8),(8,9),(9,10)]
    for k,(a_,b_) in enumerate(seq):
        print(f"{names[k+1]:32s} median {np.median(b[:,b_]-b[:,a_]):8.0f} cyc")
    print(f"{'wave total':32s} median {np.median(b[:,10]-b[:,0]):8.0f} cyc")
    last = b[:, 1:11].max(axis=1)
    t0 = b[:, 0].min()
    print(f"{'first-stamp skew':32s} median {np.median(b[:,0]-t0):8.0f}  max {np.max(b[:,0]-t0):8.0f}")
    print(f"{'span first stamp -> last stamp':32s} {last.max()-t0:8.0f} cyc  (median wave end {np.median(last-t0):8.0f})")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    segs = (_lib.Segment * 8)(); nsegs = C.c_int32(0)
    ev0.record()
    for _ in range(20):
        nsegs.value = 0
        L.cgvp_node_update_bwd(C.byref(dims), C.byref(layout), P(image), 1 if head else 0, P(hh), P(dh), P(None), P(None), None,
                               P(hh if head else None), P(gout if head else None), P(None if head else g), P(None), P(None), N, head, P(gdh), P(None), P(None),
                               P(gp), P(ws), segs, C.byref(nsegs), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    ev1.record(); torch.cuda.synchronize()
    print(f"{'kernel alone (deferred reduce), back to back':32s} {ev0.elapsed_time(ev1)/20*1000:8.1f} us")
