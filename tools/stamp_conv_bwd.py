"""Diagnostic: where does a conv_bwd wave spend its cycles?  Uses the -DCGVP_STAMPS build."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "caster-dta_amd"), REPO]
from gvp_hip import _lib
_lib.LIB_PATH = os.path.join(REPO, "caster-dta_amd", "lib", "_stamps", "libcaster_gvp_stamps.so")
import davis_synth as ds
import __graft_entry__ as entry
dev = torch.device("cuda:0")
model, state = entry._load_model(dev)
wl = sys.argv[1] if len(sys.argv) > 1 else "davis"
pb = ds.protein_batch(64, 0) if wl == "davis" else ds.protein_batch(64, 0, length=1000, thresh=20, thresh_type="num")
d = {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in ds.to_torch(pb).items()}
h = ctypes.CDLL(_lib.LIB_PATH)
nwaves = 4096
buf = torch.zeros(nwaves * 16 + 8 * nwaves * 2, dtype=torch.int64, device=dev)   # cycle stamps | wall stamps of up to 8 kernel kinds (WallStamp)
h.cgvp_debug_set_stamp_buffer_bwd(ctypes.c_void_p(buf.data_ptr()))
params = [p for p in model.protein_gnn.parameters() if p.numel()]
for _ in range(3):
    out = model.protein_gnn(**d)
    torch.autograd.grad(out, params, torch.ones_like(out))
torch.cuda.synchronize()
b = buf.cpu().numpy()[:nwaves * 16].reshape(nwaves, 16)
v2 = os.environ.get("CGVP_CONV_BWD") != "1"
# every backward kernel stamps into the same buffer (node_bwd runs 1,920 waves, conv_bwd2 960): keep the conv waves only
b = b[b[:, 15] > 0] if v2 else b[b[:, 9] > 0]
names = {1: "stage + barrier", 2: "gather (last tile)", 3: "fwd recompute", 4: "3 msg GVP bwd + wgrads", 5: "edge LN/GVP bwd + wgrads",
         6: "g_src atomics", 7: "scan + LDS adds"}
print("waves stamped:", len(b), "(per-tile segments are those of the wave's LAST tile)")
prev = 1
b[:, 1] = b[:, 1]
print(f"{'stage + barrier':28s} median {np.median(b[:,1]-b[:,0]):8.0f}")
for s in range(3, 8):
    dt = b[:, s] - b[:, s - 1]
    print(f"{names[s]:28s} median {np.median(dt):8.0f} cyc   p90 {np.percentile(dt, 90):8.0f}")
seq = [(3, 10, "msg2 backward"), (10, 11, "msg2 weight_grads"), (11, 12, "msg1 backward"), (12, 13, "msg1 weight_grads"),
       (13, 14, "msg0 backward"), (14, 4, "msg0 weight_grads")]
for a_, b_, nm in seq:
    dt = b[:, b_] - b[:, a_]
    print(f"   {nm:25s} median {np.median(dt):8.0f} cyc")
if os.environ.get("CGVP_CONV_BWD") != "1":     # conv_bwd2_kernel: slot 15 = top of the wave's LAST iteration (TN tiles in lockstep)
    print(f"{'last iteration (TN tiles)':28s} median {np.median(b[:,7]-b[:,15]):8.0f} cyc   p90 {np.percentile(b[:,7]-b[:,15], 90):8.0f}")
    print(f"{'   gather':28s} median {np.median(b[:,2]-b[:,15]):8.0f}")
    print(f"{'   g_e stores':28s} median {np.median(b[:,5]-b[:,4]):8.0f}")
print(f"{'wave total (all tiles)':28s} median {np.median(b[:,8]-b[:,0]):8.0f} cyc   p90 {np.percentile(b[:,8]-b[:,0], 90):8.0f}")
print(f"{'final barrier + slab write':28s} median {np.median(b[:,9]-b[:,8]):8.0f}")
