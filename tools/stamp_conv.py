"""Diagnostic: where does a conv_quad wave spend its cycles?  Uses the -DCGVP_STAMPS build."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "caster-dta_amd"), REPO]
from gvp_hip import _lib
_lib.LIB_PATH = os.path.join(REPO, "caster-dta_amd", "lib", "_stamps", "libcaster_gvp_stamps.so")
import davis_synth as ds
from gvp_hip import ops, arena
import __graft_entry__ as entry
dev = torch.device("cuda:0")
model, state = entry._load_model(dev)
pb = ds.protein_batch(64, 0)
d = {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in ds.to_torch(pb).items()}
h = ctypes.CDLL(_lib.LIB_PATH)
nwaves = 4096
buf = torch.zeros(nwaves * 16 + 8 * nwaves * 2, dtype=torch.int64, device=dev)   # cycle stamps | wall stamps of up to 8 kernel kinds (WallStamp)
h.cgvp_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
with torch.no_grad():
    for _ in range(5):
        model.protein_gnn(**d)
    torch.cuda.synchronize()
b = buf.cpu().numpy()[:nwaves * 16].reshape(nwaves, 16)
b = b[b[:, 0] > 0]
names = {1: "gather issue+stage+clear", 2: "barrier", 3: "loop head", 5: "edge GVP+LN (incl. gather wait)", 6: "3 message GVPs",
         7: "scan + LDS adds", 8: "barrier", 9: "write dh"}
print("waves stamped:", len(b))
prev = 0
for s in sorted(names):
    dt = b[:, s] - b[:, prev]
    print(f"{names[s]:22s} median {np.median(dt):8.0f} cyc   p90 {np.percentile(dt, 90):8.0f}")
    prev = s
print(f"{'fused node update (+head)':22s} median {np.median(b[:, 10] - b[:, 9]):8.0f} cyc   p90 {np.percentile(b[:, 10] - b[:, 9], 90):8.0f}   (last conv launch of the pass)")
tot = b[:, 10] - b[:, 0]
print(f"{'wave total':22s} median {np.median(tot):8.0f} cyc   p90 {np.percentile(tot, 90):8.0f}")
r0, r1 = b[:, 14].astype(np.float64), b[:, 15].astype(np.float64)
print(f"wall clock (s_memrealtime, 100 MHz): first wave start -> last wave end {(r1.max() - r0.min()) / 100:.2f} us; "
      f"wave start skew median {np.median(r0 - r0.min()) / 100:.2f} us, p90 {np.percentile(r0 - r0.min(), 90) / 100:.2f}, max {(r0 - r0.min()).max() / 100:.2f}; "
      f"wave duration median {np.median(r1 - r0) / 100:.2f} us")
print("kernel span (first stamp -> last stamp):", (b[:, 10].max() - b[:, 0].min()), "cyc;  first-stamp skew median", np.median(b[:, 0] - b[:, 0].min()), "max", (b[:, 0] - b[:, 0].min()).max())
