"""Diagnostic (-DCGVP_STAMPS build): wall-clock start / end of every wave of each protein backward kernel in one training
pass -- start skew (a second round of workgroups shows up as a late group of waves), wave durations, first-start to
last-end span.  The kernel kinds that run twice per pass (node_bwd, conv_bwd) show their LAST launch (layer 0)."""
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
model.train()
pb = ds.protein_batch(64, 0)
d = {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in ds.to_torch(pb).items()}
h = ctypes.CDLL(_lib.LIB_PATH)
NW, KINDS = 4096, ["node_bwd", "head_bwd", "conv_bwd", "edge_bwd", "embed_bwd"]
buf = torch.zeros(NW * 16 + len(KINDS) * NW * 2, dtype=torch.int64, device=dev)
h.cgvp_debug_set_stamp_buffer_bwd(ctypes.c_void_p(buf.data_ptr()))
params = [p for p in model.protein_gnn.parameters() if p.numel()]
for _ in range(3):
    out = model.protein_gnn(**d)
    torch.autograd.grad(out, params, torch.ones_like(out))
torch.cuda.synchronize()
w = buf.cpu().numpy()[NW * 16:].reshape(len(KINDS), NW, 2).astype(np.float64)
for k, name in enumerate(KINDS):
    b = w[k][w[k][:, 1] > 0]
    t0 = b[:, 0].min()
    skew, dur = (b[:, 0] - t0) / 100, (b[:, 1] - b[:, 0]) / 100
    print(f"{name:10s} waves {len(b):5d}  span {(b[:, 1].max() - t0) / 100:6.2f} us | start skew median {np.median(skew):5.2f} "
          f"p90 {np.percentile(skew, 90):5.2f} max {skew.max():5.2f} | wave duration median {np.median(dur):6.2f} p10 "
          f"{np.percentile(dur, 10):6.2f} p90 {np.percentile(dur, 90):6.2f} max {dur.max():6.2f}")
