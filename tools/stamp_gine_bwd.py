"""Diagnostic: where does a gine_quad_bwd wave spend its cycles?  Needs a -DCGVP_STAMPS build of gine_quad_kernels.hip:
    bash tools/build_variant.sh stamps gine_quad_kernels "-DCGVP_STAMPS"        (then run this on the GPU box)
Per-tile segments are those of the wave's LAST tile; one row per wave, layer 1 (16,64,64) first, then layer 0."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "caster-dta_amd"), REPO]
os.environ["CGVP_LIB_PATH"] = os.path.join(REPO, "caster-dta_amd", "lib", "ab", "libcaster_gvp_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamps"))
from gvp_hip import _lib
import davis_synth as ds
import __graft_entry__ as entry
dev = torch.device("cuda:0")
model, state = entry._load_model(dev)
mb = ds.drug_batch(64, 0) if hasattr(ds, "drug_batch") else ds.pair_batch(64, 0)[1]
d = {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in ds.to_torch(mb).items()}
h = ctypes.CDLL(os.environ["CGVP_LIB_PATH"])
nwaves = 512
names = ["stage", "barrier", "A aggregate (all chunks)", "B MLP fwd + data bwd", "B weight grads dW1 dW0", "B d eps / d x / rows",
         "C edges: dmsg, atomics, dWe", "slab write"]
params = [p for p in model.molecule_gnn.parameters() if p.numel()]
for which in ("both layers (last launch wins = layer 0)",):
    buf = torch.zeros(nwaves * 16, dtype=torch.int64, device=dev)
    h.cgvp_debug_set_stamp_buffer_gine(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        out = model.molecule_gnn(**d)
        torch.autograd.grad(out, params, torch.ones_like(out))
    torch.cuda.synchronize()
    b = buf.cpu().numpy().reshape(nwaves, 16)
    b = b[b[:, 7] > 0]
    print(which, "waves stamped:", len(b))
    for s in range(1, 8):
        dt = b[:, s] - b[:, s - 1]
        print(f"  {names[s]:32s} median {np.median(dt):9.0f} cyc   p90 {np.percentile(dt, 90):9.0f}")
    one = b[(b[:, 8] - b[:, 1]) < 3000]              # waves whose LAST tile is their first (stamp 8 right after the barrier)
    print(f"  one-tile waves: {len(one)};  rowptr + loop entry {np.median(one[:, 8] - one[:, 1]):7.0f}   gather (load_chunk, last chunk) "
          f"{np.median(b[:, 9] - b[:, 8]):7.0f} p90 {np.percentile(b[:, 9] - b[:, 8], 90):7.0f}   messages + scan + row adds {np.median(b[:, 2] - b[:, 9]):7.0f} "
          f"p90 {np.percentile(b[:, 2] - b[:, 9], 90):7.0f}")
    print(f"  {'wave total':32s} median {np.median(b[:, 7] - b[:, 0]):9.0f} cyc (100 MHz s_memtime? see note)  max {(b[:, 7] - b[:, 0]).max():9.0f}")
