#!/usr/bin/env python3
"""CPU time of the oracle (oracle/gvp_oracle.py, the `cpu_baseline` of bench.py) against the UNMODIFIED reference protein
encoder on the same batch ("the restatement/reference time ratio is known").  Runs only in the build container
(needs /root/reference; same stand-ins as make_golden.py); prints one line for BASELINE.md."""
import importlib.util, json, os, statistics, sys, time
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
mg.install_standins()
sys.path.insert(0, mg.REF)
from models.protein_gnn import SelectableProteinModelWrapper          # the reference
sys.path.insert(0, REPO)
from oracle import gvp_oracle as O
ds = mg.ds
ckpt = torch.load(os.path.join(mg.REF, "pretrained_model_downstream", "bestvalmodel_bindingdb_val0.6889_epoch01011.pt"),
                  map_location="cpu", weights_only=True)
ckpt = {k.replace("_orig_mod.", ""): v for k, v in ckpt.items()}
kw = json.load(open(os.path.join(mg.REF, "pretrained_model_downstream", "model_kwargs.json")))["protein_gnn_kwargs"]
for k in ("in_channels", "edge_dim", "hidden_channels", "edge_hidden_channels"):
    kw[k] = tuple(kw[k])
ref = SelectableProteinModelWrapper(**kw).eval()
ref.load_state_dict({k[len("protein_gnn."):]: v for k, v in ckpt.items() if k.startswith("protein_gnn.")})
P = {k[len("protein_gnn.gnn_model."):]: v for k, v in ckpt.items() if k.startswith("protein_gnn.gnn_model.")}
d = ds.to_torch(ds.protein_batch(16, 0))                               # BASELINE config 1: 16 x 300 residues


def bench(fn, grad):
    ts = []
    with torch.set_grad_enabled(grad):
        for i in range(13):
            t0 = time.perf_counter()
            out = fn()
            if grad:
                out.square().mean().backward()
            ts.append(time.perf_counter() - t0)
    return statistics.median(ts[3:])


for threads in (8, 1):
    torch.set_num_threads(threads)
    for grad in (False, True):
        Pg = {k: v.clone().requires_grad_(grad and v.numel() > 0) for k, v in P.items()}
        for p in ref.parameters():
            p.requires_grad_(grad)
        t_ref = bench(lambda: ref(d["x"], d["edge_index"], d["ntypes"], d["etypes"], eattr=d["eattr"], batch=d["batch"]), grad)
        t_or = bench(lambda: O.protein_lba_forward(Pg, d["x"], d["edge_index"], d["ntypes"], d["etypes"], d["eattr"]), grad)
        print(f"{threads} threads, {'fwd+bwd' if grad else 'forward'}: reference {t_ref * 1e3:.1f} ms, oracle {t_or * 1e3:.1f} ms, "
              f"oracle / reference = {t_or / t_ref:.2f}")
