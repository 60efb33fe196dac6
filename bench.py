#!/usr/bin/env python3
"""bench.py -- graph-pairs/sec of the CASTER-DTA encoder hot path on MI355X.

One "step" = one pass of the hot path over one batch of Davis-shaped synthetic
pairs resident in HBM: destination-sorted CSR build for both graphs, the protein
GVP encoder (node embed, 2 x [conv, node update], head) and the drug GINE encoder
(2 layers).  N>1: one process per GPU (torchrun), each rank runs its own shard of
pairs -- the encoders have no cross-pair term, so there is no data-path
collective (weak scaling); timing is barrier + synchronize on both sides, MAX
over ranks.

Prints ONE JSON line (see DESIGN.md "Measurement" for the definitions of
`roofline` and `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "caster-dta_amd"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: Davis, batch 64 pairs, CASTER-DTA(2,2), 300-residue radius graphs (4 A)
    "davis_b64": dict(pairs=64, length=300, thresh=4.0, thresh_type="dist"),
    # configs[3]: 1000-residue proteins, ~20 edges/residue (kNN 20), replicated to fill the device
    "long_graph_x64": dict(pairs=64, length=1000, thresh=20, thresh_type="num"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="davis_b64", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fwdbwd", choices=["fwd", "fwdbwd"],
                    help="fwdbwd (default, BASELINE config 2): forward + backward of both encoders incl. all "
                         "weight gradients; fwd: inference forward only")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--cache-csr", action="store_true", help="reuse the CSR tables across steps")
    ap.add_argument("--collate-csr", action="store_true",
                    help="per step, assemble the batch CSR from per-graph CSRs sorted once (SURVEY 8 f-2 wire format) "
                         "instead of sorting the batch's COO edge list; not the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--drug-stream", default="side", choices=["side", "main"],
                    help="diagnostic: 'main' runs the drug encoder on the protein stream (no overlap)")
    ap.add_argument("--only", default=None, choices=["protein", "drug"],
                    help="diagnostic: time one encoder alone (the reported metric needs both; never the default)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # Rehearsal on a box with fewer GPUs than ranks (BENCH_REHEARSAL=1): all ranks share cuda:0 and the
    # timing collective runs over gloo.  The real multi-GPU run is one rank per GPU over RCCL ("nccl").
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import davis_synth as ds
    from gvp_hip import ops
    import __graft_entry__ as entry

    wl = WORKLOADS[args.workload]
    model, state = entry._load_model(dev)
    pb, mb = ds.pair_batch(wl["pairs"], seed=rank, length=wl["length"], thresh=wl["thresh"],
                           thresh_type=wl["thresh_type"])
    to = lambda d: {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in d.items()}
    pdata_cpu, mdata_cpu = ds.to_torch(pb), ds.to_torch(mb)
    pdata, mdata = to(pdata_cpu), to(mdata_cpu)
    ops.CSR_CACHE_ENABLED = bool(args.cache_csr or args.collate_csr)
    side = torch.cuda.Stream(device=dev)
    collate = None
    if args.collate_csr:
        # wire format of SURVEY 8 f-2: every unique graph's CSR is sorted ONCE (outside the timed region, as a
        # dataset would at load time); a step assembles its batch tables from them in one launch per graph type
        def store_of(gb, data):
            ptr, eptr = [int(v) for v in gb.ptr], [int(v) for v in gb.eptr]
            ei = data["edge_index"]
            graphs = [(ei[:, eptr[g]:eptr[g + 1]] - ptr[g]).contiguous() for g in range(gb.num_graphs)]
            st = ops.CsrStore(graphs, [ptr[g + 1] - ptr[g] for g in range(gb.num_graphs)])
            return st, st.plan(range(gb.num_graphs))
        collate = (store_of(pb, pdata), store_of(mb, mdata))

    train = args.mode == "fwdbwd"
    prot_params = [p for p in model.protein_gnn.parameters() if p.numel()]
    drug_params = [p for p in model.molecule_gnn.parameters() if p.numel()]
    enc_params = prot_params + drug_params
    for p in model.parameters():
        p.requires_grad_(train)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    g_res = torch.randn(pb.num_nodes, 64, device=dev, generator=gen)     # upstream gradients of the embeddings
    g_atm = torch.randn(mb.num_nodes, 64, device=dev, generator=gen)

    def step():
        main_s = torch.cuda.current_stream()
        if args.only == "protein":                    # diagnostic: one encoder alone
            residues = model.protein_gnn(**pdata)
            return torch.autograd.grad([residues], prot_params, [g_res]) if train else residues
        if args.only == "drug":
            atoms = model.molecule_gnn(**mdata)
            return torch.autograd.grad([atoms], drug_params, [g_atm]) if train else atoms
        side.wait_stream(main_s)
        with torch.cuda.stream(side if args.drug_stream == "side" else main_s):   # drug graphs are tiny: run them beside the protein kernels
            if collate:
                collate[1][0].collate(collate[1][1], attach_to=mdata["edge_index"])
            atoms = model.molecule_gnn(**mdata)
        if collate:
            collate[0][0].collate(collate[0][1], attach_to=pdata["edge_index"])
        residues = model.protein_gnn(**pdata)
        main_s.wait_stream(side)
        if not train:
            return residues, atoms
        # backward of both encoders: every weight gradient (22,507 parameters) is produced
        return torch.autograd.grad([residues, atoms], enc_params, [g_res, g_atm])

    with torch.set_grad_enabled(train):
        out = step()
        torch.cuda.synchronize()
        graph = None
        if not args.no_graph and args.only != "drug" and args.drug_stream == "side":   # drug-only / same-stream diagnostics run eagerly
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = step()
        run = graph.replay if graph is not None else step

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)

        # ---- roofline leg: per-launch duration of the dominant kernel (conv_fwd), HIP events on its stream
        roof = None
        if rank == 0:
            ops.KERNEL_EVENTS = []
            for _ in range(min(args.steps, 50)):
                step()
            torch.cuda.synchronize()
            by = {}
            for (name, a, b) in ops.KERNEL_EVENTS:
                by.setdefault(name, []).append(a.elapsed_time(b) * 1e-3)
            ops.KERNEL_EVENTS = None
            N, E = pb.num_nodes, pb.num_edges
            conv_bytes = 224 * N + 156 * E            # algorithmic bytes of one conv launch
            # dominant kernel = the conv kernel with the largest share of the step (backward when training);
            # backward counted as 2x the forward bytes (fwd + bwd = 3x forward)
            if not by:                                # --only drug: no conv kernel to report
                by = {"conv_fwd": [float("nan")]}
            name = max(by, key=lambda k: sum(by[k]))
            times = sorted(by[name])
            nbytes = conv_bytes * (2 if name == "conv_bwd" else 1)
            avg = sum(times) / len(times)
            peak = 8000.0
            kname = {"conv_fwd": "conv_quad_kernel" if ops.VARIANT == "mfma" else "conv_fwd_kernel",
                     "conv_bwd": "conv_bwd_kernel"}[name]
            # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
            # separate runs, gfx950 correction applied; profiles/r01/pmc_traffic.json) -- same workload only
            traffic = None
            pmc = os.path.join(REPO, "profiles", "r01", "pmc_traffic.json")
            if args.workload == "davis_b64" and os.path.exists(pmc):
                k = "conv_bwd_kernel" if name == "conv_bwd" else "conv_quad_kernel"
                traffic = json.load(open(pmc))["kernels"].get(k, {}).get("hbm_bytes_per_launch")
            roof = dict(bound="hbm", achieved=round(nbytes / avg / 1e9, 1), peak=peak, unit="GB/s",
                        frac=round(nbytes / avg / 1e9 / peak, 4), traffic=traffic, kernel=kname,
                        avg_us=round(avg * 1e6, 2), median_us=round(times[len(times) // 2] * 1e6, 2),
                        bytes_per_launch=nbytes, launches=len(times),
                        other={k: round(sum(v) / len(v) * 1e6, 2) for k, v in by.items() if k != name})

    pairs_per_step = wl["pairs"] * world
    value = pairs_per_step * args.steps / dt

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import gvp_oracle as O
        pp = {k[len("protein_gnn.gnn_model."):]: v for k, v in state.items() if k.startswith("protein_gnn.gnn_model.")}
        mp = {k[len("molecule_gnn.gnn_model."):]: v for k, v in state.items() if k.startswith("molecule_gnn.gnn_model.")}

        def cpu_step():
            O.protein_lba_forward(pp, pdata_cpu["x"], pdata_cpu["edge_index"], pdata_cpu["ntypes"],
                                  pdata_cpu["etypes"], pdata_cpu["eattr"])
            O.molecule_gine_forward(mp, mdata_cpu["x"], mdata_cpu["edge_index"], mdata_cpu["ntypes"],
                                    mdata_cpu["etypes"], mdata_cpu["eattr"])
        with torch.no_grad():
            cpu_step()
            n, t0 = 0, time.perf_counter()
            while n < 3 or time.perf_counter() - t0 < args.cpu_seconds:
                cpu_step()
                n += 1
            cdt = time.perf_counter() - t0
        cpu = dict(value=round(wl["pairs"] * n / cdt, 1), unit="graph-pairs/sec", cores=torch.get_num_threads(),
                   kind="port", sample=f"{n} forward passes of the same {wl['pairs']}-pair batch through "
                   "oracle/gvp_oracle.py (torch CPU eager fp32)")

    if rank == 0:
        line = {
            "metric": "graph-pairs/sec (Davis-shaped protein+drug), encoders " + ("fwd+bwd" if train else "forward"),
            "value": round(value, 1), "unit": "graph-pairs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic Davis-shaped graphs (davis_synth, seed=rank); pretrained CASTER-DTA(2,2) weights",
            "config": {"workload": args.workload, "pairs_per_gpu": wl["pairs"], "residues_per_gpu": pb.num_nodes,
                       "protein_edges_per_gpu": pb.num_edges, "atoms_per_gpu": mb.num_nodes,
                       "drug_edges_per_gpu": mb.num_edges, "encoder": "CASTER-DTA(2,2)", "pass": args.mode,
                       "csr_build_in_step": ("collate" if args.collate_csr else not args.cache_csr), "hip_graph": graph is not None,
                       "kernels": ops.VARIANT,
                       "parallelism": f"pairs sharded x{world}, no collective"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
