#!/usr/bin/env python3
"""bench.py -- graph-pairs/sec of the CASTER-DTA encoder hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic pairs resident in HBM: destination-sorted CSR
build for both graphs, the protein GVP encoder (node embed, L x [conv + node update], head) and the drug GINE encoder,
forward + backward with every weight gradient, in TRAINING mode (dropout p = 0.2 drawn and applied every step, as
train_model.py:291 trains).  N > 1: one process per GPU (torchrun), each rank runs its own shard of pairs (weak
scaling) and the step ends with the two collectives data-parallel training of this model needs over RCCL: the
all-gather of the per-pair embeddings [pairs, 512] fp32 (joint_gnn.py:272) and ONE flat all-reduce of the encoder
weight gradients.  Timing: barrier + synchronize on both sides, MAX over ranks.

Prints ONE JSON line (DESIGN.md "Measurement" defines `roofline` and `cpu_baseline`).
"""
import argparse
import hashlib
import json
import os
import statistics
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
    # configs[2], one rank's share: KIBA, 32 pairs per GPU, protein lengths drawn from the KIBA sequence-length table
    "kiba_b32": dict(pairs=32, lengths="kiba", thresh=4.0, thresh_type="dist"),
    # configs[3]: 1000-residue proteins, 20 edges/residue (kNN 20), replicated x64 to fill the device
    "long_graph_x64": dict(pairs=64, length=1000, thresh=20, thresh_type="num"),
    # configs[4], one rank's share: BindingDB-scale synthetic (32 pairs, mean ~558 residues), CASTER-DTA(4,4) = four
    # conv layers in both encoders (seeded random weights: no (4,4) checkpoint ships); meant for --dtype bf16
    "bindingdb_b32_44": dict(pairs=32, lengths="bindingdb", thresh=4.0, thresh_type="dist", convs=4),
}
MIN_WARMUP = 30                # untimed steps before the timed region, whatever --warmup says (reported in config)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD = the fp32 vector rate
# Matrix-core work of the conv kernels per 16-edge tile (MFMA issues of 16x16x4 = 2,048 FLOP
# each) and the algorithmic MACs per edge behind them (edge embed 1,126 + message 2,543 = 3,669;
# the backward recomputes the forward and back-propagates data and weights: 3x).
# Counted in the gfx950 ISA of the kernels (v_mfma in one tile body): conv backward 230 since the edge embedding moved to
# its own once-per-step kernel (319 in round 1); forward 103 in the layer that derives the edge embedding, 71 in layers
# that read the store.  Useful MACs per edge of the conv backward: 3 x the message GVPs' 2,543.
# --dtype bf16 (tools/count_mfma.py): every GEMM with more than one k-step runs on v_mfma_f32_16x16x16_bf16 (8,192 FLOP,
# 4 passes = 16 cycles), one-step GEMMs stay on the fp32 instruction (32 cycles): (fp32 count, bf16 count) per tile.
MFMA_PER_TILE = {"f32": {"conv_fwd": (103, 0), "conv_bwd": (230, 0)}, "bf16": {"conv_fwd": (18, 26), "conv_bwd": (30, 54)}}
MAC_PER_EDGE = {"conv_fwd": 3669, "conv_bwd": 3 * 2543}
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (the K=32 forms; the K=16 form used here is half of it)
MATRIX_PIPE_CYCLES_PER_US = 1024 * 2400.0       # 1,024 SIMDs at 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configuration (1-based as DESIGN.md counts them): 2 = Davis batch 64 (default, "
                         "davis_b64 f32), 3 = KIBA 32 pairs per GPU (kiba_b32 f32), 4 = long-graph stress (long_graph_x64), "
                         "5 = BindingDB-scale CASTER-DTA(4,4) in bf16 (bindingdb_b32_44 --dtype bf16); --workload / --dtype override")
    ap.add_argument("--mode", default="fwdbwd", choices=["fwd", "fwdbwd"],
                    help="fwdbwd (default, BASELINE config 2): training step of both encoders (dropout on, forward + "
                         "backward incl. all weight gradients); fwd: inference forward only (eval mode)")
    ap.add_argument("--eval-mode", action="store_true",
                    help="fwdbwd without dropout (model.eval()): the round-1 measurement, for A/B only")
    ap.add_argument("--scope", default="encoders", choices=["encoders", "joint"],
                    help="encoders (default: the metric of BASELINE.json); joint: the whole JointGNN training step "
                         "(encoders + cross-attention head + MSE loss) with enable_pair_parallel() on N > 1")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--cache-csr", action="store_true", help="reuse the CSR tables across steps")
    ap.add_argument("--collate-csr", action="store_true",
                    help="per step, assemble the batch CSR from per-graph CSRs sorted once (SURVEY 8 f-2 wire format) "
                         "instead of sorting the batch's COO edge list; not the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--drug-stream", default="side", choices=["side", "main"],
                    help="diagnostic: 'main' runs the drug encoder on the protein stream (no overlap)")
    ap.add_argument("--compile", action="store_true",
                    help="--scope joint: run the model through torch.compile(dynamic=True), what train_model.py:422 does "
                         "(the encoders and the attention core stay single custom-op nodes)")
    ap.add_argument("--autocast", default=None, choices=["bf16", "fp16"],
                    help="--scope joint: run the step under torch.autocast as train_model.py:561 does (the torch head then "
                         "computes in half precision; the encoder ops' autocast rule keeps them in fp32)")
    ap.add_argument("--compile-graph", action="store_true",
                    help="with --compile: additionally capture the compiled step into a HIP graph (experiment)")
    ap.add_argument("--two-lane-head", action="store_true",
                    help="--scope joint: JointGNN.two_stream_head = True (the atom side of the head on a side stream)")
    ap.add_argument("--tunable-gemms", action="store_true",
                    help="--scope joint: let PyTorch's TunableOp pick the library GEMM solution per shape (torch.cuda.tunable; "
                         "tuned during the untimed warm-up; the results file goes to $TMPDIR)")
    ap.add_argument("--issue-order", default="protein", choices=["protein", "drug"],
                    help="which encoder's launches are issued (and captured) first in a step (A/B)")
    ap.add_argument("--drug-priority", type=int, default=0,
                    help="diagnostic: priority of the drug encoder's side stream (-1 = high)")
    ap.add_argument("--drug-atoms", type=int, default=0,
                    help="diagnostic: every drug graph gets exactly this many atoms (0 = the Davis-like size distribution); "
                         "with 1-2 atoms the drug chain is launches only: what it then adds to the step is queue / graph overhead")
    ap.add_argument("--gine-bwd-wgs", type=int, default=0,
                    help="diagnostic: workgroup cap of the GINE backward (0 = library default of 16)")
    ap.add_argument("--only", default=None, choices=["protein", "drug"],
                    help="diagnostic: time one encoder alone (the reported metric needs both; never the default)")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="activation storage of the protein encoder (bf16 = BASELINE config 5: bf16 storage / fp32 "
                         "accumulate; gradients and weights stay fp32; the drug encoder stays fp32 storage)")
    ap.add_argument("--epoch", default="both", choices=["off", "nominal", "real", "both"],
                    help="second leg (N = 1, --scope encoders, davis_b64 only): ONE EPOCH the way train_model.py:548-587 drives "
                         "the model -- 329 DIFFERENT batches of 64 pairs drawn from 442 unique proteins x 68 unique drugs (the "
                         "Davis table sizes), a new edge_index every step, eager (no HIP graph); 'nominal' = 300-residue "
                         "proteins, 'real' = lengths drawn from the Davis sequence-length table (mean 789), 'both' (default) = the "
                         "nominal epoch in config.epoch and the real-length one (eager + bucketed legs) in config.epoch_real")
    ap.add_argument("--epoch-steps", type=int, default=329, help="batches in the epoch leg (Davis: 21,039 train pairs / 64)")
    ap.add_argument("--cpu-runs", type=int, default=20, help="timed CPU-baseline runs per thread count (median)")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU sample (0 = choose for ~10-30 s)")
    args = ap.parse_args()
    wl, dt = {2: ("davis_b64", "f32"), 3: ("kiba_b32", "f32"), 4: ("long_graph_x64", "f32"), 5: ("bindingdb_b32_44", "bf16")}[args.config]
    args.workload = args.workload or wl
    args.dtype = args.dtype or dt
    return args


def kernel_source_digest():
    h = hashlib.sha256()
    d = os.path.join(REPO, "caster-dta_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_copy_gbs(dev, nbytes=1 << 30, reps=5):
    """Device-to-device copy bandwidth of this box (read + write bytes / time, best of `reps`), torch's copy kernel."""
    try:
        src = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        best = None
        for _ in range(reps + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dst.copy_(src)
            b.record()
            b.synchronize()
            t = a.elapsed_time(b) * 1e-3
            best = t if best is None or t < best else best
        del src, dst
        return round(2 * nbytes / best / 1e9, 1)
    except Exception:
        return None


def committed_traffic(workload, kernel, dtype="f32"):
    """HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    gfx950 correction applied: tools/pmc_summarise.py).  Only reported when the file was measured on the SAME
    kernel sources and workload; otherwise null (a stale number is worse than none)."""
    tag = workload if dtype == "f32" else f"{workload}_{dtype}"
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(REPO, "profiles", rnd, f"pmc_traffic_{tag}.json")
        if not os.path.exists(path) and workload == "davis_b64":
            path = os.path.join(REPO, "profiles", rnd, "pmc_traffic.json")
        if not os.path.exists(path):
            continue
        doc = json.load(open(path))
        if doc.get("kernel_source_digest") != kernel_source_digest():
            continue
        return doc["kernels"].get(kernel, {}).get("hbm_bytes_per_launch")
    return None


def launcher_command(argv, gpus, port=None):
    """Command line that starts `gpus` ranks of this script on one node (what the driver uses for N > 1)."""
    if port is None:
        import socket
        with socket.socket() as s:                 # a free port now; torchrun binds it a moment later
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: this process -- which has touched no GPU and
    never will -- starts the N ranks as CHILD processes, relays their output (rank 0 prints the one JSON line) and
    returns their exit status.  Never an exec: the ranks are children, the parent only waits."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(args.gpus, 1))))
    cmd = launcher_command(argv, args.gpus)
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` "
                         f"(self-launching) or under torch.distributed.run with --nproc-per-node {args.gpus}")
    if os.environ.get("BENCH_DRY_RUN") == "1":
        # Control flow of the launch path without a GPU (tests/test_host_logic.py): rendezvous over gloo, the barrier /
        # MAX-over-ranks pattern of the timed region on a stand-in duration, ONE line from rank 0.  No measurement.
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
            dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "max_over_ranks": float(t), "dtype": args.dtype,
                              "config": {"workload": args.workload, "baseline_config": args.config, "ranks": world}}))
        return
    # Rehearsal on a box with fewer GPUs than ranks (BENCH_REHEARSAL=1): all ranks share cuda:0 and the
    # collectives run over gloo on host copies.  The real multi-GPU run is one rank per GPU over RCCL ("nccl").
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    # BENCH_FORCE_COLLECTIVES=1: issue the N > 1 collectives at world size 1 too (exercises the RCCL calls on one GPU)
    force_coll = os.environ.get("BENCH_FORCE_COLLECTIVES") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or force_coll:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    collectives = dist is not None

    import davis_synth as ds
    from gvp_hip import ops, autograd_ops
    autograd_ops.GINE_BWD_WORKGROUPS = args.gine_bwd_wgs
    import __graft_entry__ as entry

    wl = WORKLOADS[args.workload]
    train = args.mode == "fwdbwd"
    dropout_on = train and not args.eval_mode
    convs = wl.get("convs", 2)
    model, state = entry._load_model(dev, num_convs=convs)
    model.train(dropout_on)
    lengths = ds.real_lengths(wl["lengths"], wl["pairs"], seed=1000 + rank) if "lengths" in wl else None
    pb, mb = ds.pair_batch(wl["pairs"], seed=rank, length=wl.get("length", 300), thresh=wl["thresh"],
                           thresh_type=wl["thresh_type"], lengths=lengths)
    if args.drug_atoms > 0:
        import numpy as np
        rng_d = np.random.default_rng(77 + rank)
        mb = ds.collate([ds.drug_graph(rng_d, n_atoms=args.drug_atoms) for _ in range(wl["pairs"])])
    to = lambda d: {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in d.items()}
    pdata_cpu, mdata_cpu = ds.to_torch(pb), ds.to_torch(mb)
    pdata, mdata = to(pdata_cpu), to(mdata_cpu)
    if args.dtype == "bf16":                      # features held in HBM as bf16 (a dataset stored that way)
        pdata = dict(pdata, x=tuple(t.bfloat16() for t in pdata["x"]), eattr=tuple(t.bfloat16() for t in pdata["eattr"]))
    # joint scope: graph offsets as PyG Batch objects carry them (the head then has no data-dependent shape)
    jp = dict(pdata, ptr=torch.as_tensor(pb.ptr).to(dev))
    jm = dict(mdata, ptr=torch.as_tensor(mb.ptr).to(dev))
    ops.CSR_CACHE_ENABLED = bool(args.cache_csr or args.collate_csr)
    side = torch.cuda.Stream(device=dev, priority=args.drug_priority)
    collate = None

    # wire format of SURVEY 8 f-2: every unique graph's CSR is sorted ONCE (outside the timed region, as a
    # dataset would at load time); a step assembles its batch tables from them in one launch per graph type
    def store_of(gb, data):
        ptr, eptr = [int(v) for v in gb.ptr], [int(v) for v in gb.eptr]
        ei = data["edge_index"]
        graphs = [(ei[:, eptr[g]:eptr[g + 1]] - ptr[g]).contiguous() for g in range(gb.num_graphs)]
        st = ops.CsrStore(graphs, [ptr[g + 1] - ptr[g] for g in range(gb.num_graphs)])
        return st, st.plan(range(gb.num_graphs))
    if args.collate_csr:
        collate = (store_of(pb, pdata), store_of(mb, mdata))

    if args.two_lane_head:
        model.two_stream_head = True
    if args.tunable_gemms:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "caster_gvp_tunableop.csv"))    # (results file: scratch)
    prot_params = [p for p in model.protein_gnn.parameters() if p.numel()]
    drug_params = [p for p in model.molecule_gnn.parameters() if p.numel()]
    enc_params = prot_params + drug_params
    n_enc = sum(p.numel() for p in enc_params)
    for p in model.parameters():
        p.requires_grad_(train)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    g_res = torch.randn(pb.num_nodes, 64, device=dev, generator=gen)     # upstream gradients of the embeddings
    if args.dtype == "bf16":
        g_res = g_res.bfloat16()
    g_atm = torch.randn(mb.num_nodes, 64, device=dev, generator=gen)
    target = torch.randn(wl["pairs"] * world, 1, device=dev, generator=gen)
    # payloads of the N > 1 collectives in --scope encoders: the tensor joint_gnn.py:272 all-gathers (synthetic
    # values, real shape / dtype) and the flat gradient bucket of both encoders (real gradients)
    pair_local = torch.randn(wl["pairs"], 512, device=dev, generator=gen)
    pair_all = torch.empty(wl["pairs"] * world, 512, device=dev)
    grad_bucket = torch.zeros(n_enc, device=dev)
    all_params = [p for p in model.parameters() if p.numel()]
    pre_ids = {id(p) for p in model.pre_gather_parameters()}
    pre_idx = [i for i, p in enumerate(all_params) if id(p) in pre_ids]      # rank-local gradients under pair parallelism
    joint_bucket = torch.zeros(sum(all_params[i].numel() for i in pre_idx), device=dev)
    if args.scope == "joint" and collectives and (world > 1 or force_coll):
        if rehearsal:
            raise SystemExit("--scope joint gathers device tensors inside the model (all_gather_into_tensor): it needs RCCL, "
                             "the gloo rehearsal (BENCH_REHEARSAL=1) covers --scope encoders only")
        model.enable_pair_parallel()

    def encoders_step():
        main_s = torch.cuda.current_stream()
        if args.only == "protein":                    # diagnostic: one encoder alone
            residues = model.protein_gnn(**pdata)
            return torch.autograd.grad([residues], prot_params, [g_res]) if train else residues
        if args.only == "drug":
            atoms = model.molecule_gnn(**mdata)
            return torch.autograd.grad([atoms], drug_params, [g_atm]) if train else atoms
        side.wait_stream(main_s)
        # the protein chain is the critical path: it is issued first and stays on the launch stream (a HIP graph keeps
        # the first-captured branch on the launch queue; the other branch pays the cross-queue joins)
        def drug_forward():
            with torch.cuda.stream(side if args.drug_stream == "side" else main_s):   # drug graphs are tiny: run them beside the protein kernels
                if collate:
                    collate[1][0].collate(collate[1][1], attach_to=mdata["edge_index"])
                return model.molecule_gnn(**mdata)
        if args.issue_order == "drug":
            atoms = drug_forward()
        if collate:
            collate[0][0].collate(collate[0][1], attach_to=pdata["edge_index"])
        residues = model.protein_gnn(**pdata)
        if args.issue_order != "drug":
            atoms = drug_forward()
        if not train:
            main_s.wait_stream(side)
            return residues, atoms
        # backward of both encoders: every weight gradient (22,507 parameters) is produced.  Two engine calls, so that
        # the protein backward is again the first-issued branch and the drug backward stays on its side stream
        def drug_backward():
            if args.drug_stream == "side":
                with torch.cuda.stream(side):
                    return torch.autograd.grad([atoms], drug_params, [g_atm])
            return torch.autograd.grad([atoms], drug_params, [g_atm])
        if args.issue_order == "drug":
            gd = drug_backward()
        gp = torch.autograd.grad([residues], prot_params, [g_res])
        if args.issue_order != "drug":
            gd = drug_backward()
        main_s.wait_stream(side)
        return gp + gd

    jmodel = torch.compile(model, dynamic=True) if (args.compile and args.scope == "joint") else model

    def joint_step():
        if collate:
            collate[1][0].collate(collate[1][1], attach_to=mdata["edge_index"])
            collate[0][0].collate(collate[0][1], attach_to=pdata["edge_index"])
        if args.autocast:
            with torch.autocast("cuda", dtype=torch.bfloat16 if args.autocast == "bf16" else torch.float16):
                pred, _ = jmodel(jp, jm)
            pred = pred.float()
        else:
            pred, _ = jmodel(jp, jm)
        if not train:
            return pred
        loss = torch.nn.functional.mse_loss(pred, target[:pred.shape[0]])
        # every gradient of the model (764,396 parameters); autograd.grad instead of .backward() keeps the step free of
        # AccumulateGrad nodes, so it captures into a HIP graph like the encoder step
        return torch.autograd.grad(loss, all_params)

    step = encoders_step if args.scope == "encoders" else joint_step

    def comm(out):
        """The per-step collectives of data-parallel training (issued eagerly after the compute of the step)."""
        if not collectives:
            return
        if args.scope == "joint":           # the all-gather ran inside the model; what is left is the flat gradient reduce
            if train and not rehearsal:
                torch.cat([out[i].reshape(-1) for i in pre_idx], out=joint_bucket)
                dist.all_reduce(joint_bucket)
            return
        if rehearsal:                                  # gloo on host copies (cuda:0 is shared by all ranks)
            gathered = [torch.empty(pair_local.shape) for _ in range(world)]
            dist.all_gather(gathered, pair_local.cpu())
            if train:
                flat = torch.cat([g.reshape(-1) for g in out]).cpu()
                dist.all_reduce(flat)
            return
        # the two collectives are independent: the gather runs on the side stream while the gradient bucket is packed and
        # reduced on the main one (each is ~12 us of launch latency at these payloads; back to back they cost 24)
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            dist.all_gather_into_tensor(pair_all, pair_local)
        if train:
            torch.cat([g.reshape(-1) for g in out], out=grad_bucket)
            dist.all_reduce(grad_bucket)
        main_s.wait_stream(side)

    with torch.set_grad_enabled(train):
        out = step()
        torch.cuda.synchronize()
        graph = None
        # both scopes replay from a HIP graph (earlier in round 2 a captured joint step faulted on replay; with the
        # per-model weight-image handling and the in-graph image build it captures and replays cleanly).  torch.compile
        # and the multi-rank joint scope (collectives inside the model) stay eager.
        # (the eval-mode joint forward also stays eager: it materialises the padded attention weights, whose shape needs a
        # host-side maximum)
        eager_joint = args.scope == "joint" and ((args.compile and not args.compile_graph) or not train or
                                                 (collectives and (world > 1 or force_coll)))
        if not args.no_graph and not eager_joint and args.drug_stream == "side":
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):         # the stream the warm-up ran on: its per-stream CSR counters exist
                out = step()

        def run():
            o = out
            if graph is not None:
                graph.replay()
            else:
                o = step()
            comm(o)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # W untimed warm-up steps as asked, topped up to MIN_WARMUP: a 20-step timed region lasts 5 ms, and the first
        # few dozen replays after start-up run at ramping clocks (0.276 ms/step at W = 5 vs 0.263 at W = 30)
        for _ in range(max(args.warmup, MIN_WARMUP)):
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

        # ---- roofline leg: per-launch duration of the dominant conv kernel, HIP events on its launch stream
        roof = None
        if rank == 0:
            # the library's opt-in diagnostics bracket every conv-layer launch of the whole-pass entry points with HIP
            # events on the launch stream (cgvp_debug_kernel_timing; eager steps, outside the timed region above)
            import ctypes
            from gvp_hip import _lib
            L_ = _lib.lib()
            L_.cgvp_debug_kernel_timing(1)
            for _ in range(min(args.steps, 50)):
                step()
            torch.cuda.synchronize()
            L_.cgvp_debug_kernel_timing(0)
            cap = 4096
            ms, kinds = (ctypes.c_float * cap)(), (ctypes.c_int32 * cap)()
            n_timed = min(L_.cgvp_debug_kernel_times(ms, kinds, cap), cap)
            by = {}
            for i in range(n_timed):
                by.setdefault("conv_bwd" if kinds[i] == 1 else "conv_fwd", []).append(ms[i] * 1e-3)
            N, E = pb.num_nodes, pb.num_edges
            # algorithmic bytes of one conv launch (bf16 storage: float terms halved, 16 B/edge of indices kept)
            conv_bytes = (224 * N + 156 * E) if args.dtype == "f32" else (112 * N + 86 * E)
            # dominant kernel = the conv kernel with the largest share of the step (backward when training);
            # backward counted as 2x the forward bytes (fwd + bwd = 3x forward)
            if by:
                name = max(by, key=lambda k: sum(by[k]))
                times = sorted(by[name])
                nbytes = conv_bytes * (2 if name == "conv_bwd" else 1)
                avg = sum(times) / len(times)
                kname = {"conv_fwd": "conv_quad_kernel" if ops.VARIANT == "mfma" else "conv_fwd_kernel",
                         "conv_bwd": "conv_bwd_kernel" if os.environ.get("CGVP_CONV_BWD") == "1" else "conv_bwd2_kernel"}[name]
                hbm = dict(achieved=round(nbytes / avg / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                           frac=round(nbytes / avg / 1e9 / PEAK_HBM_GBS, 4), floor_us=round(nbytes / PEAK_HBM_GBS / 1e3, 2),
                           device_copy_gbs=measured_copy_gbs(dev))      # the box's own copy bandwidth, for context
                tiles = (E + 15) // 16                # 16 sorted edges per wave tile
                n32, n16 = MFMA_PER_TILE[args.dtype][name]
                issued = (n32 * 2048.0 + n16 * 8192.0) * tiles
                useful = 2.0 * MAC_PER_EDGE[name] * E
                peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
                mfma = dict(achieved=round(useful / avg / 1e12, 2), issued=round(issued / avg / 1e12, 2),
                            peak=peak, unit="TFLOP/s",
                            frac=round(useful / avg / 1e12 / peak, 4),
                            frac_issued=round(issued / avg / 1e12 / peak, 4),
                            # time the matrix pipes need for the instructions issued (32 cycles per fp32, 16 per bf16 MFMA)
                            floor_us=round(tiles * (n32 * 32.0 + n16 * 16.0) / MATRIX_PIPE_CYCLES_PER_US, 2),
                            mfma_per_16_edges=n32 + n16, mfma_f32_bf16=[n32, n16], useful_mac_per_edge=MAC_PER_EDGE[name])
                # the bound is whichever floor is higher for THIS kernel at THIS size
                bound = "mfma" if mfma["floor_us"] > hbm["floor_us"] else "hbm"
                top = mfma if bound == "mfma" else hbm
                roof = dict(bound=bound, achieved=top["achieved"], peak=top["peak"], unit=top["unit"], frac=top["frac"],
                            traffic=committed_traffic(args.workload, kname, args.dtype), kernel=kname,
                            avg_us=round(avg * 1e6, 2), median_us=round(times[len(times) // 2] * 1e6, 2),
                            bytes_per_launch=nbytes, launches=len(times), hbm=hbm, mfma=mfma,
                            other={k: round(sum(v) / len(v) * 1e6, 2) for k, v in by.items() if k != name})

    pairs_per_step = wl["pairs"] * world
    value = pairs_per_step * args.steps / dt

    # ---- the same captured step with the batch wire format of SURVEY 8 f-2 instead of the in-step CSR build: per-graph
    # CSR tables sorted once at "dataset load", ONE collate launch per encoder per step (the headline above keeps the
    # build in the step: the reference hands over a new COO list every step)
    collated = None
    if rank == 0 and world == 1 and graph is not None and args.scope == "encoders" and args.only is None \
            and not args.collate_csr and not args.cache_csr and not collectives and args.epoch != "off":
        collate = (store_of(pb, pdata), store_of(mb, mdata))
        old_cache, ops.CSR_CACHE_ENABLED = ops.CSR_CACHE_ENABLED, True
        try:
            with torch.set_grad_enabled(train):
                s2 = torch.cuda.Stream(device=dev)
                s2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s2):
                    for _ in range(3):
                        step()
                torch.cuda.current_stream().wait_stream(s2)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=s2):
                    step()
                for _ in range(MIN_WARMUP):
                    g2.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    g2.replay()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
            collated = {"what": "the same captured step with per-graph CSR tables (sorted once, outside the step) collated "
                                "per batch in one launch per encoder (gvp_hip.ops.CsrStore) instead of the in-step build",
                        "ms_per_step": round(dt2 / args.steps * 1e3, 4), "pairs_per_s": round(pairs_per_step * args.steps / dt2, 1)}
            del g2
        finally:
            collate = None
            ops.CSR_CACHE_ENABLED = old_cache

    epoch = epoch_real = None
    if rank == 0 and world == 1 and args.epoch != "off" and args.scope == "encoders" and args.workload == "davis_b64" \
            and args.only is None and not collectives:
        if args.epoch in ("real", "both"):                       # first: its fused-parameters leg (nominal only) cannot be undone
            epoch_real = epoch_leg(args, model, dev, train, prot_params, drug_params, variant="real", light=args.epoch == "both")
        if args.epoch in ("nominal", "both"):
            epoch = epoch_leg(args, model, dev, train, prot_params, drug_params, variant="nominal")
        elif args.epoch == "real":
            epoch, epoch_real = epoch_real, None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # a reported baseline of the N = 1 line only
        cpu = cpu_baseline(args, wl, state, pb, mb, train)

    if rank == 0:
        what = "encoders" if args.scope == "encoders" else "JointGNN"
        par = f"pairs sharded x{world}"
        if collectives and args.scope == "encoders":
            par += (f"; per step over RCCL: all-gather [{wl['pairs']},512] fp32 pair embeddings (synthetic payload) + "
                    f"all-reduce of {n_enc} encoder gradients" if not rehearsal else "; gloo rehearsal of the collectives")
        elif collectives:
            par += "; enable_pair_parallel(): all-gather of pair embeddings + flat all-reduce of pre-gather gradients"
        else:
            par += ", no collective"
        line = {
            "metric": f"graph-pairs/sec (Davis-shaped protein+drug), {what} " + ("fwd+bwd" if train else "forward"),
            "value": round(value, 1), "unit": "graph-pairs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic Davis-shaped graphs (davis_synth, seed=rank); " + (
                "pretrained CASTER-DTA(2,2) weights" if convs == 2 else f"seeded random CASTER-DTA({convs},{convs}) weights"),
            "config": {"workload": args.workload, "pairs_per_gpu": wl["pairs"], "residues_per_gpu": pb.num_nodes,
                       "protein_edges_per_gpu": pb.num_edges, "atoms_per_gpu": mb.num_nodes,
                       "drug_edges_per_gpu": mb.num_edges, "encoder": f"CASTER-DTA({convs},{convs})", "pass": args.mode,
                       "activation_storage": "bf16 (protein encoder: bf16 activations in HBM and bf16 matrix-core operands, fp32 accumulate, fp32 weights and gradient buffers)" if args.dtype == "bf16" else "fp32",
                       "scope": args.scope, "untimed_warmup_steps": max(args.warmup, MIN_WARMUP), "torch_compile": bool(args.compile and args.scope == "joint"), "tunable_gemms": bool(args.tunable_gemms), "two_lane_head": bool(args.two_lane_head), "autocast": args.autocast if args.scope == "joint" else None, "train_mode": bool(dropout_on), "dropout_p": 0.2 if dropout_on else 0.0,
                       "csr_build_in_step": ("collate" if args.collate_csr else not args.cache_csr),
                       "hip_graph": graph is not None, "kernels": ops.VARIANT, "parallelism": par,
                       "baseline_config": args.config, "rccl_ranks": (dist.get_world_size() if (dist is not None and not rehearsal) else 0),
                       "timed_region": ("replays of ONE captured step on one batch (CSR build, dropout draw and weight-image build "
                                        "inside every replay)" if graph is not None else "eager steps on one batch"),
                       "collated_csr": collated, "epoch": epoch, "epoch_real": epoch_real},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


EPOCH_PASSES = 3      # timed passes of every host-bound epoch leg; the MEDIAN pass is reported, all of them listed


def _timed_passes(run_epoch, steps, passes=EPOCH_PASSES):
    """(wall ms/step, host-issue ms/step) of the median pass + every pass's wall ms/step.  Host-bound legs swing by up to
    2x between passes on a GPU box whose host cores are shared (0.33-0.79 ms per step measured back to back, one process):
    one pass is not a measurement."""
    res = []
    for _ in range(passes):
        t0 = time.perf_counter()
        run_epoch()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0, t_issue))
    med = sorted(res)[len(res) // 2]
    return med[0], med[1], [round(r[0] / steps * 1e3, 4) for r in res]


def epoch_leg(args, model, dev, train, prot_params, drug_params, variant="nominal", light=False):
    """BASELINE config 2 as stated ("Davis full epoch ... batch 64"): every step a DIFFERENT batch -- other proteins, other
    N / E, a new edge_index -- launched eagerly through the nn.Module API, exactly the call pattern of
    train_model.py:548-587 (forward of both encoders, backward with every weight gradient, dropout on).  The batches are
    assembled on the device beforehand (inputs resident in HBM) from 442 unique protein graphs and 68 unique drug graphs,
    the sizes of the Davis tables (data/deepdta_data/davis: 442 proteins x 68 ligands).  Two numbers: the wall time per
    step (host issue + device, synchronised once at the end of the epoch) and the host time to issue a step."""
    import numpy as np
    import davis_synth as ds
    rng = np.random.default_rng(2024)
    n_prot, n_drug, B, steps = 442, 68, 64, args.epoch_steps
    if variant == "real":
        lengths = ds.real_lengths("davis", n_prot, seed=7)
    else:
        lengths = [300] * n_prot
    t_gen = time.perf_counter()
    prots = [ds.protein_graph(int(L), rng, 4.0, "dist") for L in lengths]
    drugs = [ds.drug_graph(rng) for _ in range(n_drug)]
    f = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    P = [dict(x_s=f(g["x_s"]), x_v=f(g["x_v"]), ei=f(g["edge_index"]), e_s=f(g["e_s"]), e_v=f(g["e_v"]), nt=f(g["ntypes"]),
              et=f(g["etypes"]), n=g["x_s"].shape[0]) for g in prots]
    D = [dict(x=f(g["x_s"]), ei=f(g["edge_index"]), ea=f(g["e_s"]), nt=f(g["ntypes"]), et=f(g["etypes"]), n=g["x_s"].shape[0])
         for g in drugs]
    batches = []
    for _ in range(steps):
        pi, di = rng.integers(0, n_prot, B), rng.integers(0, n_drug, B)
        off = np.concatenate([[0], np.cumsum([P[i]["n"] for i in pi])])
        pd = dict(x=(torch.cat([P[i]["x_s"] for i in pi]), torch.cat([P[i]["x_v"] for i in pi])),
                  edge_index=torch.cat([P[i]["ei"] + int(o) for i, o in zip(pi, off)], 1).contiguous(),
                  ntypes=torch.cat([P[i]["nt"] for i in pi]), etypes=torch.cat([P[i]["et"] for i in pi]),
                  eattr=(torch.cat([P[i]["e_s"] for i in pi]), torch.cat([P[i]["e_v"] for i in pi])))
        doff = np.concatenate([[0], np.cumsum([D[i]["n"] for i in di])])
        md = dict(x=torch.cat([D[i]["x"] for i in di]),
                  edge_index=torch.cat([D[i]["ei"] + int(o) for i, o in zip(di, doff)], 1).contiguous(),
                  ntypes=torch.cat([D[i]["nt"] for i in di]), etypes=torch.cat([D[i]["et"] for i in di]),
                  eattr=torch.cat([D[i]["ea"] for i in di]))
        if args.dtype == "bf16":
            pd = dict(pd, x=tuple(t.bfloat16() for t in pd["x"]), eattr=tuple(t.bfloat16() for t in pd["eattr"]))
        batches.append((pd, md, int(off[-1]), int(doff[-1]), torch.as_tensor(off).to(dev), torch.as_tensor(doff).to(dev)))
    maxn, maxa = max(b[2] for b in batches), max(b[3] for b in batches)
    gen = torch.Generator(device=dev).manual_seed(99)
    g_res = torch.randn(maxn, 64, device=dev, generator=gen)
    if args.dtype == "bf16":
        g_res = g_res.bfloat16()
    g_atm = torch.randn(maxa, 64, device=dev, generator=gen)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    side = torch.cuda.Stream(device=dev)

    def step(pd, md, n, na, *_):
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        residues = model.protein_gnn(**pd)
        with torch.cuda.stream(side):
            atoms = model.molecule_gnn(**md)
        if not train:
            main_s.wait_stream(side)
            return residues, atoms
        # what `loss.backward()` (train_model.py:570) does to the encoders: one backward pass in ACCUMULATE mode (every
        # parameter's .grad is set; each node runs on its forward's stream), then the reset optimizer.zero_grad() does
        torch.autograd.backward([residues, atoms], [g_res[:n], g_atm[:na]])
        main_s.wait_stream(side)
        for q in enc_leaves:
            q.grad = None

    enc_leaves = list(prot_params) + list(drug_params)
    from gvp_hip import _lib as _l
    br = _l.bridge()
    with torch.set_grad_enabled(train):
        # untimed first pass over the epoch's batches: the timed passes are later epochs, i.e. the caching allocator already
        # holds blocks for every workspace size the epoch asks for -- the same footing as the bucketed leg, whose graphs
        # are captured in an untimed pass
        for b in batches:
            step(*b)
        torch.cuda.synchronize()
        fast0 = br.fast_leaf_passes() if br is not None else 0

        def run_epoch():
            for b in batches:
                step(*b)

        t_all, t_issue, eager_passes = _timed_passes(run_epoch, steps)
        fast_passes = (br.fast_leaf_passes() - fast0) if br is not None else 0
    # ---- the WHOLE model over the same epoch (encoders + cross-attention head + MSE loss, all 764k gradients):
    # eager `loss.backward()` and shape-bucketed whole-step graphs (gvp_hip.graphed.GraphedTrainStep)
    joint = joint_epoch(args, model, dev, batches, B, steps) if (train and args.dtype != "bf16" and not light) else None
    # ---- the same epoch replayed from shape-bucketed HIP graphs (gvp_hip.graphed): per step ONE staging launch that
    # copies the batch into the bucket's padded static buffers + one graph replay
    graphed = None
    if train:
        from gvp_hip.graphed import GraphedEncoderStep
        runner = GraphedEncoderStep(model.protein_gnn, model.molecule_gnn)
        with torch.set_grad_enabled(True):
            for b in batches[:MIN_WARMUP]:
                runner.run(b[0], b[1], g_res[:b[2]], g_atm[:b[3]])
            first = {k: v.steps for k, v in runner.buckets.items()}
            for b in batches:                                   # untimed pass: captures every bucket the epoch visits
                runner.run(b[0], b[1], g_res[:b[2]], g_atm[:b[3]])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for b in batches:
                runner.run(b[0], b[1], g_res[:b[2]], g_atm[:b[3]])
            t_gi = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_ga = time.perf_counter() - t0
        sizes = sorted({k[0] for k in runner.buckets})
        graphed = {"what": "same epoch, every step = 1 staging launch (cgvp_stage_buffers) + 1 replay of the captured step of "
                           "the batch's shape bucket (sizes rounded up by <= 12.5 %; padded nodes isolated, padded edges dropped)",
                   "ms_per_step": round(t_ga / steps * 1e3, 4), "host_issue_ms_per_step": round(t_gi / steps * 1e3, 4),
                   "pairs_per_s": round(B * steps / t_ga, 1), "buckets": len(runner.buckets),
                   "bucket_sizes_residues_edges_atoms_bonds": [list(k) for k in sizes][:12]}
    # ---- the eager epoch once more with the protein encoder's parameters fused into ONE autograd leaf
    # (JointGNN.fuse_encoder_parameters, opt-in): what the 74 per-tensor leaves cost on the host.  Last: cannot be undone.
    fused = None
    gm = model.protein_gnn.gnn_model
    if train and not light and args.dtype != "bf16" and getattr(gm, "_onehot_ntypes", False) and gm._onehot_etypes:
        model.fuse_encoder_parameters()
        prot_params = [p for p in model.protein_gnn.parameters() if p.numel()]
        drug_params = [p for p in model.molecule_gnn.parameters() if p.numel()]
        enc_leaves[:] = list(prot_params) + list(drug_params)
        with torch.set_grad_enabled(True):
            for b in batches:                                   # untimed first pass, as above
                step(*b)
            torch.cuda.synchronize()
            t_fa, t_fi, fused_passes = _timed_passes(run_epoch, steps)
        fused = {"passes_ms": fused_passes, "what": "same eager epoch after JointGNN.fuse_encoder_parameters(): each encoder's parameter arena is its "
                         "one trainable leaf (checkpoint keys unchanged)", "leaves": len(prot_params) + len(drug_params),
                 "ms_per_step": round(t_fa / steps * 1e3, 4), "host_issue_ms_per_step": round(t_fi / steps * 1e3, 4),
                 "pairs_per_s": round(B * steps / t_fa, 1)}
    edges = [int(b[0]["edge_index"].shape[1]) for b in batches]
    ret_extra = {"eager_fused_parameters": fused, "joint": joint, "passes_ms": eager_passes,
                 # backward passes of the eager leg whose node wrote the leaves' .grad itself (csrc/torch_bridge.cpp LeafScatter)
                 "eager_backward_passes_without_leaf_tasks": int(fast_passes)}
    return {**_epoch_summary(args, variant, steps, B, n_prot, n_drug, t_all, t_issue, batches, edges, t_gen, graphed), **ret_extra}


def joint_epoch(args, model, dev, batches, B, steps):
    """The loop of train_model.py:548-587 at MODEL scope over the epoch's different batches: forward of the whole JointGNN,
    MSE loss, backward -- (a) eagerly through the nn.Module API with `loss.backward()` + the optimizer's gradient reset,
    (b) from shape-bucketed whole-step HIP graphs (one staging launch + one replay per step)."""
    from gvp_hip.graphed import GraphedTrainStep
    gen = torch.Generator(device=dev).manual_seed(7)
    target = torch.randn(B, 1, device=dev, generator=gen)
    loss_fn = torch.nn.functional.mse_loss
    dicts = [(dict(b[0], ptr=b[4]), dict(b[1], ptr=b[5])) for b in batches]

    leaves = [p for p in model.parameters()]

    def eager(pd, md):
        pred, _ = model(pd, md)
        loss_fn(pred, target).backward()
        for p in leaves:                  # what optimizer.zero_grad(set_to_none=True) does (train_model.py:566); nn.Module.zero_grad
            p.grad = None                 # walks the module tree instead: 0.8 ms of Python per step for this model

    out = {}
    with torch.enable_grad():
        for pd, md in dicts:                                      # untimed first pass (allocator pool of a second epoch, see epoch_leg)
            eager(pd, md)
        torch.cuda.synchronize()

        def run_eager():
            for pd, md in dicts:
                eager(pd, md)

        t_a, t_i, jp = _timed_passes(run_eager, steps)
        out["eager"] = {"what": "model(pdata, mdata) -> mse_loss -> loss.backward() -> gradient reset as optimizer.zero_grad() does it, every step a different batch; "
                                "median of the timed passes", "passes_ms": jp,
                        "ms_per_step": round(t_a / steps * 1e3, 4), "host_issue_ms_per_step": round(t_i / steps * 1e3, 4),
                        "pairs_per_s": round(B * steps / t_a, 1)}
        runner = GraphedTrainStep(model, loss_fn)
        for pd, md in dicts:                                      # untimed pass: captures every bucket the epoch visits
            runner.run(pd, md, target)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for pd, md in dicts:
            runner.run(pd, md, target)
        t_i = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_a = time.perf_counter() - t0
        out["bucketed_graphs"] = {"what": "GraphedTrainStep: one staging launch + one replay of the captured WHOLE-MODEL step of "
                                          "the batch's shape bucket; p.grad of all parameters set per step",
                                  "ms_per_step": round(t_a / steps * 1e3, 4), "host_issue_ms_per_step": round(t_i / steps * 1e3, 4),
                                  "pairs_per_s": round(B * steps / t_a, 1), "buckets": len(runner.buckets)}
    model.zero_grad(set_to_none=True)
    return out


def _epoch_summary(args, variant, steps, B, n_prot, n_drug, t_all, t_issue, batches, edges, t_gen, graphed):
    return {"what": "one epoch of DIFFERENT batches (untimed first pass, then the median of %d timed passes), eager (no HIP graph), nn.Module API (forward, backward(), gradient reset), C++ autograd fast path" % EPOCH_PASSES if
            __import__("gvp_hip._lib", fromlist=["bridge"]).bridge() is not None else
            "one epoch of DIFFERENT batches (untimed first pass, then the median of %d timed passes), eager (no HIP graph), nn.Module API, Python custom ops" % EPOCH_PASSES,
            "variant": variant, "steps": steps, "pairs_per_step": B, "unique_proteins": n_prot, "unique_drugs": n_drug,
            "ms_per_step": round(t_all / steps * 1e3, 4), "host_issue_ms_per_step": round(t_issue / steps * 1e3, 4),
            "pairs_per_s": round(B * steps / t_all, 1),
            "residues_per_step": [min(b[2] for b in batches), max(b[2] for b in batches)],
            "protein_edges_per_step": [min(edges), max(edges)], "setup_s": round(t_gen, 1), "bucketed_graphs": graphed}


def _cgroup_cpus():
    """CPU share of this process's cgroup (the GPU box hands a 1-GPU job a slice of the host's cores)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except Exception:
        pass
    return 1 << 20


def cpu_baseline(args, wl, state, pb, mb, train):
    """oracle/gvp_oracle.py (kind "port": torch CPU eager fp32, the reference's op order) on the host cores of the GPU
    box, running the SAME pass as the headline (forward + backward through oracle autograd when training) on a
    bounded sample of the same batch; thread counts {1, physical cores}; median of >= 20 runs after 3 warm-ups."""
    import davis_synth as ds
    from oracle import gvp_oracle as O
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    phys = max(1, min(phys, len(os.sched_getaffinity(0)), _cgroup_cpus()))
    # bounded sample: whole graphs from the front of the batch, sized so 2 x (3 + runs) passes stay within ~30 s
    budget_edges = 60000 if train else 180000
    n = args.cpu_pairs or wl["pairs"]
    if not args.cpu_pairs:
        while n > 1 and int(pb.eptr[n]) > budget_edges:
            n //= 2

    def cut(gb):
        ids = range(n)
        return ds.collate([dict(x_s=gb.x_s[gb.ptr[i]:gb.ptr[i + 1]], x_v=None if gb.x_v is None else gb.x_v[gb.ptr[i]:gb.ptr[i + 1]],
                                edge_index=gb.edge_index[:, gb.eptr[i]:gb.eptr[i + 1]] - gb.ptr[i],
                                e_s=gb.e_s[gb.eptr[i]:gb.eptr[i + 1]],
                                e_v=None if gb.e_v is None else gb.e_v[gb.eptr[i]:gb.eptr[i + 1]],
                                ntypes=gb.ntypes[gb.ptr[i]:gb.ptr[i + 1]], etypes=gb.etypes[gb.eptr[i]:gb.eptr[i + 1]])
                           for i in ids])
    sp, sm = (pb, mb) if n == wl["pairs"] else (cut(pb), cut(mb))
    pd, md = ds.to_torch(sp), ds.to_torch(sm)
    pp = {k[len("protein_gnn.gnn_model."):]: v.clone().requires_grad_(train and v.numel() > 0)
          for k, v in state.items() if k.startswith("protein_gnn.gnn_model.")}
    mp = {k[len("molecule_gnn.gnn_model."):]: v.clone().requires_grad_(train)
          for k, v in state.items() if k.startswith("molecule_gnn.gnn_model.")}
    gr = torch.randn(sp.num_nodes, 64)
    ga = torch.randn(sm.num_nodes, 64)
    leaves = [v for v in list(pp.values()) + list(mp.values()) if v.requires_grad]

    def cpu_step():
        nc = wl.get("convs", 2)
        res = O.protein_lba_forward(pp, pd["x"], pd["edge_index"], pd["ntypes"], pd["etypes"], pd["eattr"], num_convs=nc)
        atm = O.molecule_gine_forward(mp, md["x"], md["edge_index"], md["ntypes"], md["etypes"], md["eattr"], num_convs=nc)
        if train:
            torch.autograd.grad([res, atm], leaves, [gr, ga])

    results = {}
    old = torch.get_num_threads()
    with torch.set_grad_enabled(train):
        for threads in sorted({1, phys}, reverse=True):
            torch.set_num_threads(threads)
            for _ in range(3):
                cpu_step()
            ts = []
            for _ in range(max(args.cpu_runs, 3)):
                t0 = time.perf_counter()
                cpu_step()
                ts.append(time.perf_counter() - t0)
            results[threads] = statistics.median(ts)
    torch.set_num_threads(old)
    best = min(results, key=results.get)         # `value` is the faster of the two settings, `cores` its thread count
    return dict(value=round(n / results[best], 1), unit="graph-pairs/sec", cores=best, kind="port",
                by_threads={str(k): round(n / v, 1) for k, v in sorted(results.items())},
                sample=f"median of {max(args.cpu_runs, 3)} runs after 3 warm-ups of the same pass as the headline "
                       f"({'forward + backward (oracle autograd, all weight gradients)' if train else 'forward'}) over "
                       f"the first {n} of the {wl['pairs']} pairs of the batch ({sp.num_nodes} residues, {sp.num_edges} "
                       f"protein edges) through oracle/gvp_oracle.py, torch CPU eager fp32; timed at 1 thread "
                       f"and at {phys} threads (the physical cores this process may use), `value` = the faster")


if __name__ == "__main__":
    main()
