#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MMS hot path on MI355X.

Metric (BASELINE.json): QA pairs/sec (fwd+bwd) at batch 4096, 300-d; % HBM roofline.
Workload (cfg 2): SimCross dist_mode 1 (Euclidean), q,a (4096,1,300) fp32 ->
T (4096,1,1,1), forward + backward with a given top_diff, per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one forward+backward pass over one 4096-pair batch through the C
ABI (mms_simcross_forward_backward_f32: one launch).  Inputs are resident in
HBM before the timed region.  To keep the numbers HBM-bound rather than
Infinity-Cache-bound, the steps walk a ring of RING distinct batches
(RING x 19.7 MB > 1 GiB >> 256 MiB of L3); `--warm` re-uses one batch instead.
Steps are captured GROUP at a time into hipGraphs (launch-bound inner loop) and
replayed; N > 1 shards pairs over ranks (weak scaling: 4096 pairs per GPU) and
all-gathers the per-pair scores of each GROUP of steps with one RCCL call on a
side stream, overlapped with the next group's compute.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`
and `cpu_baseline` objects added.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PAIRS, DIM = 4096, 300
S = 4  # sizeof(float)
# algorithmic bytes per fwd+bwd pass of SimCross modes 0/1
B_FWD = S * (N_PAIRS * 2 * DIM + N_PAIRS)              # 9,846,784
B_BWD = S * (2 * N_PAIRS * 2 * DIM + 2 * N_PAIRS)      # 19,693,568
B_UNFUSED = B_FWD + B_BWD                              # 29,540,352  (7,212 B/pair)
B_FUSED = S * (2 * N_PAIRS * 2 * DIM + 2 * N_PAIRS)    # q,a read once; dq,da written once; dT in, T out
HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4096)
    p.add_argument("--warmup", type=int, default=256)
    p.add_argument("--group", type=int, default=64, help="steps per hipGraph / all-gather bucket")
    p.add_argument("--ring", type=int, default=64, help="distinct batches walked (HBM-cold)")
    p.add_argument("--warm", action="store_true", help="re-use one batch (Infinity-Cache-warm)")
    p.add_argument("--path", choices=["fused", "layers", "triplet"], default="fused",
                   help="fused: one launch fwd+bwd (default); layers: Forward then Backward "
                        "launches (the Layer API sequence); triplet: fused (q,a+,a-) step")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-variants", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=10.0)
    p.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                   help="nccl = RCCL over xGMI (default). gloo = rehearsal of the N>1 control flow "
                        "on a box with fewer GPUs than ranks (scores staged through the host)")
    return p.parse_args()


class Batches:
    """RING batches of synthetic GloVe-like pairs, generated on device (seed 1701)."""

    def __init__(self, torch, ring, path, rank):
        g = torch.Generator(device="cuda").manual_seed(1701 + rank)
        mk = lambda *s: torch.randn(*s, device="cuda", generator=g)
        self.q = mk(ring, N_PAIRS, 1, DIM) * 0.4          # N(0, 0.4^2): SURVEY 8(d)
        self.a = mk(ring, N_PAIRS, 1, DIM) * 0.4
        self.dT = mk(ring, N_PAIRS, 1, 1, 1)
        self.dq = torch.empty_like(self.q)
        self.da = torch.empty_like(self.a)
        self.top_own = torch.empty(ring, N_PAIRS, 1, 1, 1, device="cuda")
        if path == "cosine":
            self.n0 = torch.empty(ring, N_PAIRS, 1, device="cuda")
            self.n1 = torch.empty(ring, N_PAIRS, 1, device="cuda")
        if path == "triplet":
            self.an = mk(ring, N_PAIRS, 1, DIM) * 0.4
            self.y = (torch.rand(ring, N_PAIRS, 1, device="cuda", generator=g) < 0.8).float()
            self.dan = torch.empty_like(self.a)
            self.sneg = torch.empty(ring, N_PAIRS, 1, device="cuda")
            self.loss = torch.empty(ring, 1, device="cuda")


def make_step(capi, bt, path):
    """step(slot, top): one pass over batch `slot`, scores written to `top`."""
    if path == "fused":
        def step(i, top):
            capi.simcross_forward_backward(1, bt.q[i], bt.a[i], bt.dT[i], top, bt.dq[i], bt.da[i])
    elif path == "cosine":
        def step(i, top):
            capi.simcross_forward_backward(0, bt.q[i], bt.a[i], bt.dT[i], top, bt.dq[i], bt.da[i],
                                           norm0=bt.n0[i], norm1=bt.n1[i])
    elif path == "layers":
        def step(i, top):
            capi.simcross_forward(1, bt.q[i], bt.a[i], top)
            capi.simcross_backward(1, bt.q[i], bt.a[i], top, bt.dT[i], bt.dq[i], bt.da[i])
    else:
        def step(i, top):
            capi.triplet_euclid_step(bt.q[i], bt.a[i], bt.an[i], bt.y[i], top.view(N_PAIRS, 1),
                                     bt.sneg[i], bt.loss[i], bt.dq[i], bt.da[i], bt.dan[i],
                                     margin=0.05)
    return step


def cpu_baseline(seconds):
    """The oracle (CPU restatement of the reference loops), one thread, on a bounded
    sample of the same workload: whole 4096x300 fwd+bwd passes for ~`seconds`."""
    import numpy as np
    from oracle import cpu_oracle as O
    r = np.random.default_rng(1701)
    q = (r.standard_normal((N_PAIRS, 1, DIM)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N_PAIRS, 1, DIM)) * 0.4).astype(np.float32)
    dT = r.standard_normal((N_PAIRS, 1, 1, 1)).astype(np.float32)
    t1 = O.time_simcross_fwd_bwd(1, q, a, dT, iters=2) / 2          # warm-up + estimate
    iters = max(3, min(2000, int(seconds / max(t1, 1e-6))))
    t = O.time_simcross_fwd_bwd(1, q, a, dT, iters=iters)
    out = {"value": N_PAIRS * iters / t, "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": "%d fwd+bwd passes of SimCross Euclid (4096,1,300) fp32, oracle/mms_oracle.c "
                     "-O2 single thread, %.1f s on %d-cpu host" % (iters, t, os.cpu_count() or 0)}
    # courtesy upper bound: the same loops with the pairs dealt to the cores this
    # process may use; the reference layer itself is single-threaded
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))           # a one-GPU box's share of the host
    if ncores > 1:
        it_mt = max(8, min(20000, int(iters * ncores * 0.5)))
        O.time_simcross_fwd_bwd_mt(1, q, a, dT, iters=4, threads=ncores)
        tm = O.time_simcross_fwd_bwd_mt(1, q, a, dT, iters=it_mt, threads=ncores)
        out["all_cores"] = {"value": N_PAIRS * it_mt / tm, "unit": "pairs/s", "cores": ncores,
                            "sample": "%d passes, OpenMP over pairs, %.1f s" % (it_mt, tm)}
    return out


def load_traffic(path_name):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/),
    collected and corrected by tools/pmc_traffic.py; None when absent."""
    f = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        d = json.load(open(f))
        return d.get(path_name, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def run(args):
    import torch
    import torch.distributed as dist
    from mms_answer_selection_amd import build, capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch --gpus %d with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    local = local % torch.cuda.device_count() if args.backend == "gloo" else local
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    if rank == 0:
        build.build_all()
    if world > 1:
        dist.barrier()
    capi.lib()

    G = max(1, args.group)
    ring = 1 if args.warm else max(G, args.ring // G * G)
    bt = Batches(torch, ring, args.path, rank)
    step = make_step(capi, bt, args.path)
    ngroups = ring // G if not args.warm else 1

    # score buckets: group g writes its G x 4096 scores into bucket g % 2
    buckets = [torch.empty(G, N_PAIRS, 1, 1, 1, device="cuda") for _ in range(2)]
    gathered = [torch.empty(world * G, N_PAIRS, 1, 1, 1, device="cuda") for _ in range(2)] if world > 1 else None
    comm = torch.cuda.Stream() if world > 1 else None
    bucket_free = [torch.cuda.Event() for _ in range(2)]   # gather of that bucket finished
    main = torch.cuda.current_stream()

    def group_body(gi):
        b = buckets[gi % 2]
        for s in range(G):
            slot = 0 if args.warm else (gi % ngroups) * G + s
            step(slot, b[s])

    # eager warm-up of every code path (also fills caches / instantiates kernels)
    for gi in range(max(2, min(ngroups, 4))):
        group_body(gi)
    torch.cuda.synchronize()

    graphs = {}
    use_graph = not args.no_graph
    if use_graph:
        # group gi uses ring slots (gi % ngroups) and bucket gi % 2 -> lcm(ngroups, 2) distinct graphs
        period = ngroups if ngroups % 2 == 0 else ngroups * 2
        cap = torch.cuda.Stream()
        cap.wait_stream(main)
        with torch.cuda.stream(cap):
            for gi in range(period):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=cap):
                    group_body(gi)
                graphs[gi] = gph
        main.wait_stream(cap)
        torch.cuda.synchronize()
    else:
        period = 1

    def gather_bucket(bi):
        """All-gather of the per-pair scores of one bucket (GROUP steps x 4096 pairs per rank),
        on the side stream, so it overlaps the next group's compute."""
        comm.wait_stream(main)
        with torch.cuda.stream(comm):
            if args.backend == "nccl":
                dist.all_gather_into_tensor(gathered[bi].view(-1), buckets[bi].view(-1))
            else:  # rehearsal: host-staged
                h = buckets[bi].view(-1).cpu()
                ho = torch.empty(world * h.numel())
                dist.all_gather_into_tensor(ho, h)
                gathered[bi].view(-1).copy_(ho)
            bucket_free[bi].record(comm)

    def run_group(gi):
        if world > 1:
            main.wait_event(bucket_free[gi % 2])      # previous gather of this bucket done
        if use_graph:
            graphs[gi % period].replay()
        else:
            group_body(gi)
        if world > 1:
            gather_bucket(gi % 2)

    tails = {}

    def tail_graph(gi, rem):
        """A hipGraph of the first `rem` steps of group gi (a --steps / --warmup that is not a
        multiple of GROUP must not fall back to host-paced launches inside the timed region)."""
        key = (gi % period, rem)
        if use_graph and key not in tails:
            b = buckets[gi % 2]
            cap2 = torch.cuda.Stream()
            cap2.wait_stream(main)
            with torch.cuda.stream(cap2):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=cap2):
                    for s in range(rem):
                        step(0 if args.warm else (gi % ngroups) * G + s, b[s])
            main.wait_stream(cap2)
            tails[key] = gph
        return tails.get(key)

    def run_steps(k, g0):
        full, rem = divmod(k, G)
        for i in range(full):
            run_group(g0 + i)
        if rem:                                        # time EXACTLY k steps
            gi = g0 + full
            b = buckets[gi % 2]
            if world > 1:
                main.wait_event(bucket_free[gi % 2])
            gph = tail_graph(gi, rem)
            if gph is not None:
                gph.replay()
            else:
                for s in range(rem):
                    step(0 if args.warm else (gi % ngroups) * G + s, b[s])
            if world > 1:
                gather_bucket(gi % 2)
        return g0 + full + (1 if rem else 0)

    # capture the partial groups this run will need BEFORE anything is timed
    _fw, _rw = divmod(args.warmup, G)
    if _rw:
        tail_graph(_fw, _rw)
    _g_after_warm = _fw + (1 if _rw else 0)
    _fs, _rs = divmod(args.steps, G)
    if _rs:
        tail_graph(_g_after_warm + _fs, _rs)
    torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1:
        # communicator set-up (lazy on the first collective) must not land in the timed region even
        # when --warmup is 0: one untimed all-gather per bucket
        gather_bucket(0)
        gather_bucket(1)
        fence()
    g0 = run_steps(args.warmup, 0)
    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(main)
    run_steps(args.steps, g0)
    e1.record(main)
    fence()
    t1 = time.perf_counter()
    wall = t1 - t0
    ev_ms = e0.elapsed_time(e1)

    tmax = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())

    out = None
    if rank == 0:
        launches_per_step = {"fused": 1, "layers": 2, "triplet": 2}[args.path]
        step_us_ev = ev_ms * 1e3 / args.steps            # HIP events on the launch stream
        achieved = B_UNFUSED / (step_us_ev * 1e-6) / 1e9 if args.path != "triplet" else None
        value = world * N_PAIRS * args.steps / wall
        kernel = {"fused": "mms::euclid_pair32_kernel<75,true,true,...> (SimCross Euclid fwd+bwd, one launch)",
                  "layers": "mms::euclid_pair32_kernel<75,true,false,...> + <75,false,true,...>",
                  "triplet": "mms::triplet_wave_kernel<3> + loss_finish_kernel"}[args.path]
        out = {
            "metric": "QA pairs/sec (fwd+bwd) at batch 4096, 300-d; % HBM roofline",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: SimCross dist_mode=1 (Euclid) fwd+bwd, q,a (4096,1,300) fp32 per GPU",
                       "pairs_per_gpu": N_PAIRS, "dim": DIM, "global_batch": world * N_PAIRS,
                       "path": args.path, "launches_per_step": launches_per_step,
                       "euclid_backward_arithmetic": capi.get_euclid_backward_mode() +
                       (" (scores bit-identical to the CPU code; gradient elements <= 2 ulp from it, bar 1e-5)"
                        if capi.get_euclid_backward_mode() == "fp32" else " (gradients bit-identical too)"),
                       "residency": "cache-warm (1 batch)" if args.warm else
                                    "HBM-cold ring of %d batches (%.2f GiB)" % (ring, ring * 19.7e6 / 2**30),
                       "hip_graph_group": G if use_graph else 0,
                       "parallelism": "pair-sharded x%d%s" % (
                           world, ", %s all-gather of scores per %d steps" % (
                               "RCCL" if args.backend == "nccl" else "host-staged gloo (rehearsal)", G)
                           if world > 1 else "")},
        }
        if achieved is not None:
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(args.path),
                "kernel": kernel,
                "algorithmic_bytes_per_launch": B_UNFUSED if args.path == "fused" else None,
                "algorithmic_bytes_per_step": B_UNFUSED,
                "fused_compulsory_bytes_per_step": B_FUSED if args.path == "fused" else None,
                "frac_vs_fused_bytes": (B_FUSED / (step_us_ev * 1e-6) / 1e9 / HBM_PEAK_GBS
                                        if args.path == "fused" else None),
                "avg_step_us_hip_events": step_us_ev,
                "note": "duration = HIP events over the timed region / steps on rank 0 "
                        "(launch-to-launch, includes inter-kernel gaps)"}
    if world > 1:
        dist.barrier()

    # side measurements on rank 0 at N=1 only (not part of the timed region above)
    if rank == 0 and world == 1 and not args.no_variants:
        out["variants"] = variants(torch, capi, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


DEFAULT_BWD_MODE = os.environ.get("MMS_EUCLID_BWD", "fp32")
if DEFAULT_BWD_MODE not in ("fp32", "reference"):
    DEFAULT_BWD_MODE = "reference" if DEFAULT_BWD_MODE in ("exact", "1") else "fp32"


def variants(torch, capi, args):
    """Short interleaved measurements of the other entry points / residency, same device."""
    res = {}
    K, G = 1024, 16
    for name, path, ring in (("fused_cold", "fused", 64), ("fused_warm", "fused", 1),
                             ("fused_cold_reference_rounding_bwd", "fused", 64),
                             ("layers_cold", "layers", 64), ("layers_warm", "layers", 1),
                             ("cosine_fused_cold", "cosine", 64),
                             ("triplet_cold", "triplet", 48)):
        ref_mode = name.endswith("reference_rounding_bwd")
        if path == args.path and ((ring == 1) == args.warm) and not ref_mode:
            continue
        # the launcher picks the kernel variant at capture time (include/mms.h)
        capi.set_euclid_backward_mode("reference" if ref_mode else DEFAULT_BWD_MODE)
        bt = Batches(torch, ring, path, 0)
        step = make_step(capi, bt, path)
        top = torch.empty(G, N_PAIRS, 1, 1, 1, device="cuda")
        ng = max(1, ring // G)
        graphs = []
        for gi in range(ng):
            for s in range(G):
                step((gi * G + s) % ring, top[s])
        torch.cuda.synchronize()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            for gi in range(ng):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=cap):
                    for s in range(G):
                        step((gi * G + s) % ring, top[s])
                graphs.append(gph)
        torch.cuda.current_stream().wait_stream(cap)
        for i in range(8):
            graphs[i % ng].replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K // G):
            graphs[i % ng].replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / K
        pairs = N_PAIRS
        res[name] = {"us_per_step": us, "pairs_per_s": pairs / (us * 1e-6)}
        if path != "triplet":
            res[name]["frac_hbm_unfused_bytes"] = B_UNFUSED / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
        del bt, graphs
        torch.cuda.empty_cache()
    capi.set_euclid_backward_mode(DEFAULT_BWD_MODE)
    res.update(other_configs(torch, capi))
    return res


def _graph_time(torch, fn, iters=16, reps=6):
    """Median us per call of `fn`, replayed from a hipGraph of `iters` calls (cache-warm)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=cap):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(cap)
    gph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(ts)[len(ts) // 2]


def other_configs(torch, capi):
    """The other single-GPU BASELINE configs, measured next to the headline (cache-warm,
    hipGraph-replayed): cfg 3 (SimMatrix on fp32 MFMA) and cfg 5's per-GPU shard (fp16 storage)."""
    out = {}
    g = torch.Generator(device="cuda").manual_seed(1701)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.4
    # cfg 3: SimMatrix q,a (16384,300), W (300,300) fp32
    N, K = 16384, 300
    q, a = rnd(N, K), rnd(N, K)
    W = torch.rand(K, K, device="cuda", generator=g) * 0.16 - 0.08
    dT = torch.randn(N, 1, device="cuda", generator=g)
    top, scr = torch.empty(N, 1, device="cuda"), torch.empty(N, K, device="cuda")
    dq, da, dW = torch.empty_like(q), torch.empty_like(a), torch.zeros_like(W)
    ws = capi.Workspace()

    # the Layer's call sequence: Backward reuses the forward's Q.W for da (3 GEMMs executed per step)
    def cfg3():
        capi.simmatrix_forward(q, a, W, top, scr)
        capi.simmatrix_backward(q, a, W, dT, dq, da, dW, ws=ws, qw=scr)
    us = _graph_time(torch, cfg3, iters=4)
    flops = 2.0 * N * K * K + 2.0 * N * K + 6.0 * N * K * K        # the reference's 4 products
    done = 2.0 * N * K * K + 2.0 * N * K + 4.0 * N * K * K         # executed: Q.W once
    out["cfg3_simmatrix_16384x300x300_fwd_bwd"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6), "TFLOPs": flops / us / 1e6,
        "frac_mfma_fp32_peak": flops / (us * 1e-6) / 157.3e12,
        "executed_TFLOPs": done / us / 1e6, "executed_frac_mfma_fp32_peak": done / (us * 1e-6) / 157.3e12,
        "bound": "mfma", "dtype": "f32"}

    def cfg3_nocache():
        capi.simmatrix_forward(q, a, W, top, scr)
        capi.simmatrix_backward(q, a, W, dT, dq, da, dW, ws=ws)
    us = _graph_time(torch, cfg3_nocache, iters=4)
    out["cfg3_simmatrix_recomputing_backward"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6), "TFLOPs": flops / us / 1e6,
        "frac_mfma_fp32_peak": flops / (us * 1e-6) / 157.3e12, "bound": "mfma", "dtype": "f32"}
    del q, a, W, dT, top, scr, dq, da, dW
    # cfg 1: the driver's training geometry, SimCross bilinear M = 4 with bias (do_trec_qa_clean.py:452-496)
    for (N1, Wd, D1, M1) in ((32, 40, 300, 4), (50, 40, 50, 4)):
        q1, a1 = rnd(N1, Wd, D1), rnd(N1, Wd, D1)
        W1 = torch.rand(M1, D1, D1, device="cuda", generator=g) * 0.16 - 0.08
        b1 = torch.zeros(M1, Wd, Wd, device="cuda")
        t1 = torch.empty(N1, M1, Wd, Wd, device="cuda")
        dT1 = torch.randn(N1, M1, Wd, Wd, device="cuda", generator=g)
        dq1, da1, dW1, db1 = torch.empty_like(q1), torch.empty_like(a1), torch.empty_like(W1), torch.zeros_like(b1)

        def cfg1():
            capi.simcross_forward(2, q1, a1, t1, W=W1, bias=b1, ws=ws)
            capi.simcross_backward(2, q1, a1, t1, dT1, dq1, da1, W=W1, bias_term=True, dW=dW1, dbias=db1, ws=ws)
        us = _graph_time(torch, cfg1, iters=4)
        fl = 2.0 * N1 * M1 * Wd * D1 * (D1 + Wd) + 8.0 * N1 * M1 * Wd * D1 * (D1 + Wd)
        out["cfg1_bilinear_%dx%dx%dx%d_M%d_fwd_bwd" % (N1, Wd, Wd, D1, M1)] = {
            "us_per_step": us, "pairs_per_s": N1 / (us * 1e-6), "TFLOPs": fl / us / 1e6,
            "frac_mfma_fp32_peak": fl / (us * 1e-6) / 157.3e12, "bound": "mfma / launch", "dtype": "f32"}
        del q1, a1, W1, b1, t1, dT1, dq1, da1, dW1, db1
    # PairRankLoss alone (forward s*5*count bytes, backward s*5*count): the batch of cfg 2 and a
    # size at which bandwidth rather than the launch floor is visible
    for cnt in (N_PAIRS, 1 << 22):
        pa, pb = torch.rand(cnt, 1, device="cuda", generator=g), torch.rand(cnt, 1, device="cuda", generator=g)
        py = (torch.rand(cnt, 1, device="cuda", generator=g) < 0.2).float()
        po, ps = torch.empty_like(pa), torch.empty_like(pa)
        pl = torch.empty(1, device="cuda")
        pda, pdb = torch.empty_like(pa), torch.empty_like(pa)

        def prl():
            capi.pairrank_forward(pa, pb, py, po, ps, pl, margin=0.1, ws=ws)
            capi.pairrank_backward(py, po, ps, pda, pdb)
        us = _graph_time(torch, prl, iters=8)
        out["pairrankloss_%dx1_fwd_bwd" % cnt] = {
            "us_per_step": us, "pairs_per_s": cnt / (us * 1e-6),
            "frac_hbm_unfused_bytes": 4.0 * 10 * cnt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "bound": "hbm / launch"}
        del pa, pb, py, po, ps, pl, pda, pdb
    # cfg 5 shard: 8192 pairs x 1024-d, fp16 storage, fused fwd+bwd (one launch)
    N, D = 8192, 1024
    qh, ah = rnd(N, 1, D).half(), rnd(N, 1, D).half()
    dT = torch.randn(N, 1, 1, 1, device="cuda", generator=g)
    top = torch.empty(N, 1, 1, 1, device="cuda")
    dqh, dah = torch.empty_like(qh), torch.empty_like(ah)
    us = _graph_time(torch, lambda: capi.simcross_euclid_forward_backward_f16(qh, ah, dT, top, dqh, dah))
    b_unfused = 2 * (3 * N * 2 * D) + 4 * 3 * N                      # SURVEY 8(d), s = 2: 100.7 MB
    out["cfg5_shard_8192x1024_fp16_storage_fused"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6),
        "frac_hbm_unfused_bytes": b_unfused / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "bound": "hbm",
        "dtype": "f16 storage / f32 arithmetic"}
    del qh, ah, dT, top, dqh, dah
    # cfg 4: scoring only -- the TREC-QA test split (1517 candidates, 40 x 40 word grids, Dw = 50),
    # whole and as the 190-candidate shard one of 8 GPUs scores, plus MAP + MRR on 1517 sentence scores
    for name, n in (("cfg4_scoring_1517x40x40x50_forward", 1517), ("cfg4_shard_190x40x40x50_forward", 190)):
        qg, ag = rnd(n, 40, 50), rnd(n, 40, 50)
        tg = torch.empty(n, 1, 40, 40, device="cuda")
        us = _graph_time(torch, lambda: capi.simcross_forward(1, qg, ag, tg))
        b = 4.0 * (2 * n * 40 * 50 + n * 1600)                       # SURVEY 8(d) B_fwd
        out[name] = {"us_per_step": us, "pairs_per_s": n / (us * 1e-6), "GBps_algorithmic": b / us / 1e3,
                     "bound": "fp32 VALU (3 flop per (j,k,d), d-ordered sums)"}
        del qg, ag, tg
    # cfg 4 in the mode the reference's network_v4 scores with: SimCross bilinear, M = 4, bias (forward only)
    n = 1517
    qg, ag = rnd(n, 40, 50), rnd(n, 40, 50)
    Wb = torch.rand(4, 50, 50, device="cuda", generator=g) * 0.16 - 0.08
    bb = torch.zeros(4, 40, 40, device="cuda")
    tb = torch.empty(n, 4, 40, 40, device="cuda")
    us = _graph_time(torch, lambda: capi.simcross_forward(2, qg, ag, tb, W=Wb, bias=bb, ws=ws))
    fl = 2.0 * n * 4 * 40 * 50 * (50 + 40)
    out["cfg4_scoring_bilinear_M4_1517x40x40x50_forward"] = {
        "us_per_step": us, "pairs_per_s": n / (us * 1e-6), "TFLOPs": fl / us / 1e6,
        "note": "one fused launch per step (Q_n W_m kept in LDS); 89 us as two batched GEMMs"}
    del qg, ag, Wb, bb, tb
    # the same scores from WORD IDS: Embed (50-d table, 20,000 words) then SimCross, and the fused call
    n, K = 1517, 20000
    tab = rnd(K, 50)
    iq = torch.randint(0, K, (n, 40), device="cuda", generator=g).float()
    ia = torch.randint(0, K, (n, 40), device="cuda", generator=g).float()
    qe, ae = torch.empty(n, 40, 50, device="cuda"), torch.empty(n, 40, 50, device="cuda")
    tg = torch.empty(n, 1, 40, 40, device="cuda")

    def embed_then_score():
        capi.embed_forward(iq, tab, qe.view(-1, 50))
        capi.embed_forward(ia, tab, ae.view(-1, 50))
        capi.simcross_forward(1, qe, ae, tg)
    us2 = _graph_time(torch, embed_then_score)
    us1 = _graph_time(torch, lambda: capi.embed_simcross_forward(1, iq, ia, tab, tg))
    out["cfg4_scoring_from_word_ids_1517x40x40x50"] = {
        "us_per_step_embed_then_simcross": us2, "us_per_step_fused": us1, "pairs_per_s": n / (us1 * 1e-6),
        "note": "Embed gather fused into SimCross's loads (mms_embed_simcross_forward_f32) vs three launches"}
    del tab, iq, ia, qe, ae, tg
    n = 1517
    sc = torch.rand(n, device="cuda", generator=g)
    prob = torch.stack([1 - sc, sc], 1).contiguous()
    lab = (torch.rand(n, device="cuda", generator=g) < 0.2).float()
    grp = torch.sort(torch.randint(0, 68, (n,), device="cuda", generator=g).float()).values
    capi.rank_map_mrr(prob, lab, grp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        capi.rank_map_mrr(prob, lab, grp)
    out["cfg4_map_mrr_1517_candidates"] = {"us_per_call": (time.perf_counter() - t0) / 20 * 1e6,
                                            "note": "sort + walks + folds + device-to-host copy of the three scalars"}
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    run(parse())
