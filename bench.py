#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MMS hot path on MI355X.

Metric (BASELINE.json): QA pairs/sec (fwd+bwd) at batch 4096, 300-d; % HBM roofline.
Workload (cfg 2): SimCross dist_mode 1 (Euclidean), q,a (4096,1,300) fp32 ->
T (4096,1,1,1), forward + backward with a given top_diff, per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: starts its own N rank processes, launch_ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (the driver's form; same ranks)

A "step" is one forward + one backward pass over one 4096-pair batch through the C ABI.  The
default `--path layers` issues what a Caffe host can issue through the Layer API: one Forward
launch (mms_simcross_forward_f32), then one Backward launch (mms_simcross_backward_f32) --
Net::ForwardFromTo / BackwardFromTo (net.cpp:535-546, 581-591) and `caffe time`
(tools/caffe.cpp:349-361) run all forwards, then all backwards, so no reference caller can hand
top_diff to the forward.  `--path fused` (one launch for both, mms_simcross_forward_backward_f32)
is a labelled variant for hosts that do know top_diff up front.

Protocol (tools/caffe.cpp:318-385): inputs resident in HBM; W untimed warm-up steps;
then the K-step region is timed REPEATS (5) times, each repeat bracketed by barrier +
torch.cuda.synchronize() on both sides and by two HIP events on the launch stream; ONE clock -- the
events, max over ranks -- gives `value`, `ms_per_step` and `roofline`; the MEDIAN repeat is
reported (all repeats are in `config.repeats_ms_per_step`).  Steps walk a ring of RING distinct
batches (RING x 19.7 MB > 1 GiB >> 256 MiB of Infinity Cache) continuing where the warm-up stopped,
and the caches are flushed (a 512 MiB fill) before the warm-up, so every timed step reads its q and
a from HBM ("cold"); `--warm` re-uses one batch instead.  Between each repeat's opening fence and its
start event, `--lead-in` (64) UNTIMED steps of the same walk are enqueued on ring slots other than
the region's: the device is busy while the host enqueues the start event and the region's first
hipGraph, so a short region (the driver's --steps 20) does not count host submission latency as
kernel time; what remains of a short region's fixed cost (two event packets and a graph boundary,
about 9 us) is visible next to `roofline.long_region`, the same walk over 2048 steps.  Steps are ALWAYS captured into
hipGraphs of up to GROUP steps (round 3: a region launched kernel by kernel from the host is only as good as the host is
idle; `--launch eager` keeps that mode for comparison).  N > 1 shards pairs over ranks (weak scaling: 4096 pairs
per GPU) and all-gathers the per-pair scores of each GROUP of steps with one RCCL call on a side
stream, overlapped with the next group's compute, and adds `config.strong_split_variant` (ONE 4096-pair batch split over
the ranks, gather per step); `--workload cfg5` (`--f16-distance ordered|tree`) / `cfg4` are the strong-scaling legs of
the two BASELINE configurations that are multi-GPU by definition.  `python bench.py --gpus N` typed as is starts its own
N rank processes (launch_ranks); the torchrun form works too.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
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
REPEATS = 5


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=4096)
    p.add_argument("--warmup", type=int, default=256)
    p.add_argument("--group", type=int, default=256, help="steps per hipGraph / all-gather bucket")
    p.add_argument("--ring", type=int, default=64, help="distinct batches walked (HBM-cold)")
    p.add_argument("--warm", action="store_true", help="re-use one batch (Infinity-Cache-warm)")
    p.add_argument("--path", choices=["layers", "fused", "triplet"], default="layers",
                   help="layers (default): Forward launch then Backward launch, the Layer API sequence; "
                        "fused: one launch fwd+bwd (needs top_diff before the forward: not reachable from a "
                        "Caffe Net); triplet: fused (q,a+,a-) step")
    p.add_argument("--workload", choices=["cfg2", "cfg4", "cfg5"], default="cfg2",
                   help="cfg2 (default, the metric's configuration); cfg5: 65,536 x 1024 fp16-storage pairs "
                        "split over the ranks (strong scaling); cfg4: 1,517 candidates sharded -> all-gather "
                        "of scores -> MAP/MRR on every rank (strong scaling)")
    p.add_argument("--repeats", type=int, default=REPEATS)
    p.add_argument("--lead-in", type=int, default=64,
                   help="untimed steps of the same walk enqueued between each repeat's opening fence and its start "
                        "event (device busy and at its running clocks when the timed region's first graph arrives)")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--launch", choices=["auto", "graph", "eager"], default="auto",
                   help="graph: steps captured into hipGraphs of --group steps; eager: every kernel launched by "
                        "itself through pre-bound C-ABI calls, the way a native Layer host does (a hipGraphLaunch "
                        "costs 6.5 us of device time, a third of a microsecond per step in the driver's 20-step "
                        "regions) -- host-paced, so only as good as the host is idle; auto = graph")
    p.add_argument("--f16-distance", choices=["ordered", "tree"], default="ordered",
                   help="--workload cfg5: 'ordered' = the reference's d-ascending fp32 sum of the 1024 squares, scores "
                        "bit-identical to the fp32 CPU code on the fp16-rounded inputs; 'tree' = fixed-shape tree sum, "
                        "within ~1e-6 relative of it -- SURVEY 8(d) asks 1e-3 of this configuration (the reference has "
                        "no fp16), so both are contract-compliant; the line reports both accountings either way")
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
        if path in ("triplet", "triplet_noloss"):
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
    elif path == "triplet_noloss":     # loss = NULL: scores and gradients only (the scalar is a display value)
        def step(i, top):
            capi.triplet_euclid_step(bt.q[i], bt.a[i], bt.an[i], bt.y[i], top.view(N_PAIRS, 1),
                                     bt.sneg[i], None, bt.dq[i], bt.da[i], bt.dan[i], margin=0.05)
    else:
        def step(i, top):
            capi.triplet_euclid_step(bt.q[i], bt.a[i], bt.an[i], bt.y[i], top.view(N_PAIRS, 1),
                                     bt.sneg[i], bt.loss[i], bt.dq[i], bt.da[i], bt.dan[i],
                                     margin=0.05)
    return step


def make_raw_step(capi, bt, path, torch):
    """step(slot, top_ptr) through the argument-block entry points (include/mms.h: mms_simcross_*_block_f32): one
    pre-filled block per ring slot, two ctypes arguments per call -- ~1.3 us of host time per launch instead of ~4
    for the 21-argument form, so the host stays ahead of a 3.8-us kernel without a hipGraph."""
    import ctypes as C
    lib = capi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    blocks = []
    for i in range(bt.q.shape[0]):
        blk = capi.SimCrossArgs()
        blk.dist_mode, blk.N, blk.W1, blk.W2, blk.D, blk.M = 1, N_PAIRS, 1, 1, DIM, 1
        blk.q, blk.a = bt.q[i].data_ptr(), bt.a[i].data_ptr()
        blk.top_diff, blk.dq, blk.da = bt.dT[i].data_ptr(), bt.dq[i].data_ptr(), bt.da[i].data_ptr()
        blk.propagate_down0 = blk.propagate_down1 = 1
        blocks.append((blk, C.byref(blk)))
    fwd, bwd = lib.mms_simcross_forward_block_f32, lib.mms_simcross_backward_block_f32
    if path != "layers":
        return None

    def step(i, top):
        blk, ref = blocks[i]
        blk.top = top
        if fwd(ref, st) | bwd(ref, st):
            raise RuntimeError("C ABI call failed")
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
    """HBM bytes per STEP from the committed rocprofv3 --pmc passes (profiles/traffic.json; separate
    FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes, by tools/pmc_traffic.py);
    None when absent."""
    f = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        d = json.load(open(f))
        e = d.get(path_name, {})
        return e.get("hbm_bytes_per_step", e.get("hbm_bytes_per_launch"))
    except Exception:
        return None


def init_dist(args):
    import torch
    import torch.distributed as dist
    from mms_answer_selection_amd import build, capi
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                   # under a launcher the environment is authoritative
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
    return torch, dist, capi, world, rank


def flush_caches(torch, mib=512):
    """Untimed: overwrite `mib` MiB so that nothing of the ring is left in L2 / the 256-MiB Infinity Cache."""
    scratch = torch.empty(mib << 18, device="cuda")
    scratch.fill_(1.0)
    torch.cuda.synchronize()
    del scratch


class Region:
    """Steps [first, first+k) of the ring walk, cut into hipGraphs of at most G steps.  Step i reads ring
    slot i % ring and writes its scores into row (i - chunk start) of a bucket; chunks alternate buckets."""

    def __init__(self, torch, step, ring, G, buckets, use_graph, raw_step=None):
        self.torch, self.step, self.ring, self.G = torch, step, ring, G
        self.buckets, self.use_graph = buckets, use_graph
        self.raw_step = raw_step                     # eager mode: pre-bound calls on raw pointers
        self.bucket_ptr = [b.data_ptr() for b in buckets]
        self.row_bytes = buckets[0][0].numel() * 4
        self.graphs = {}
        self.nchunk = 0

    def chunks(self, first, k):
        out, i = [], first
        while k > 0:
            c = min(self.G, k)
            out.append((i, c))
            i += c
            k -= c
        return out

    def body(self, i0, cnt, bi):
        if self.raw_step is not None and not self.use_graph:
            base, rb, ring, raw = self.bucket_ptr[bi], self.row_bytes, self.ring, self.raw_step
            for s in range(cnt):
                raw((i0 + s) % ring, base + s * rb)
            return
        b = self.buckets[bi]
        for s in range(cnt):
            self.step((i0 + s) % self.ring, b[s])

    def capture(self, plan):
        """Capture every (slot, count, bucket) graph `plan` (a list of chunk lists) will replay."""
        if not self.use_graph:
            return
        torch = self.torch
        n = 0
        main = torch.cuda.current_stream()
        cap = torch.cuda.Stream()
        cap.wait_stream(main)
        with torch.cuda.stream(cap):
            for chunks in plan:
                for (i0, cnt) in chunks:
                    key = (i0 % self.ring, cnt, n & 1)
                    n += 1
                    if key in self.graphs:
                        continue
                    gph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gph, stream=cap):
                        self.body(i0, cnt, key[2])
                    self.graphs[key] = gph
        main.wait_stream(cap)
        torch.cuda.synchronize()

    def capture_fixed(self, chunks, bi):
        """Graphs of `chunks` writing bucket `bi`, whatever the running parity (untimed lead-in replays)."""
        if not self.use_graph:
            return
        torch = self.torch
        main = torch.cuda.current_stream()
        cap = torch.cuda.Stream()
        cap.wait_stream(main)
        with torch.cuda.stream(cap):
            for (i0, cnt) in chunks:
                key = (i0 % self.ring, cnt, bi)
                if key not in self.graphs:
                    gph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gph, stream=cap):
                        self.body(i0, cnt, bi)
                    self.graphs[key] = gph
        main.wait_stream(cap)
        torch.cuda.synchronize()

    def run_fixed(self, chunks, bi):
        for (i0, cnt) in chunks:
            if self.use_graph:
                self.graphs[(i0 % self.ring, cnt, bi)].replay()
            else:
                self.body(i0, cnt, bi)

    def run(self, chunks, before=None, after=None):
        for (i0, cnt) in chunks:
            bi = self.nchunk & 1
            self.nchunk += 1
            if before:
                before(bi)
            if self.use_graph:
                self.graphs[(i0 % self.ring, cnt, bi)].replay()
            else:
                self.body(i0, cnt, bi)
            if after:
                after(bi)


class _TimingEvents:
    """HIP timing events created with hipEventDisableSystemFence (hip_runtime_api.h: 'for events that are only
    being used to measure timing ... avoids the cost of cache writeback and invalidation, and the performance
    impact of those actions on the execution of following work'): torch.cuda.Event records with the default
    system-scope fence, which a 20-step region pays twice.  Falls back to torch events if the runtime refuses."""
    FLAG = 0x20000000

    def __init__(self, torch):
        import ctypes as C
        self.torch, self.C, self.hip = torch, C, None
        try:
            hip = C.CDLL("libamdhip64.so")
            hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
            hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
            hip.hipEventDestroy.argtypes = [C.c_void_p]
            e = C.c_void_p()
            if hip.hipEventCreateWithFlags(C.byref(e), self.FLAG) == 0:
                hip.hipEventDestroy(e)
                self.hip = hip
        except OSError:
            pass

    @property
    def kind(self):
        return "hipEventDisableSystemFence" if self.hip else "torch.cuda.Event (default flags)"

    def pair(self):
        if not self.hip:
            return (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))
        C = self.C
        a, b = C.c_void_p(), C.c_void_p()
        assert self.hip.hipEventCreateWithFlags(C.byref(a), self.FLAG) == 0
        assert self.hip.hipEventCreateWithFlags(C.byref(b), self.FLAG) == 0
        return (a, b)

    def record(self, ev, stream):
        if not self.hip:
            ev.record(stream)
        else:
            assert self.hip.hipEventRecord(ev, self.C.c_void_p(stream.cuda_stream)) == 0

    def elapsed_ms(self, a, b):
        if not self.hip:
            return a.elapsed_time(b)
        ms = self.C.c_float()
        rc = self.hip.hipEventElapsedTime(self.C.byref(ms), a, b)
        self.hip.hipEventDestroy(a)
        self.hip.hipEventDestroy(b)
        assert rc == 0, "hipEventElapsedTime failed (%d)" % rc
        return float(ms.value)


_EVENTS = {}


def time_regions(torch, dist, world, main, repeats, run_one, pad=None):
    """Time `repeats` regions: barrier + synchronize on both sides of each, HIP events on the launch stream
    inside the fences; returns the per-repeat milliseconds, MAX over ranks.  `pad(r)`, if given, enqueues UNTIMED
    steps of the same walk (on ring slots other than the region's) between the opening fence and the start
    event: the device is then busy while the host enqueues the start event and the region's first hipGraph, so a
    short region (the driver's --steps 20) measures launch-to-launch kernel time like a long one instead of
    counting the host's submission latency of its one graph (8.3 vs 7.65 us per step without it)."""
    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    if "ev" not in _EVENTS:
        _EVENTS["ev"] = _TimingEvents(torch)
    ev = _EVENTS["ev"]
    ms, wall = [], []
    import gc
    gc_was = gc.isenabled()
    gc.disable()                                     # an eager region must not lose the host to a collection
    for r in range(repeats):
        e0, e1 = ev.pair()
        fence()
        t0 = time.perf_counter()
        if pad:
            pad(r)
        ev.record(e0, main)
        run_one(r)
        ev.record(e1, main)
        fence()
        wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(ev.elapsed_ms(e0, e1))
    if gc_was:
        gc.enable()
    t = torch.tensor(ms + wall, dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = t.tolist()
    return t[:repeats], t[repeats:]


def median(xs):
    s = sorted(xs)
    return s[len(s) // 2]


def run(args):
    torch, dist, capi, world, rank = init_dist(args)
    if args.workload == "cfg5":
        return run_cfg5(args, torch, dist, capi, world, rank)
    if args.workload == "cfg4":
        return run_cfg4(args, torch, dist, capi, world, rank)

    K, Wm = args.steps, args.warmup
    G = max(1, min(args.group, max(K, 1)))
    ring = 1 if args.warm else max(2, args.ring)
    bt = Batches(torch, ring, args.path, rank)
    step = make_step(capi, bt, args.path)
    buckets = [torch.empty(G, N_PAIRS, 1, 1, 1, device="cuda") for _ in range(2)]
    gathered = [torch.empty(world * G, N_PAIRS, 1, 1, 1, device="cuda") for _ in range(2)] if world > 1 else None
    comm = torch.cuda.Stream() if world > 1 else None
    bucket_free = [torch.cuda.Event() for _ in range(2)]   # gather of that bucket finished
    main = torch.cuda.current_stream()
    raw_step = make_raw_step(capi, bt, args.path, torch) if args.path == "layers" else None
    # auto: short regions (the driver's --steps 20) launch kernel by kernel -- one hipGraphLaunch per region would cost
    # 6.5 us of device time, a third of a microsecond per step (8.05-8.11 vs 7.71 us); long regions keep hipGraphs of
    # --group steps, where the host never has to keep pace (7.57 us in every repeat vs 7.6-8.2 eager)
    # round 3: `auto` is ALWAYS hipGraphs of up to --group steps (the driver's --steps 20 is one graph per region).
    # Round 2 launched short regions kernel by kernel from the host (7.6-7.8 us per step when the host kept pace); a
    # round-3 run on a busier box had the host fall behind in two of five repeats (13 us per step) and a median of
    # 9.8 us: a host-paced number is not a measurement of the kernels.  A graph region costs one hipGraphLaunch
    # (~6.5 us of device time, a third of a microsecond per step at K = 20) and nothing else depends on the host;
    # `roofline.long_region` still reports the same walk over 2048 steps.  `--launch eager` keeps the other mode.
    eager = args.no_graph or args.launch == "eager"
    use_graph = not eager
    reg = Region(torch, step, ring, G, buckets, use_graph, raw_step)

    # eager pass over every code path (instantiates kernels), then the graphs of the warm-up and of
    # every timed repeat -- all captured before anything is timed
    reg.body(0, min(G, ring), 0)
    torch.cuda.synchronize()
    warm_chunks = reg.chunks(0, Wm)
    rep_chunks = [reg.chunks(Wm + r * K, K) for r in range(args.repeats)]
    # untimed lead-in of each repeat: the Gp steps of the walk that PRECEDE the region's first step
    Gp = max(1, min(G, 32, args.lead_in)) if args.lead_in > 0 else 0
    pad_reps = (args.lead_in + Gp - 1) // Gp if Gp else 0
    pad_chunks = [reg.chunks((Wm + r * K - Gp) % ring + ring, Gp) if Gp else [] for r in range(args.repeats)]
    reg.capture([warm_chunks] + rep_chunks)
    for r in range(args.repeats):
        reg.capture_fixed(pad_chunks[r], 0)

    def gather_bucket(bi):
        """All-gather of the per-pair scores of one bucket (up to GROUP steps x 4096 pairs per rank), on the
        side stream, so it overlaps the next chunk's compute."""
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

    before = (lambda bi: main.wait_event(bucket_free[bi])) if world > 1 else None
    after = gather_bucket if world > 1 else None

    if world > 1:
        # communicator set-up (lazy on the first collective) must not land in a timed region
        gather_bucket(0)
        gather_bucket(1)
        torch.cuda.synchronize()
    if not args.warm:
        flush_caches(torch)
    reg.run(warm_chunks, before, after)

    def one(r):
        reg.run(rep_chunks[r], before, after)
        if world > 1:                                  # the last gathers belong to the region
            main.wait_event(bucket_free[0])
            main.wait_event(bucket_free[1])
    def lead_in(r):
        for _ in range(pad_reps):
            reg.run_fixed(pad_chunks[r], 0)       # no gather: nothing reads these scores, and every gather
                                                  # of an earlier region finished before the fence
    ev_ms, wall_ms = time_regions(torch, dist, world, main, args.repeats, one, pad=lead_in if Gp else None)
    t_ms = median(ev_ms)

    out = None
    if rank == 0:
        launches_per_step = {"fused": 1, "layers": 2, "triplet": 1}[args.path]
        step_us = t_ms * 1e3 / K
        achieved = B_UNFUSED / (step_us * 1e-6) / 1e9 if args.path != "triplet" else None
        kernel = {"fused": "mms::euclid_pair32_kernel<75,true,true,...> (SimCross Euclid fwd+bwd, one launch)",
                  "layers": "mms::euclid_pair32_kernel<75,true,false,...> (Forward launch) then mms::euclid_block_kernel<75,false,true,...> (Backward launch)",
                  "triplet": "mms::triplet32_kernel<75,...>"}[args.path]
        mode = capi.get_euclid_backward_mode()
        out = {
            "metric": "QA pairs/sec (fwd+bwd) at batch 4096, 300-d; % HBM roofline",
            "value": world * N_PAIRS * K / (t_ms * 1e-3), "unit": "pairs/s", "n_gpus": world, "steps": K,
            "warmup": Wm, "ms_per_step": t_ms / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: SimCross dist_mode=1 (Euclid) fwd+bwd, q,a (4096,1,300) fp32 per GPU",
                       "pairs_per_gpu": N_PAIRS, "dim": DIM, "global_batch": world * N_PAIRS,
                       "path": args.path, "launches_per_step": launches_per_step,
                       "path_note": {"layers": "Forward launch, then Backward launch: the sequence a Caffe Net / "
                                               "`caffe time` issues through the Layer API",
                                     "fused": "ONE launch for forward+backward; needs top_diff before the forward "
                                              "(not reachable from a Caffe Net; labelled variant)",
                                     "triplet": "fused (q, a+, a-) SimCross x2 + PairRankLoss step"}[args.path],
                       "euclid_backward_arithmetic": mode +
                       (" (scores bit-identical to the CPU code; gradient elements <= 2 ulp from it, bar 1e-5)"
                        if mode == "fp32" else " (gradients bit-identical too)"),
                       "residency": "cache-warm (1 batch)" if args.warm else
                                    "HBM-cold: ring of %d batches (%.2f GiB), caches flushed before the warm-up, "
                                    "timed steps continue the ring walk" % (ring, ring * 19.7e6 / 2**30),
                       "hip_graph_group": G if use_graph else 0,
                       "launch": "hipGraphs of %d steps" % G if use_graph else
                                 "eager: every kernel launched by itself through pre-bound C-ABI calls on the launch "
                                 "stream (host ~2 us per launch, ahead of the device behind the lead-in steps)",
                       "clock": "HIP events (" + _EVENTS["ev"].kind + ") on the launch stream inside barrier+synchronize fences, max over ranks; "
                                "median of %d repeats of the %d-step region; %d untimed lead-in steps of the same "
                                "walk run between each opening fence and its start event (the region's first "
                                "hipGraph is enqueued while the device is busy and at its running clocks; "
                                "--lead-in 0 turns it off)" % (args.repeats, K, pad_reps * Gp),
                       "lead_in_steps_per_repeat": pad_reps * Gp,
                       "repeats_ms_per_step": [x / K for x in ev_ms],
                       "host_wall_ms_per_step_median": median(wall_ms) / K,
                       "ranks_seen": dist.get_world_size() if world > 1 else 1,
                       "parallelism": "pair-sharded x%d%s" % (
                           world, ", %s all-gather of scores per %d steps" % (
                               "RCCL" if args.backend == "nccl" else "host-staged gloo (rehearsal)", G)
                           if world > 1 else "")},
        }
        if achieved is not None:
            traffic = load_traffic(args.path)
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "frac_real": (traffic / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "kernel": kernel,
                "algorithmic_bytes_per_step": B_UNFUSED,
                "algorithmic_bytes_per_launch": {"layers": [B_FWD, B_BWD], "fused": [B_UNFUSED]}[args.path],
                "avg_step_us_hip_events": step_us,
                "note": "achieved = SURVEY 8(d) unfused bytes per step / (median events time / steps), i.e. "
                        "launch-to-launch including inter-kernel gaps; frac_real = PMC HBM bytes per step "
                        "(profiles/traffic.json) over the same time"}
    if world > 1:
        dist.barrier()

    # N > 1: the same walk with the scores all-gathered after EVERY step (no bucketing), for the record
    if world > 1 and not args.no_variants:
        K2 = min(K, 256)
        reg1 = Region(torch, step, ring, 1, buckets, use_graph, raw_step)
        first = Wm + args.repeats * K
        ch = [reg1.chunks(first + r * K2, K2) for r in range(3)]
        reg1.capture(ch)

        def one1(r):
            reg1.run(ch[r], before, after)
            main.wait_event(bucket_free[0])
            main.wait_event(bucket_free[1])
        ev1, _ = time_regions(torch, dist, world, main, 3, one1)
        if rank == 0:
            out["config"]["per_step_gather_variant"] = {
                "ms_per_step": median(ev1) / K2, "value": world * N_PAIRS * K2 / (median(ev1) * 1e-3),
                "note": "one all-gather of 4096 scores per rank after every step (16 KiB messages: latency-bound)"}

    # N > 1, the north-star's split: ONE 4096-pair batch divided over the ranks (strong scaling), each rank runs the
    # Forward and Backward launches on its contiguous 4096/N pairs and the per-pair scores are all-gathered after
    # every step -- latency-bound by construction (SURVEY 8e "scaling caveat"), reported next to the weak line
    if world > 1 and not args.no_variants:
        from mms_answer_selection_amd import sharded
        lo, hi = sharded.shard_range(N_PAIRS, rank, world)
        ns = hi - lo
        K4 = min(K, 256)
        tops = [torch.empty(ns, 1, 1, 1, device="cuda") for _ in range(2)]
        fulls = [torch.empty(N_PAIRS, device="cuda") for _ in range(2)]
        free4 = [torch.cuda.Event() for _ in range(2)]
        cnt4 = [0]

        def strong_step():
            i = cnt4[0]
            cnt4[0] += 1
            bi, sl = i & 1, i % ring
            main.wait_event(free4[bi])
            if ns:
                qs, as_ = bt.q[sl][lo:hi], bt.a[sl][lo:hi]
                capi.simcross_forward(1, qs, as_, tops[bi])
                capi.simcross_backward(1, qs, as_, tops[bi], bt.dT[sl][lo:hi], bt.dq[sl][lo:hi], bt.da[sl][lo:hi])
            comm.wait_stream(main)
            with torch.cuda.stream(comm):
                if args.backend == "nccl":
                    sharded.all_gather_scores(tops[bi].view(ns), N_PAIRS, out=fulls[bi])
                else:
                    fulls[bi].copy_(sharded.all_gather_scores(tops[bi].view(ns).cpu(), N_PAIRS))
                free4[bi].record(comm)

        def one4(r):
            for _ in range(K4):
                strong_step()
            main.wait_event(free4[0])
            main.wait_event(free4[1])
        for _ in range(4):
            strong_step()
        ev4, _ = time_regions(torch, dist, world, main, 3, one4)
        if rank == 0:
            out["config"]["strong_split_variant"] = {
                "scaling": "strong", "pairs_total": N_PAIRS, "pairs_per_gpu": ns, "steps": K4,
                "ms_per_step": median(ev4) / K4, "value": N_PAIRS * K4 / (median(ev4) * 1e-3), "unit": "pairs/s",
                "note": "ONE 4096-pair batch split over the ranks: Forward + Backward launches on 4096/N pairs per "
                        "rank, all-gather of the 4096 scores after every step (launched kernel by kernel; "
                        "%s)" % ("RCCL" if args.backend == "nccl" else "host-staged gloo rehearsal")}

    # a SHORT region (the driver's --steps 20) carries a fixed cost of two event packets and one graph boundary
    # (about 9 us: 0.4 us per step at K = 20, nothing at K = 4096); for the record, the same walk over 2048 steps
    if rank == 0 and world == 1 and K < 1024 and "roofline" in out:
        K3 = 2048
        first = Wm + args.repeats * K
        G3 = max(1, min(args.group, K3))
        reg3 = Region(torch, step, ring, G3, [torch.empty(G3, N_PAIRS, 1, 1, 1, device="cuda") for _ in range(2)],
                      use_graph, raw_step)
        ch3 = [reg3.chunks(first + r * K3, K3) for r in range(3)]
        reg3.capture(ch3)
        ev3, _ = time_regions(torch, dist, world, main, 3, lambda r: reg3.run(ch3[r]), pad=lead_in if Gp else None)
        us3 = median(ev3) * 1e3 / K3
        out["roofline"]["long_region"] = {
            "steps": K3, "us_per_step": us3, "frac": B_UNFUSED / (us3 * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "note": "same path, same clock, one region of %d steps (median of 3): what the per-step time is once "
                    "the two event packets and the graph boundary of a region are amortised" % K3}

    # side measurements on rank 0 at N=1 only (not part of the timed region above)
    if rank == 0 and world == 1 and not args.no_variants:
        if "roofline" in out:
            out["roofline"].update(per_kernel_roofline(torch, capi))
            fl = out["roofline"]["launch_floor"]["us_per_empty_launch"] * launches_per_step + \
                (out["roofline"]["launch_floor"]["us_per_graph_launch"] / max(G, 1) if use_graph else 0.0)
            above = max(out["roofline"]["avg_step_us_hip_events"] - fl, 1e-3)
            out["roofline"]["launch_floor"].update({
                "launches_per_step": launches_per_step, "steps_per_graph": G if use_graph else 0, "us_per_step": fl,
                "step_us_above_launch_floor": above,
                "GBps_above_launch_floor": B_UNFUSED / (above * 1e-6) / 1e9,
                "frac_ceiling_if_data_were_free": B_UNFUSED / (fl * 1e-6) / 1e9 / HBM_PEAK_GBS})
        out["variants"] = variants(torch, capi, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def per_kernel_roofline(torch, capi):
    """The two launches of the Layer sequence timed on their own (HBM-cold ring, hipGraph of 32 launches,
    events on the launch stream): SURVEY 8(d) bytes per launch / average launch-to-launch time."""
    bt = Batches(torch, 64, "layers", 0)
    top = torch.empty(64, N_PAIRS, 1, 1, 1, device="cuda")
    res = {}
    for name, nbytes, fn in (
            ("forward", B_FWD, lambda i: capi.simcross_forward(1, bt.q[i], bt.a[i], top[i])),
            ("backward", B_BWD, lambda i: capi.simcross_backward(1, bt.q[i], bt.a[i], top[i], bt.dT[i],
                                                                   bt.dq[i], bt.da[i]))):
        for i in range(64):
            fn(i)
        torch.cuda.synchronize()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        graphs = []
        with torch.cuda.stream(cap):
            for g0 in (0, 32):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=cap):
                    for i in range(g0, g0 + 32):
                        fn(i)
                graphs.append(gph)
        torch.cuda.current_stream().wait_stream(cap)
        flush_caches(torch)
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(8):
                graphs[r & 1].replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 256)
        us = median(ts)
        res[name] = {"us_per_launch": us, "algorithmic_bytes": nbytes,
                     "achieved_GBps": nbytes / (us * 1e-6) / 1e9,
                     "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    # what a launch costs by itself under the same protocol: graphs of 16 and of 128 EMPTY kernels (256 x 512
    # threads) replayed back to back -- the per-kernel floor inside a graph and the cost of a hipGraphLaunch boundary
    # separate as the slope and the intercept (tools/graphbound.hip: a graph of n empty kernels takes 6.5 + 1.56 n us)
    per = {}
    for nk in (16, 128):
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=cap):
                for i in range(nk):
                    capi.null_launch(256)
        torch.cuda.current_stream().wait_stream(cap)
        gph.replay()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(16):
                gph.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 16)
        per[nk] = median(ts)
    floor = (per[128] - per[16]) / 112.0
    boundary = per[16] - 16 * floor
    return {"per_kernel_cold": res,
            "launch_floor": {"us_per_empty_launch": floor, "us_per_graph_launch": boundary,
                             "note": "EMPTY kernels (mms_null_launch, 256 x 512 threads) in hipGraphs of 16 and 128, "
                                     "replayed back to back: slope = what a kernel node costs by itself, intercept = what "
                                     "a hipGraphLaunch costs by itself; a step of L launches in graphs of G steps cannot "
                                     "take less than L x slope + intercept / G"}}


def run_cfg5(args, torch, dist, capi, world, rank):
    """BASELINE cfg 5: 65,536 pairs x 1024-d, fp16 storage / fp32 arithmetic, SimCross Euclid fwd+bwd, the batch
    SPLIT over the ranks (strong scaling): every rank runs the fused launch on its contiguous shard and the
    per-pair scores are all-gathered after every step (256 KiB in all) on a side stream, double-buffered."""
    from mms_answer_selection_amd import sharded
    NT, D = 65536, 1024
    capi.set_f16_distance_mode(args.f16_distance)
    lo, hi = sharded.shard_range(NT, rank, world)
    n = hi - lo
    ring = max(2, min(16, (1 << 30) // (n * D * 2 * 4)))      # ~1 GiB of distinct shards per rank
    g = torch.Generator(device="cuda").manual_seed(1701 + rank)
    q = (torch.randn(ring, n, 1, D, device="cuda", generator=g) * 0.4).half()
    a = (torch.randn(ring, n, 1, D, device="cuda", generator=g) * 0.4).half()
    dT = torch.randn(ring, n, 1, 1, 1, device="cuda", generator=g)
    dq, da = torch.empty_like(q), torch.empty_like(a)
    tops = [torch.empty(n, 1, 1, 1, device="cuda") for _ in range(2)]
    full = [torch.empty(NT, device="cuda") for _ in range(2)]
    main = torch.cuda.current_stream()
    comm = torch.cuda.Stream() if world > 1 else None
    free = [torch.cuda.Event() for _ in range(2)]
    K, Wm = args.steps, args.warmup
    cnt = [0]

    def step():
        i = cnt[0]
        cnt[0] += 1
        bi = i & 1
        if world > 1:
            main.wait_event(free[bi])
        s = i % ring
        capi.simcross_euclid_forward_backward_f16(q[s], a[s], dT[s], tops[bi], dq[s], da[s])
        if world > 1:
            comm.wait_stream(main)
            with torch.cuda.stream(comm):
                if args.backend == "nccl":
                    sharded.all_gather_scores(tops[bi].view(n), NT, out=full[bi])
                else:
                    full[bi].copy_(sharded.all_gather_scores(tops[bi].view(n).cpu(), NT))
                free[bi].record(comm)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    flush_caches(torch)
    for _ in range(Wm):
        step()

    def one(r):
        for _ in range(K):
            step()
        if world > 1:
            main.wait_event(free[0])
            main.wait_event(free[1])
    ev_ms, wall_ms = time_regions(torch, dist, world, main, args.repeats, one)
    t_ms = median(ev_ms)
    if rank == 0:
        b_unfused = 2 * (3 * NT * 2 * D) + 4 * 3 * NT                 # SURVEY 8(d), s = 2, whole batch
        b_moved = 2 * (2 * NT * 2 * D) + 4 * 2 * NT                   # the fused launch: q, a read once; dq, da written once
        step_us = t_ms * 1e3 / K
        out = {"metric": "QA pairs/sec (fwd+bwd), cfg 5: 65,536 x 1024 fp16 storage; % HBM roofline",
               "value": NT * K / (t_ms * 1e-3), "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": Wm,
               "ms_per_step": t_ms / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f16 storage / f32 arithmetic", "data": "synthetic",
               "config": {"workload": "cfg5: SimCross Euclid fwd+bwd, 65,536 pairs x 1024-d fp16 storage, batch split "
                                      "over %d rank(s), scores all-gathered every step" % world,
                          "pairs_total": NT, "pairs_per_gpu": n, "dim": D, "ranks_seen":
                              dist.get_world_size() if world > 1 else 1,
                          "repeats_ms_per_step": [x / K for x in ev_ms],
                          "parallelism": "pair-sharded x%d (contiguous shards), %s" % (
                              world, "RCCL all-gather per step" if args.backend == "nccl" else "gloo rehearsal")},
               "roofline": {"bound": "hbm", "achieved": b_unfused / (step_us * 1e-6) / 1e9 / world,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": b_unfused / (step_us * 1e-6) / 1e9 / world / HBM_PEAK_GBS, "traffic": None,
                            "moved_bytes_per_step_per_gpu": b_moved / world,
                            "frac_on_moved_bytes": b_moved / (step_us * 1e-6) / 1e9 / world / HBM_PEAK_GBS,
                            "f16_distance": args.f16_distance,
                            "note": "per GPU: SURVEY 8(d) unfused bytes of the whole batch / ranks / step time; "
                                    "frac_on_moved_bytes: the bytes the one fused launch actually moves (q, a read once; "
                                    "dq, da written once) over the same time"}}
        out["config"]["f16_distance"] = args.f16_distance + (
            " (scores bit-identical to the fp32 CPU sum of the fp16-rounded inputs)" if args.f16_distance == "ordered"
            else " (fixed tree sum, ~1e-6 relative; the configuration's bar is 1e-3)")
        print(json.dumps(out), flush=True)
    capi.set_f16_distance_mode("ordered")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_cfg4(args, torch, dist, capi, world, rank):
    """BASELINE cfg 4: scoring only.  1,517 candidates (40 x 40 word grids, Dw = 50: the TREC-QA test split) are
    sharded over the ranks; each rank runs the SimCross forward on its shard and reduces every candidate's map to
    one score; the scores are all-gathered and EVERY rank computes MAP and MRR over the 68 question groups on its
    GPU.  Asserted before timing: the gathered scores equal the unsharded scoring bit for bit on every rank (hence
    so does any ranking), and MAP / MRR agree across ranks."""
    import numpy as np
    from mms_answer_selection_amd import sharded
    n, Wd, D, groups = 1517, 40, 50, 68
    g = torch.Generator(device="cuda").manual_seed(1701)            # the SAME candidates on every rank
    qa = torch.randn(n, Wd, D, device="cuda", generator=g) * 0.4
    aa = torch.randn(n, Wd, D, device="cuda", generator=g) * 0.4
    label = (torch.rand(n, device="cuda", generator=g) < 0.2).float()
    grp = torch.sort(torch.randint(0, groups, (n,), device="cuda", generator=g).float()).values
    lo, hi = sharded.shard_range(n, rank, world)
    m = hi - lo
    q, a = qa[lo:hi].contiguous(), aa[lo:hi].contiguous()
    top = torch.empty(m, 1, Wd, Wd, device="cuda")
    full = torch.empty(n, device="cuda")

    # every buffer of a step is allocated once: the timed step is launches only (no allocation, no host copy)
    s_loc = torch.empty(m, device="cuda")
    prob = torch.zeros(n, 2, device="cuda")          # MAP / MRR read column fixed_axis = 1 only (map_layer.cpp:50)
    res = torch.empty(2, device="cuda")              # MAP, MRR -- left on the device, like a Layer's Forward_gpu
    eff = torch.empty(1, dtype=torch.int32, device="cuda")
    ws = capi.Workspace()

    def step():
        capi.simcross_forward(1, q, a, top)
        torch.amax(top.view(m, -1), dim=1, out=s_loc)      # one score per candidate: its best word-pair similarity
        if world > 1:
            if args.backend == "nccl":
                sharded.all_gather_scores(s_loc, n, out=full)
            else:
                full.copy_(sharded.all_gather_scores(s_loc.cpu(), n))
            prob[:, 1].copy_(full)
        else:
            prob[:, 1].copy_(s_loc)
        capi.rank_map_mrr_device(prob, label, grp, res, eff, ws=ws)

    # ranking identity: sharded == unsharded, on every rank (host copies here, before anything is timed)
    step()
    torch.cuda.synchronize()
    mp, mrr = (float(x) for x in res.tolist())
    full_now = prob[:, 1].contiguous()
    top_all = torch.empty(n, 1, Wd, Wd, device="cuda")
    capi.simcross_forward(1, qa, aa, top_all)
    ref = top_all.view(n, -1).amax(dim=1)
    same = bool((full_now.view(torch.int32) == ref.view(torch.int32)).all().item())
    prob_ref = torch.zeros(n, 2, device="cuda")
    prob_ref[:, 1].copy_(ref)
    mp_ref, mrr_ref, _ = capi.rank_map_mrr(prob_ref, label, grp)
    same = same and mp == mp_ref and mrr == mrr_ref
    flag = torch.tensor([1.0 if same else 0.0, mp, -mp, mrr, -mrr], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    f = flag.tolist()
    identical = f[0] == 1.0 and f[1] == -f[2] and f[3] == -f[4]
    if not identical:
        raise SystemExit("cfg4: sharded scoring differs from the unsharded one (scores same=%s, MAP %r, MRR %r)"
                         % (same, mp, mrr))
    K, Wm = args.steps, args.warmup
    for _ in range(Wm):
        step()
    main = torch.cuda.current_stream()
    ev_ms, wall_ms = time_regions(torch, dist, world, main, args.repeats, lambda r: [step() for _ in range(K)])
    # device-resident steps: HIP events on the launch stream are the clock (the gloo rehearsal stages the gather
    # through the host, so there the host wall is)
    device_only = world == 1 or args.backend == "nccl"
    t_ms = median(ev_ms) if device_only else median(wall_ms)
    if rank == 0:
        out = {"metric": "candidates scored + ranked per second, cfg 4: 1,517 x (40 x 40 x 50) SimCross forward -> "
                         "all-gather -> MAP/MRR",
               "value": n * K / (t_ms * 1e-3), "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": Wm,
               "ms_per_step": t_ms / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "cfg4: scoring only, 1,517 candidates sharded over %d rank(s)" % world,
                          "candidates": n, "candidates_per_gpu": m, "groups": groups, "ranks_seen":
                              dist.get_world_size() if world > 1 else 1,
                          "ranking_identity": "gathered scores bit-identical to the unsharded scoring on every rank; "
                                              "MAP %.6f / MRR %.6f equal on all ranks" % (mp, mrr),
                          "clock": ("HIP events on the launch stream inside barrier+synchronize fences: the step is "
                                    "launches only -- scoring, per-candidate max, all-gather, MAP/MRR with the results "
                                    "left on the device; " if device_only else
                                    "host wall inside barrier+synchronize fences (gloo rehearsal: the gather is staged "
                                    "through the host); ") + "median of %d repeats" % args.repeats,
                          "host_wall_ms_per_step_median": median(wall_ms) / K,
                          "parallelism": "pair-sharded x%d, %s" % (
                              world, "RCCL all-gather per step" if args.backend == "nccl" else "gloo rehearsal")}}
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
    # 128 steps per hipGraph: a hipGraphLaunch costs ~6.5 us of device time by itself (tools/graphbound.hip), which at
    # 16 steps per graph (round 1) added 0.4 us to every per-step figure below
    K, G = 1024, 128
    for name, path, ring in (("fused_cold", "fused", 64), ("fused_warm", "fused", 1),
                             ("fused_cold_reference_rounding_bwd", "fused", 64),
                             ("layers_cold", "layers", 64), ("layers_warm", "layers", 1),
                             ("cosine_fused_cold", "cosine", 64),
                             ("triplet_cold", "triplet", 48),
                             ("triplet_cold_without_loss_scalar", "triplet_noloss", 48)):
        ref_mode = name.endswith("reference_rounding_bwd")
        if path == args.path and ((ring == 1) == args.warm) and not ref_mode:
            continue
        # the launcher picks the kernel variant at capture time (include/mms.h)
        capi.set_euclid_backward_mode("reference" if ref_mode else DEFAULT_BWD_MODE)
        bt = Batches(torch, ring, path, 0)
        step = make_step(capi, bt, path)
        top = torch.empty(G, N_PAIRS, 1, 1, 1, device="cuda")
        ng = 1 if ring == 1 else 3                    # three graphs walk the ring (G is a multiple of it) in turn
        graphs = []
        for s in range(min(G, max(ring, 1))):
            step(s % ring, top[s])
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
        if ring > 1:
            flush_caches(torch)
        for i in range(8):
            graphs[i % ng].replay()
        torch.cuda.synchronize()
        ts = []
        nxt = 8
        for rep in range(5):                            # median of 5 repeats, continuing the ring walk
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(K // G):
                graphs[(nxt + i) % ng].replay()
            e1.record()
            torch.cuda.synchronize()
            nxt += K // G
            ts.append(e0.elapsed_time(e1) * 1e3 / K)
        us = median(ts)
        pairs = N_PAIRS
        res[name] = {"us_per_step": us, "pairs_per_s": pairs / (us * 1e-6)}
        if not path.startswith("triplet"):
            res[name]["frac_hbm_unfused_bytes"] = B_UNFUSED / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
        del bt, graphs
        torch.cuda.empty_cache()
    capi.set_euclid_backward_mode(DEFAULT_BWD_MODE)
    res.update(other_configs(torch, capi))
    return res


def _graph_time(torch, fn, iters=64, reps=6):
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
        capi.simmatrix_forward(q, a, W, top, scr, ws=ws)
        capi.simmatrix_backward(q, a, W, dT, dq, da, dW, ws=ws, qw=scr)
    flops = 2.0 * N * K * K + 2.0 * N * K + 6.0 * N * K * K        # the reference's 4 products
    done = 2.0 * N * K * K + 2.0 * N * K + 4.0 * N * K * K         # executed: Q.W once
    # round 3: the three products run on the BF16 matrix pipe at fp32 accuracy (every fp32 operand = the exact sum of
    # three bf16 values; six bf16 products per fp32 product, fp32 accumulate: csrc/bx3_gemm.h) -- the default -- next
    # to the fp32-MFMA pipe of rounds 1-2 (mms_set_matrix_mode(1)).  Same 1e-5 parity tests on both.
    us = _graph_time(torch, cfg3, iters=16)
    bf16_done = 6.0 * 3 * 2.0 * N * 304 * 320                       # bf16 flop actually issued (padded tiles, 6 products)
    out["cfg3_simmatrix_16384x300x300_fwd_bwd"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6), "TFLOPs": flops / us / 1e6,
        "frac_mfma_fp32_peak": flops / (us * 1e-6) / 157.3e12,
        "executed_TFLOPs": done / us / 1e6, "executed_frac_mfma_fp32_peak": done / (us * 1e-6) / 157.3e12,
        "bf16_pipe_TFLOPs_issued": bf16_done / us / 1e6, "frac_mfma_bf16_dense_peak": bf16_done / (us * 1e-6) / 2.5e15,
        "bound": "mfma (bf16 pipe, 6 bf16 products per fp32 product) / L2->LDS operand stream / HBM epilogue",
        "dtype": "f32 (exact 3-way bf16 splits, fp32 accumulate)", "matrix_mode": capi.get_matrix_mode()}
    capi.set_matrix_mode("fp32")
    us = _graph_time(torch, cfg3, iters=16)
    capi.set_matrix_mode("bf16x3")
    out["cfg3_simmatrix_fp32_mfma_pipe"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6), "TFLOPs": flops / us / 1e6,
        "frac_mfma_fp32_peak": flops / (us * 1e-6) / 157.3e12,
        "executed_TFLOPs": done / us / 1e6, "executed_frac_mfma_fp32_peak": done / (us * 1e-6) / 157.3e12,
        "bound": "mfma", "dtype": "f32", "matrix_mode": "fp32"}

    # fp16-STORAGE scoring with the learned metric (round 3): q, a as halves (K1 padded to 304 = a multiple of 8), W fp32,
    # scores only -- the bilinear member of the fp16-storage family (mms_simmatrix_forward_f16)
    q16 = torch.zeros(N, 304, device="cuda", dtype=torch.float16); q16[:, :K] = q.half()
    a16 = a.half()
    W16 = torch.zeros(304, K, device="cuda"); W16[:K] = W
    us = _graph_time(torch, lambda: capi.simmatrix_forward_f16(q16, a16, W16, top, ws=ws), iters=16)
    out["simmatrix_scoring_16384x304x300_fp16_storage"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6), "TFLOPs": (2.0 * N * K * K + 2.0 * N * K) / us / 1e6,
        "dtype": "f16 storage / exact bf16 x 2 x 3 products, f32 accumulate", "bound": "mfma (bf16 pipe) / L2->LDS operand stream",
        "note": "mms_simmatrix_forward_f16: scores only (no Q.W output); the fp32-storage forward above writes Q.W too"}
    # ... and the training pair on 304 x 304 (both inner dimensions multiples of 8), next to the fp32-storage step of that shape
    a16b = torch.zeros(N, 304, device="cuda", dtype=torch.float16); a16b[:, :K] = a.half()
    W16b = torch.zeros(304, 304, device="cuda"); W16b[:K, :K] = W
    qw16, dW16 = torch.empty(N, 304, device="cuda"), torch.zeros(304, 304, device="cuda")
    dq16, da16 = torch.empty_like(q16), torch.empty_like(a16b)

    def step16():
        capi.simmatrix_forward_train_f16(q16, a16b, W16b, top, qw16, ws=ws)
        capi.simmatrix_backward_f16(q16, a16b, W16b, qw16, dT, dq16, da16, dW16, ws=ws)
    us16 = _graph_time(torch, step16, iters=16)
    q32b, a32b = q16.float(), a16b.float()
    dq32b, da32b = torch.empty_like(q32b), torch.empty_like(a32b)

    def step32():
        capi.simmatrix_forward(q32b, a32b, W16b, top, qw16, ws=ws)
        capi.simmatrix_backward(q32b, a32b, W16b, dT, dq32b, da32b, dW16, ws=ws, qw=qw16)
    us32 = _graph_time(torch, step32, iters=16)
    out["simmatrix_training_16384x304x304_fp16_storage"] = {
        "us_per_step": us16, "us_per_step_fp32_storage_same_shape": us32, "pairs_per_s": N / (us16 * 1e-6),
        "dtype": "f16 storage (q, a, dq, da) / f32 W, dW, scores; exact bf16 splits, f32 accumulate",
        "note": "mms_simmatrix_forward_train_f16 + mms_simmatrix_backward_f16"}
    del q16, a16, W16, a16b, W16b, qw16, dW16, dq16, da16, q32b, a32b, dq32b, da32b

    def cfg3_nocache():
        capi.simmatrix_forward(q, a, W, top, scr, ws=ws)
        capi.simmatrix_backward(q, a, W, dT, dq, da, dW, ws=ws)
    us = _graph_time(torch, cfg3_nocache, iters=16)
    out["cfg3_simmatrix_recomputing_backward"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6), "TFLOPs": flops / us / 1e6,
        "frac_mfma_fp32_peak": flops / (us * 1e-6) / 157.3e12, "bound": "mfma", "dtype": "f32"}

    # the fused learned-metric TRIPLET step (round 3): SimMatrix(q, a+), SimMatrix(q, a-) with W shared -> PairRankLoss,
    # forward and backward, as one call = three products (Q.W once, dq = B.W^T, dW += Q^T.B) with the hinge and its
    # gradient in the first product's epilogue -- next to the same net run layer by layer through the C ABI
    an = rnd(N, K)
    yl = (torch.rand(N, 1, device="cuda", generator=g) < 0.8).float()
    sp, sn, ls = torch.empty(N, 1, device="cuda"), torch.empty(N, 1, device="cuda"), torch.empty(1, device="cuda")
    dan, dq2, scr2 = torch.empty_like(a), torch.empty_like(q), torch.empty(N, K, device="cuda")
    po, ps, gsp, gsn = (torch.empty(N, 1, device="cuda") for _ in range(4))
    ws2 = capi.Workspace()

    def trip_fused():
        capi.triplet_simmatrix_step(q, a, an, yl, W, sp, sn, ls, dq, da, dan, dW, margin=0.3, ws=ws2)
    us = _graph_time(torch, trip_fused, iters=16)
    done3 = 3 * 2.0 * N * K * K
    credited = 2 * (2.0 * N * K * K + 2.0 * N * K) + 2 * 6.0 * N * K * K     # two SimMatrix layers, fwd + bwd
    out["cfg3_triplet_simmatrix_16384x300x300_fused_step"] = {
        "us_per_step": us, "triplets_per_s": N / (us * 1e-6), "executed_TFLOPs": done3 / us / 1e6,
        "executed_frac_mfma_fp32_peak": done3 / (us * 1e-6) / 157.3e12,
        "TFLOPs_credited_layer_by_layer": credited / us / 1e6, "bound": "mfma", "dtype": "f32",
        "note": "mms_triplet_simmatrix_step_f32: 3 products executed (8.85 GFLOP) for the 8 the two layers' forward + "
                "backward are credited with (23.6 GFLOP)"}

    def trip_layers():
        capi.simmatrix_forward(q, a, W, sp, scr)
        capi.simmatrix_forward(q, an, W, sn, scr2)
        capi.pairrank_forward(sp, sn, yl, po, ps, ls, margin=0.3, ws=ws)
        capi.pairrank_backward(yl, po, ps, gsp, gsn)
        capi.simmatrix_backward(q, a, W, gsp, dq, da, dW, ws=ws, qw=scr)
        capi.simmatrix_backward(q, an, W, gsn, dq2, dan, dW, ws=ws, qw=scr2)
    us = _graph_time(torch, trip_layers, iters=8)
    out["cfg3_triplet_simmatrix_layer_by_layer"] = {
        "us_per_step": us, "triplets_per_s": N / (us * 1e-6), "TFLOPs_credited_layer_by_layer": credited / us / 1e6,
        "note": "the same net through the per-layer entry points (8 launches of products + PairRankLoss fwd/bwd; "
                "dq's Split sum not included)"}
    del an, yl, sp, sn, ls, dan, dq2, scr2, po, ps, gsp, gsn

    # the same arithmetic written as a SimCross layer: dist_mode 2, one measure, W1 = W2 = 1 (bias_term false, like
    # SimMatrix; with the scalar bias its gradient -- an n-ordered sum of 16384 terms, one dependent chain -- adds ~120 us)
    q3, a3, W3 = q.view(N, 1, K), a.view(N, 1, K), W.view(1, K, K)
    t3, dT3 = torch.empty(N, 1, 1, 1, device="cuda"), dT.view(N, 1, 1, 1)
    dq3, da3, dW3 = torch.empty_like(q3), torch.empty_like(a3), torch.empty_like(W3)

    def cfg3_simcross():
        capi.simcross_forward(2, q3, a3, t3, W=W3, ws=ws)
        capi.simcross_backward(2, q3, a3, t3, dT3, dq3, da3, W=W3, bias_term=False, dW=dW3, ws=ws)
    us = _graph_time(torch, cfg3_simcross, iters=16)
    out["cfg3_as_simcross_bilinear_16384x1x1x300_M1_fwd_bwd"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6), "TFLOPs": flops / us / 1e6,
        "frac_mfma_fp32_peak": flops / (us * 1e-6) / 157.3e12, "bound": "mfma", "dtype": "f32",
        "note": "SimCross dist_mode 2 at W1 = W2 = 1, M = 1: SimMatrix's panel-GEMM launches, Q.W recomputed in the backward"}
    del q, a, W, dT, top, scr, dq, da, dW, q3, a3, W3, t3, dT3, dq3, da3, dW3
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
        us = _graph_time(torch, cfg1, iters=32)
        fl = 2.0 * N1 * M1 * Wd * D1 * (D1 + Wd) + 8.0 * N1 * M1 * Wd * D1 * (D1 + Wd)
        out["cfg1_bilinear_%dx%dx%dx%d_M%d_fwd_bwd" % (N1, Wd, Wd, D1, M1)] = {
            "us_per_step": us, "pairs_per_s": N1 / (us * 1e-6), "TFLOPs": fl / us / 1e6,
            "frac_mfma_fp32_peak": fl / (us * 1e-6) / 157.3e12, "bound": "mfma / launch", "dtype": "f32"}
        del q1, a1, W1, b1, t1, dT1, dq1, da1, dW1, db1
    # The real training step of the reference's network_v4 through the library's own layers (do_trec_qa_clean.py:452-470
    # read as data): word ids (50, 40) x 2 -> Embed x 2 (one shared 50-d table, with bias) -> SimCross dist_mode 2,
    # M = 4, bias -> backward -> Embed backward x 2 into the shared table's diff.  Batch 50, vocabulary 20,000.
    Bt, Wd, Dv, Mv, Kv = 50, 40, 50, 4, 20000
    tab = rnd(Kv, Dv)
    ebias = torch.zeros(Dv, device="cuda")
    iq = torch.randint(0, Kv, (Bt, Wd), device="cuda", generator=g).float()
    ia = torch.randint(0, Kv, (Bt, Wd), device="cuda", generator=g).float()
    iq[:, 30:] = Kv - 1                                                   # zero-pad id: sentences are shorter than 40 words
    ia[:, 34:] = Kv - 1
    qe, ae = torch.empty(Bt, Wd, Dv, device="cuda"), torch.empty(Bt, Wd, Dv, device="cuda")
    Wv = torch.rand(Mv, Dv, Dv, device="cuda", generator=g) * 0.16 - 0.08
    bv = torch.zeros(Mv, Wd, Wd, device="cuda")
    tv = torch.empty(Bt, Mv, Wd, Wd, device="cuda")
    dtv = torch.randn(Bt, Mv, Wd, Wd, device="cuda", generator=g)
    dqe, dae = torch.empty_like(qe), torch.empty_like(ae)
    dWv, dbv = torch.empty_like(Wv), torch.zeros_like(bv)
    dtab, debias = torch.zeros_like(tab), torch.zeros_like(ebias)
    wsv = capi.Workspace()

    pidx = capi.EmbedPairIndex()

    def v4_step():
        # both Embed forwards in one launch, the inverted index of the word ids built beside them (it depends on the
        # ids alone); Net::Backward reaches the later layer (w2v_a) first, so it is layer 0 of the pair in both calls
        capi.embed_forward_pair(ia, iq, tab, ae.view(-1, Dv), qe.view(-1, Dv), bias=ebias, index=pidx)
        capi.simcross_forward(2, qe, ae, tv, W=Wv, bias=bv, ws=wsv)
        capi.simcross_backward(2, qe, ae, tv, dtv, dqe, dae, W=Wv, bias_term=True, dW=dWv, dbias=dbv, ws=wsv)
        capi.embed_backward_pair_indexed(ia, iq, dae.view(-1, Dv), dqe.view(-1, Dv), dtab, pidx, bias_diff=debias)
    us = _graph_time(torch, v4_step, iters=16)

    def v4_step_plain():
        capi.embed_forward(iq, tab, qe.view(-1, Dv), bias=ebias)
        capi.embed_forward(ia, tab, ae.view(-1, Dv), bias=ebias)
        capi.simcross_forward(2, qe, ae, tv, W=Wv, bias=bv, ws=wsv)
        capi.simcross_backward(2, qe, ae, tv, dtv, dqe, dae, W=Wv, bias_term=True, dW=dWv, dbias=dbv, ws=wsv)
        capi.embed_backward(ia, dae.view(-1, Dv), dtab, bias_diff=debias, ws=wsv)
        capi.embed_backward(iq, dqe.view(-1, Dv), dtab, bias_diff=debias, ws=wsv)
    us_plain = _graph_time(torch, v4_step_plain, iters=16)
    parts = {}
    for nm, fn in (("embed_forward_x2", lambda: (capi.embed_forward(iq, tab, qe.view(-1, Dv), bias=ebias),
                                                 capi.embed_forward(ia, tab, ae.view(-1, Dv), bias=ebias))),
                   ("simcross_forward", lambda: capi.simcross_forward(2, qe, ae, tv, W=Wv, bias=bv, ws=wsv)),
                   ("simcross_backward", lambda: capi.simcross_backward(2, qe, ae, tv, dtv, dqe, dae, W=Wv, bias_term=True,
                                                                        dW=dWv, dbias=dbv, ws=wsv)),
                   ("embed_backward_x2_as_two_calls", lambda: (capi.embed_backward(ia, dae.view(-1, Dv), dtab, bias_diff=debias, ws=wsv),
                                                               capi.embed_backward(iq, dqe.view(-1, Dv), dtab, bias_diff=debias, ws=wsv))),
                   ("embed_backward_pair", lambda: capi.embed_backward_pair(ia, iq, dae.view(-1, Dv), dqe.view(-1, Dv), dtab,
                                                                            bias_diff=debias, ws=wsv)),
                   ("embed_forward_pair_with_index", lambda: capi.embed_forward_pair(ia, iq, tab, ae.view(-1, Dv), qe.view(-1, Dv),
                                                                                     bias=ebias, index=pidx)),
                   ("embed_backward_pair_indexed", lambda: capi.embed_backward_pair_indexed(ia, iq, dae.view(-1, Dv), dqe.view(-1, Dv),
                                                                                           dtab, pidx, bias_diff=debias))):
        parts[nm] = _graph_time(torch, fn, iters=16)
    out["network_v4_training_step_batch50"] = {
        "us_per_step": us, "pairs_per_s": Bt / (us * 1e-6), "us_per_step_layer_by_layer_calls": us_plain, "parts_us": parts,
        "note": "Embed x2 (one launch, inverted index of the word ids built beside the gathers) -> SimCross bilinear M=4 + "
                "bias -> backward -> both Embed backwards as one pass over the shared table from that index, through the C "
                "ABI, graph-replayed; us_per_step_layer_by_layer_calls: the same step as six per-layer calls; parts alone"}
    del tab, iq, ia, qe, ae, Wv, bv, tv, dtv, dqe, dae, dWv, dbv, dtab

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
        us = _graph_time(torch, prl, iters=32)
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
    b_unfused = 2 * (3 * N * 2 * D) + 4 * 3 * N                      # SURVEY 8(d), s = 2: 100.7 MB
    b_moved = 2 * (2 * N * 2 * D) + 4 * 2 * N                        # what the fused launch moves: 67.2 MB
    for mode, key in (("ordered", "cfg5_shard_8192x1024_fp16_storage_fused"),
                      ("tree", "cfg5_shard_8192x1024_fp16_storage_fused_tree")):
        capi.set_f16_distance_mode(mode)                             # the launcher picks the kernel at capture time
        us = _graph_time(torch, lambda: capi.simcross_euclid_forward_backward_f16(qh, ah, dT, top, dqh, dah))
        out[key] = {
            "us_per_step": us, "pairs_per_s": N / (us * 1e-6),
            "frac_hbm_unfused_bytes": b_unfused / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "frac_hbm_moved_bytes": b_moved / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "bound": "hbm",
            "dtype": "f16 storage / f32 arithmetic",
            "f16_distance": mode + (": the reference's d-ascending fp32 sum, scores bit-identical" if mode == "ordered"
                                    else ": fixed tree sum, ~1e-6 relative (the configuration's bar is 1e-3)")}
    capi.set_f16_distance_mode("ordered")
    # the same shard with dist_mode 0 (cosine): norms cached as the reference does, one launch
    n0c, n1c = torch.empty(N, 1, device="cuda"), torch.empty(N, 1, device="cuda")
    us = _graph_time(torch, lambda: capi.simcross_cosine_forward_backward_f16(qh, ah, dT, top, dqh, dah, n0c, n1c))
    out["cfg5_shard_8192x1024_fp16_storage_cosine_fused"] = {
        "us_per_step": us, "pairs_per_s": N / (us * 1e-6),
        "frac_hbm_moved_bytes": (b_moved + 8.0 * N) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "bound": "hbm",
        "dtype": "f16 storage / f32 arithmetic"}
    del qh, ah, dT, top, dqh, dah, n0c, n1c
    # cfg 4: scoring only -- the TREC-QA test split (1517 candidates, 40 x 40 word grids, Dw = 50),
    # whole and as the 190-candidate shard one of 8 GPUs scores, plus MAP + MRR on 1517 sentence scores
    for name, n in (("cfg4_scoring_1517x40x40x50_forward", 1517), ("cfg4_shard_190x40x40x50_forward", 190)):
        qg, ag = rnd(n, 40, 50), rnd(n, 40, 50)
        tg = torch.empty(n, 1, 40, 40, device="cuda")
        us = _graph_time(torch, lambda: capi.simcross_forward(1, qg, ag, tg))
        b = 4.0 * (2 * n * 40 * 50 + n * 1600)                       # SURVEY 8(d) B_fwd
        out[name] = {"us_per_step": us, "pairs_per_s": n / (us * 1e-6), "GBps_algorithmic": b / us / 1e3,
                     "bound": "fp32 VALU (3 flop per (j,k,d), d-ordered sums)"}
        if n == 1517:       # ... and the BACKWARD of the same geometry (training on word grids; default fp32 backward arithmetic)
            dtg = torch.randn(n, 1, 40, 40, device="cuda", generator=g)
            dqg, dag = torch.empty_like(qg), torch.empty_like(ag)
            capi.simcross_forward(1, qg, ag, tg)
            usb = _graph_time(torch, lambda: capi.simcross_backward(1, qg, ag, tg, dtg, dqg, dag))
            out["cfg4_euclid_backward_1517x40x40x50"] = {
                "us_per_step": usb, "pairs_per_s": n / (usb * 1e-6),
                "note": "cross_bwd_lane_kernel, two waves per pair; every (j,k,d) term formed once",
                "bound": "LDS return path + fp32 VALU issue"}
            del dtg, dqg, dag
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
    # ... and in the mode network_v4 scores with (bilinear, M = 4, bias)
    Wb = torch.rand(4, 50, 50, device="cuda", generator=g) * 0.16 - 0.08
    bb = torch.zeros(4, 40, 40, device="cuda")
    tb = torch.empty(n, 4, 40, 40, device="cuda")

    def embed_then_bilinear():
        capi.embed_forward(iq, tab, qe.view(-1, 50))
        capi.embed_forward(ia, tab, ae.view(-1, 50))
        capi.simcross_forward(2, qe, ae, tb, W=Wb, bias=bb, ws=ws)
    us2 = _graph_time(torch, embed_then_bilinear)
    us1 = _graph_time(torch, lambda: capi.embed_simcross_bilinear_forward(iq, ia, tab, Wb, bb, tb))
    out["cfg4_scoring_bilinear_M4_from_word_ids_1517x40x40x50"] = {
        "us_per_step_embed_then_simcross": us2, "us_per_step_fused": us1, "pairs_per_s": n / (us1 * 1e-6),
        "note": "mms_embed_simcross_bilinear_forward_f32: the word-grid forward gathers its q / a images from the "
                "table itself, one launch"}
    del tab, iq, ia, qe, ae, tg, Wb, bb, tb
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
    # the same call with its results left on the device (what a Layer's Forward_gpu does), graph-replayed
    res, eff = torch.empty(2, device="cuda"), torch.empty(1, dtype=torch.int32, device="cuda")
    try:
        usd = _graph_time(torch, lambda: capi.rank_map_mrr_device(prob, lab, grp, res, eff, ws=ws))
        out["cfg4_map_mrr_1517_candidates"]["us_per_call_device_results_graph"] = usd
    except Exception as ex:                               # a sort that cannot be captured is reported, not hidden
        out["cfg4_map_mrr_1517_candidates"]["device_results_graph_error"] = str(ex)[:200]
    torch.cuda.empty_cache()
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` typed as is (no torchrun): this process -- which has NOT imported torch and never
    touches the GPU -- starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's
    stdout (the one JSON line) and exits with the worst child's return code.  What it replaces: the reference starts
    its per-GPU workers from one command too (`caffe train --gpu all`, tools/caffe.cpp:154-227 ->
    P2PSync::Run, parallel.cpp:421-430).  Children are plain subprocesses (no exec of a process that has
    initialised the GPU); the torchrun form of the contract keeps working because it sets WORLD_SIZE itself."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   MMS_BENCH_CHILD="1")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    worst = 0
    for rc in rcs:
        if rc != 0:
            worst = rc if rc > 0 else 128 - rc        # a child killed by a signal reports as the shell does
            break
    return worst


if __name__ == "__main__":
    _args = parse()
    if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(_args, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # native libraries chat on fd 1 ("[Gloo] Rank 0 is connected to ..."): the contract is ONE JSON line on
        # stdout, so fd 1 is pointed at stderr for the rest of the process and print() keeps the real stdout
        _real = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(_real, "w")
    run(_args)
