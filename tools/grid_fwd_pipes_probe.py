"""cfg 4 in the reference net's scoring mode (SimCross bilinear M = 4 + bias, 1517 x 40 x 40 x 50 forward) on both matrix
pipes, hipGraph-replayed, plus the word-id form (Embed gather fused into the loads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi

def gtime(fn, iters=16, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters): fn()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / iters)
    return sorted(ts)[len(ts) // 2]

g = torch.Generator(device="cuda").manual_seed(5)
for (N, Wd, D, M) in ((1517, 40, 50, 4), (1517, 40, 64, 4), (2048, 33, 50, 2)):
    q = torch.randn(N, Wd, D, device="cuda", generator=g) * 0.4
    a = torch.randn(N, Wd, D, device="cuda", generator=g) * 0.4
    W = torch.rand(M, D, D, device="cuda", generator=g) * 0.16 - 0.08
    b = torch.randn(M, Wd, Wd, device="cuda", generator=g)
    top = torch.empty(N, M, Wd, Wd, device="cuda")
    ws = capi.Workspace()
    out = {}
    for mode in ("bf16x3", "fp32"):
        capi.set_matrix_mode(mode)
        fn = lambda: capi.simcross_forward(2, q, a, top, W=W, bias=b, ws=ws)
        us = gtime(fn)
        out[mode] = top.clone()
        print("%d x %d x %d x %d M=%d  %-7s %.2f us" % (N, Wd, Wd, D, M, mode, us), flush=True)
    capi.set_matrix_mode("bf16x3")
    ref = torch.einsum("njd,mde,nke->nmjk", q.double(), W.double(), a.double()) + b.double()[None]
    for mode in out:
        print("   %-7s max |err| vs fp64 %.3e (max |ref| %.2f)" % (mode, (out[mode].double() - ref).abs().max().item(), ref.abs().max().item()))
