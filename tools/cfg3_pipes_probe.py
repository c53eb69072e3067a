"""cfg 3 (SimMatrix 16384 x 300 x 300) forward + cached backward under a graph, on both matrix pipes, with the parts
timed alone.  python tools/cfg3_pipes_probe.py"""
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

N, K = 16384, 300
g = torch.Generator(device="cuda").manual_seed(1701)
q, a = torch.randn(N, K, device="cuda", generator=g) * 0.4, torch.randn(N, K, device="cuda", generator=g) * 0.4
W = torch.rand(K, K, device="cuda", generator=g) * 0.16 - 0.08
dT = torch.randn(N, 1, device="cuda", generator=g)
top, scr = torch.empty(N, 1, device="cuda"), torch.empty(N, K, device="cuda")
dq, da, dW = torch.empty_like(q), torch.empty_like(a), torch.zeros_like(W)
ws = capi.Workspace()
an = torch.randn(N, K, device="cuda", generator=g) * 0.4
yl = (torch.rand(N, 1, device="cuda", generator=g) < 0.8).float()
sp, sn, ls = torch.empty(N, 1, device="cuda"), torch.empty(N, 1, device="cuda"), torch.empty(1, device="cuda")
dan = torch.empty_like(a)
ws2 = capi.Workspace()
for mode in ("bf16x3", "fp32", "bf16x3"):
    capi.set_matrix_mode(mode)
    fwd = lambda: capi.simmatrix_forward(q, a, W, top, scr, ws=ws)
    bwd = lambda: capi.simmatrix_backward(q, a, W, dT, dq, da, dW, ws=ws, qw=scr)
    dwo = lambda: capi.simmatrix_backward(q, a, W, dT, None, None, dW, ws=ws, propagate_down=(False, False))
    dqo = lambda: capi.simmatrix_backward(q, a, W, dT, dq, da, None, ws=ws, qw=scr, param_propagate_down=False)
    both = lambda: (fwd(), bwd())
    trip = lambda: capi.triplet_simmatrix_step(q, a, an, yl, W, sp, sn, ls, dq, da, dan, dW, margin=0.3, ws=ws2)
    print("%-7s fwd %.2f  bwd %.2f (dW only %.2f, dq+da only %.2f)  step %.2f us   fused triplet step %.2f us" % (
        mode, gtime(fwd), gtime(bwd), gtime(dwo), gtime(dqo), gtime(both), gtime(trip)), flush=True)
capi.set_matrix_mode("bf16x3")
