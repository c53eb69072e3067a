"""tools/f32_bigD_probe.py -- dev probe: the fp32 Euclid sentence-vector path at large D, fused fwd+bwd, hipGraph-replayed (round 3:
8192 x 1024: 23.6 us = 0.71 of 8 TB/s on the 134 MB moved -- memory-bound already; the lane walk was built for the fp16 form only)."""
import sys, os
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
for (N, D) in ((8192, 1024), (8192, 768), (4096, 512)):
    q = torch.randn(N, 1, D, device="cuda") * 0.4; a = torch.randn(N, 1, D, device="cuda") * 0.4
    dT = torch.randn(N, 1, 1, 1, device="cuda"); top = torch.empty(N, 1, 1, 1, device="cuda")
    dq, da = torch.empty_like(q), torch.empty_like(a)
    us = gtime(lambda: capi.simcross_forward_backward(1, q, a, dT, top, dq, da))
    b = N * (4 * D * 4 + 8)
    print("fp32 Euclid fused fwd+bwd %d x %d: %.2f us  (%.2f of 8 TB/s on %.1f MB moved)" % (N, D, us, b / us / 1e3 / 8000, b / 1e6))
