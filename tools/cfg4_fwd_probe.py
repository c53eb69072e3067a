"""tools/cfg4_fwd_probe.py -- dev-only: cfg 4's scoring forward (1517 x 40 x 40 x 50 Euclid, and the cosine mode) and
neighbouring geometries, graph-replayed (32 calls per graph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
shapes = [(1517, 40, 40, 50), (4096, 40, 40, 50), (1517, 16, 24, 50), (1517, 8, 40, 50)]
g = torch.Generator(device="cuda").manual_seed(3)
for (N, W1, W2, D) in shapes:
    q = torch.randn(N, W1, D, device="cuda", generator=g) * 0.4
    a = torch.randn(N, W2, D, device="cuda", generator=g) * 0.4
    top = torch.empty(N, 1, W1, W2, device="cuda")
    n0, n1 = torch.empty(N, W1, device="cuda"), torch.empty(N, W2, device="cuda")
    for mode in (1, 0):
        step = (lambda: capi.simcross_forward(1, q, a, top)) if mode == 1 else (lambda: capi.simcross_forward(0, q, a, top, norm0=n0, norm1=n1))
        for _ in range(3): step()
        torch.cuda.synchronize()
        cap = torch.cuda.Stream(); cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=cap):
                for _ in range(32): step()
        torch.cuda.current_stream().wait_stream(cap)
        for _ in range(3): gph.replay()
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): gph.replay()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 128)
        ts.sort()
        print("%-22s mode %d: %.2f us" % ((N, W1, W2, D), mode, ts[2]))
