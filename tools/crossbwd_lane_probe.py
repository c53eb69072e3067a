"""tools/crossbwd_lane_probe.py -- dev-only: Euclid cross-geometry backward at cfg 4's shape (1517 x 40 x 40 x 50),
eager launches timed with events, both backward arithmetic modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
for (N, W1, W2, D) in ((1517, 40, 40, 50), (190, 40, 40, 50), (4096, 20, 20, 50)):
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(N, W1, D, device="cuda", generator=g) * 0.4
    a = torch.randn(N, W2, D, device="cuda", generator=g) * 0.4
    dT = torch.randn(N, 1, W1, W2, device="cuda", generator=g)
    top = torch.empty(N, 1, W1, W2, device="cuda")
    dq, da = torch.empty_like(q), torch.empty_like(a)
    capi.simcross_forward(1, q, a, top)
    for mode in ("fp32", "reference"):
        capi.set_euclid_backward_mode(mode)
        for _ in range(3):
            capi.simcross_backward(1, q, a, top, dT, dq, da)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            capi.simcross_backward(1, q, a, top, dT, dq, da)
        e1.record(); torch.cuda.synchronize()
        print("%s backward %-10s %8.2f us" % ((N, W1, W2, D), mode, e0.elapsed_time(e1) * 1e3 / 50))
