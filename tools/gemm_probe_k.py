"""dev-only: SimMatrix forward GEMM at K1 = 300, 600, 1200 (slope = main loop, intercept = fixed cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, K2 = 16384, 300
g = torch.Generator(device="cuda").manual_seed(1)
for K1 in (300, 600, 1200):
    q = torch.randn(N, K1, device="cuda", generator=g) * 0.4
    a = torch.randn(N, K2, device="cuda", generator=g) * 0.4
    W = torch.randn(K1, K2, device="cuda", generator=g) * 0.05
    top = torch.empty(N, 1, device="cuda"); scratch = torch.empty(N, K2, device="cuda")
    for _ in range(3):
        capi.simmatrix_forward(q, a, W, top, scratch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        capi.simmatrix_forward(q, a, W, top, scratch)
    e1.record(); torch.cuda.synchronize()
    print("K1", K1, "fwd us", e0.elapsed_time(e1) * 1e3 / 50)
