"""Times the fp16-storage cosine rows kernel at cfg 5's shard (8192 x 1024) under a graph; bytes per pair 4 D + 16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi

N, D = 8192, 1024
q = torch.randn(N, 1, D, device="cuda").half(); a = torch.randn(N, 1, D, device="cuda").half()
dT = torch.randn(N, 1, 1, 1, device="cuda"); top = torch.empty(N, 1, 1, 1, device="cuda")
dq = torch.empty_like(q); da = torch.empty_like(a)
n0 = torch.empty(N, 1, device="cuda"); n1 = torch.empty(N, 1, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        capi.simcross_cosine_forward_backward_f16(q, a, dT, top, dq, da, n0, n1)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20):
            capi.simcross_cosine_forward_backward_f16(q, a, dT, top, dq, da, n0, n1)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); 
    for _ in range(10): g.replay()
    e1.record(s); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 200
b = N * (4 * D * 2 + 16)
print("cosine f16 fwd+bwd %d x %d: %.2f us/launch, %.0f GB/s algorithmic (%.2f of 8 TB/s)" % (N, D, us, b / us / 1e3, b / us / 1e3 / 8000))
