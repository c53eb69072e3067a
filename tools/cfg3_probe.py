"""tools/cfg3_probe.py -- dev-only: cfg 3 SimMatrix 16384 x 300 x 300 forward + cached backward, graph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, K = 16384, 300
g = torch.Generator(device="cuda").manual_seed(1)
q, a = torch.randn(N, K, device="cuda", generator=g) * 0.4, torch.randn(N, K, device="cuda", generator=g) * 0.4
W = torch.randn(K, K, device="cuda", generator=g) * 0.05
dT = torch.randn(N, 1, device="cuda", generator=g)
top = torch.empty(N, 1, device="cuda"); qw = torch.empty(N, K, device="cuda")
dq, da, dW = torch.empty_like(q), torch.empty_like(a), torch.zeros_like(W)
ws = capi.Workspace()
def step():
    capi.simmatrix_forward(q, a, W, top, qw)
    capi.simmatrix_backward(q, a, W, dT, dq, da, dW, qw=qw, ws=ws)
step(); torch.cuda.synchronize()
cap = torch.cuda.Stream(); cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=cap):
        for _ in range(32): step()
torch.cuda.current_stream().wait_stream(cap)
gph.replay(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 32)
ts.sort()
print("cfg3 fwd+bwd (cached): median %.2f us  min %.2f" % (ts[2], ts[0]))
