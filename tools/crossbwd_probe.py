"""tools/crossbwd_probe.py -- dev-only: Euclid word-grid backward (cfg 4's shape), default fp32 arithmetic,
hipGraph-replayed; checks the result against the reference-mode kernel at 2 ulp."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, W, D = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1517, 40, 50)))
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(N, W, D, device="cuda", generator=g) * 0.4
a = torch.randn(N, W, D, device="cuda", generator=g) * 0.4
dT = torch.randn(N, 1, W, W, device="cuda", generator=g)
top = torch.empty(N, 1, W, W, device="cuda")
capi.simcross_forward(1, q, a, top)
dq, da = torch.empty_like(q), torch.empty_like(a)
capi.set_euclid_backward_mode("reference")
capi.simcross_backward(1, q, a, top, dT, dq, da)
rq, ra = dq.clone(), da.clone()
capi.set_euclid_backward_mode("fp32")
capi.simcross_backward(1, q, a, top, dT, dq, da)
torch.cuda.synchronize()
u = lambda x, y: (x.view(torch.int32).long() - y.view(torch.int32).long()).abs().max().item()
rel = lambda x, y: ((x - y).abs().max() / y.abs().max()).item()
print("fp32 vs reference arithmetic: max rel diff dq %.2e da %.2e" % (rel(dq, rq), rel(da, ra)))
fn = lambda: capi.simcross_backward(1, q, a, top, dT, dq, da)
cap = torch.cuda.Stream(); cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph, stream=cap):
        for _ in range(8): fn()
torch.cuda.current_stream().wait_stream(cap)
gph.replay(); torch.cuda.synchronize()
ts = []
for rep in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gph.replay(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 8)
ts.sort()
print("Euclid cross backward %dx%dx%dx%d fp32 arithmetic: median %.2f us  min %.2f" % (N, W, W, D, ts[3], ts[0]))
