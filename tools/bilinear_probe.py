"""dev-only: SimCross bilinear (mode 2) at the driver's geometry, a few iterations (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, W, D, M = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 40, 300, 4)))
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(N, W, D, device="cuda", generator=g) * 0.4
a = torch.randn(N, W, D, device="cuda", generator=g) * 0.4
Wt = torch.randn(M, D, D, device="cuda", generator=g) * 0.05
bias = torch.zeros(M, W, W, device="cuda")
top = torch.empty(N, M, W, W, device="cuda"); dT = torch.randn(N, M, W, W, device="cuda", generator=g)
dq = torch.empty_like(q); da = torch.empty_like(a); dW = torch.empty_like(Wt); db = torch.zeros_like(bias)
def fwd(): capi.simcross_forward(2, q, a, top, W=Wt, bias=bias)
def bwd(): capi.simcross_backward(2, q, a, top, dT, dq, da, W=Wt, bias_term=True, dW=dW, dbias=db)
for _ in range(3): fwd(); bwd()
torch.cuda.synchronize()
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "us", e0.elapsed_time(e1) * 50)
