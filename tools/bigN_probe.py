"""tools/bigN_probe.py -- dev probe: SimMatrix forward + backward on the bf16 pipe at 131,072 pairs x 300 x 300 against fp64
(round 3: Q.W 7.5e-07, top 3.1e-07, dq 5.0e-07, da 5.1e-07, dW 9.2e-07 scaled error)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, K = 131072, 300
g = torch.Generator(device="cuda").manual_seed(9)
q = torch.randn(N, K, device="cuda", generator=g) * 0.4
a = torch.randn(N, K, device="cuda", generator=g) * 0.4
W = torch.rand(K, K, device="cuda", generator=g) * 0.16 - 0.08
dT = torch.randn(N, 1, device="cuda", generator=g)
top, qw = torch.empty(N, 1, device="cuda"), torch.empty(N, K, device="cuda")
dq, da, dW = torch.empty_like(q), torch.empty_like(a), torch.zeros_like(W)
ws = capi.Workspace()
capi.simmatrix_forward(q, a, W, top, qw, ws=ws)
capi.simmatrix_backward(q, a, W, dT, dq, da, dW, ws=ws, qw=qw)
torch.cuda.synchronize()
P = q.double() @ W.double()
def e(x, r): return ((x.double() - r).abs().max() / max(1.0, r.abs().max().item())).item()
print("N=%d: Q.W %.2e top %.2e dq %.2e da %.2e dW %.2e" % (N, e(qw, P), e(top, (P * a.double()).sum(1, keepdim=True)),
      e(dq, dT.double() * (a.double() @ W.double().T)), e(da, dT.double() * P), e(dW, q.double().T @ (dT.double() * a.double()))))
