"""tools/gemm_probe.py -- dev-only: SimMatrix cfg 3 forward / backward a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, K = 16384, 300
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(N, K, device="cuda", generator=g) * 0.4
a = torch.randn(N, K, device="cuda", generator=g) * 0.4
W = torch.randn(K, K, device="cuda", generator=g) * 0.05
top = torch.empty(N, 1, device="cuda"); scratch = torch.empty(N, K, device="cuda")
dT = torch.randn(N, 1, device="cuda", generator=g)
dq = torch.empty_like(q); da = torch.empty_like(a); dW = torch.zeros_like(W)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    capi.simmatrix_forward(q, a, W, top, scratch)
    capi.simmatrix_backward(q, a, W, dT, dq, da, dW, qw=scratch if os.environ.get('MMS_PROBE_CACHED', '1') == '1' else None)
torch.cuda.synchronize()
