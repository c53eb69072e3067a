"""tools/soak_bx3.py -- dev-only: the SimMatrix products on the bf16 pipe (csrc/bx3_gemm.h) against fp64 on random shapes
and value distributions: N in [2048, 6000], K1, K2 multiples of 4 in [4, 320], Gaussian / heavy-tailed / mixed-scale /
sparse inputs.  Forward (scores, Q.W), cached and recomputing backward (dq, da, dW).  Prints the worst scaled error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mms_answer_selection_amd import capi

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
r = np.random.default_rng(2025)
worst = {}
def upd(name, got, ref):
    ref = ref.double()
    scale = max(1.0, ref.abs().max().item())
    e = (got.double() - ref).abs().max().item() / scale
    assert np.isfinite(e), name
    worst[name] = max(worst.get(name, 0.0), e)
    return e
ws = capi.Workspace()
for it in range(runs):
    N = int(r.integers(2048, 6001))
    K1, K2 = 4 * int(r.integers(1, 81)), 4 * int(r.integers(2, 81))
    kind = it % 4
    def draw(*s):
        x = r.standard_normal(s)
        if kind == 1: x = r.standard_t(2.5, s) * 0.3
        if kind == 2: x = x * np.exp(r.uniform(-6, 2, (s[0], 1)))          # rows of very different scale
        if kind == 3: x = x * (r.uniform(size=s) < 0.3)                     # sparse
        return torch.from_numpy((x * 0.4).astype(np.float32)).cuda()
    q, a = draw(N, K1), draw(N, K2)
    W = torch.from_numpy(r.uniform(-0.1, 0.1, (K1, K2)).astype(np.float32)).cuda()
    dT = torch.from_numpy(r.standard_normal((N, 1)).astype(np.float32)).cuda()
    top, qw = torch.empty(N, 1, device="cuda"), torch.empty(N, K2, device="cuda")
    dq, da, dW = torch.empty_like(q), torch.empty_like(a), torch.zeros_like(W)
    capi.simmatrix_forward(q, a, W, top, qw, ws=ws)
    P = q.double() @ W.double()
    upd("Q.W", qw, P)
    upd("top", top, (P * a.double()).sum(1, keepdim=True))
    capi.simmatrix_backward(q, a, W, dT, dq, da, dW, ws=ws, qw=qw)
    upd("dq", dq, dT.double() * (a.double() @ W.double().T))
    upd("da (cached)", da, dT.double() * P)
    upd("dW", dW, q.double().T @ (dT.double() * a.double()))
    if K1 % 8 == 0:                                             # the fp16-storage scoring entry point on the rounded inputs
        qh, ah = q.half(), a.half()
        t16 = torch.empty(N, 1, device="cuda")
        capi.simmatrix_forward_f16(qh, ah, W, t16, ws=ws)
        upd("top (fp16 storage)", t16, ((qh.double() @ W.double()) * ah.double()).sum(1, keepdim=True))
    da2, dq2, dW2 = torch.empty_like(a), torch.empty_like(q), torch.zeros_like(W)
    capi.simmatrix_backward(q, a, W, dT, dq2, da2, dW2, ws=ws)
    assert torch.equal(da2, da) and torch.equal(dq2, dq) and torch.equal(dW2, dW), "cached and recomputing backward differ"
print("soak_bx3: %d random shapes x 4 value distributions, worst |err| / max(1, max|ref|):" % runs)
for k, v in worst.items():
    print("   %-20s %.3e %s" % (k, v, "OK" if v <= 1e-5 else "** over 1e-5 **"))
