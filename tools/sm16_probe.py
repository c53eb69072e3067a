"""tools/sm16_probe.py -- dev probe: fp16-storage SimMatrix scoring (mms_simmatrix_forward_f16) next to the fp32 forward,
16384 x 300(304) x 300, hipGraph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi

def gtime(fn, iters=16, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters): fn()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / iters)
    return sorted(ts)[len(ts) // 2]

N, K1, K2 = 16384, 304, 300
g = torch.Generator(device="cuda").manual_seed(3)
q = torch.randn(N, K1, device="cuda", generator=g) * 0.4
a = torch.randn(N, K2, device="cuda", generator=g) * 0.4
W = torch.rand(K1, K2, device="cuda", generator=g) * 0.16 - 0.08
qh, ah = q.half(), a.half()
top, scr = torch.empty(N, 1, device="cuda"), torch.empty(N, K2, device="cuda")
ws = capi.Workspace()
print("fp32 storage, forward (scores + Q.W):  %.2f us" % gtime(lambda: capi.simmatrix_forward(q, a, W, top, scr, ws=ws)))
print("fp16 storage, scoring (scores only):   %.2f us" % gtime(lambda: capi.simmatrix_forward_f16(qh, ah, W, top, ws=ws)))
K2p = 304                                              # the training pair wants K2 % 8 == 0 too
a2 = torch.randn(N, K2p, device="cuda", generator=g) * 0.4
W2 = torch.rand(K1, K2p, device="cuda", generator=g) * 0.16 - 0.08
a2h = a2.half()
dT = torch.randn(N, 1, device="cuda", generator=g)
qw2 = torch.empty(N, K2p, device="cuda")
dqh, dah = torch.empty(N, K1, device="cuda", dtype=torch.float16), torch.empty(N, K2p, device="cuda", dtype=torch.float16)
dW2 = torch.zeros_like(W2)
dq32, da32 = torch.empty(N, K1, device="cuda"), torch.empty(N, K2p, device="cuda")
def step16():
    capi.simmatrix_forward_train_f16(qh, a2h, W2, top, qw2, ws=ws)
    capi.simmatrix_backward_f16(qh, a2h, W2, qw2, dT, dqh, dah, dW2, ws=ws)
def step32():
    capi.simmatrix_forward(q, a2, W2, top, qw2, ws=ws)
    capi.simmatrix_backward(q, a2, W2, dT, dq32, da32, dW2, ws=ws, qw=qw2)
print("fp16 storage, training step (fwd + bwd), 16384 x 304 x 304:  %.2f us" % gtime(step16))
print("fp32 storage, the same step on the bf16 pipe:                 %.2f us" % gtime(step32))
ref = ((qh.double() @ W.double()) * ah.double()).sum(1, keepdim=True)
capi.simmatrix_forward_f16(qh, ah, W, top, ws=ws)
print("max |err| vs fp64 on the rounded inputs: %.3e (max |ref| %.2f)" % ((top.double() - ref).abs().max().item(), ref.abs().max().item()))
