"""tools/embed_bilinear_probe.py -- dev-only: cfg 4 scoring in network_v4's mode from word ids: Embed x2 + SimCross
(dist_mode 2, M = 4, bias) as three launches vs the fused call, graph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, W1, W2, D, M, K = 1517, 40, 40, 50, 4, 100000
g = torch.Generator(device="cuda").manual_seed(1)
table = torch.randn(K, D, device="cuda", generator=g) * 0.4
iq = torch.randint(0, K, (N, W1), device="cuda", generator=g).float()
ia = torch.randint(0, K, (N, W2), device="cuda", generator=g).float()
Wm = torch.randn(M, D, D, device="cuda", generator=g) * 0.1
bias = torch.randn(M, W1, W2, device="cuda", generator=g)
q, a = torch.empty(N, W1, D, device="cuda"), torch.empty(N, W2, D, device="cuda")
top = torch.empty(N, M, W1, W2, device="cuda")
def three():
    capi.embed_forward(iq.view(-1), table, q.view(N * W1, D))
    capi.embed_forward(ia.view(-1), table, a.view(N * W2, D))
    capi.simcross_forward(2, q, a, top, W=Wm, bias=bias)
def fused():
    capi.embed_simcross_bilinear_forward(iq, ia, table, Wm, bias, top)
for name, fn in (("Embed, Embed, SimCross", three), ("fused", fused)):
    fn(); torch.cuda.synchronize()
    cap = torch.cuda.Stream(); cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=cap):
            for _ in range(8): fn()
    torch.cuda.current_stream().wait_stream(cap)
    gph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gph.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-26s %8.2f us per scoring pass" % (name, e0.elapsed_time(e1) * 1e3 / 80))
