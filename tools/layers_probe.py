"""tools/layers_probe.py -- dev-only: the Layer-API sequence at cfg 2 (Forward launch, Backward launch), forward-only
and backward-only, HBM-cold ring, hipGraph-replayed; env MMS_PAIR32_WPB_{FWD,BWD} select waves per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, D, ring, G = 4096, 300, 64, 32
g = torch.Generator(device="cuda").manual_seed(1)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.4
q, a = mk(ring, N, 1, D), mk(ring, N, 1, D)
dT = mk(ring, N, 1, 1, 1)
top = torch.empty(ring, N, 1, 1, 1, device="cuda")
dq, da = torch.empty_like(q), torch.empty_like(a)
def timeit(name, step):
    for i in range(ring): step(i)
    torch.cuda.synchronize()
    graphs = []
    cap = torch.cuda.Stream(); cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for g0 in range(0, ring, G):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=cap):
                for i in range(g0, g0 + G): step(i)
            graphs.append(gph)
    torch.cuda.current_stream().wait_stream(cap)
    for r in range(4): graphs[r % 2].replay()
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(8): graphs[r % 2].replay()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (8 * G))
    ts.sort()
    print("%-22s wpb fwd=%s bwd=%s: median %.3f us  min %.3f" % (name, os.environ.get("MMS_PAIR32_WPB_FWD", "8"),
          os.environ.get("MMS_PAIR32_WPB_BWD", "8"), ts[3], ts[0]))
fwd = lambda i: capi.simcross_forward(1, q[i], a[i], top[i])
bwd = lambda i: capi.simcross_backward(1, q[i], a[i], top[i], dT[i], dq[i], da[i])
timeit("forward only", fwd)
timeit("backward only", bwd)
timeit("forward + backward", lambda i: (fwd(i), bwd(i)))
