"""tools/triplet_probe.py -- dev-only: the fused (q, a+, a-) step, HBM-cold ring, hipGraph-replayed (us per step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
mode = sys.argv[1] if len(sys.argv) > 1 else "launch"
capi.set_triplet_finish_mode(mode)
N, D, ring, G = 4096, 300, 48, 16
g = torch.Generator(device="cuda").manual_seed(1)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g) * 0.4
q, ap, an = mk(ring, N, 1, D), mk(ring, N, 1, D), mk(ring, N, 1, D)
y = (torch.rand(ring, N, 1, device="cuda", generator=g) < 0.8).float()
dq, dp, dn = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
sp, sn, loss = torch.empty(ring, N, 1, device="cuda"), torch.empty(ring, N, 1, device="cuda"), torch.empty(ring, 1, device="cuda")
ws = capi.TripletWorkspace()
step = lambda i: capi.triplet_euclid_step(q[i], ap[i], an[i], y[i], sp[i], sn[i], loss[i], dq[i], dp[i], dn[i], margin=0.05, ws=ws)
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
for r in range(6): graphs[r % 3].replay()
torch.cuda.synchronize()
ts = []
for rep in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(12): graphs[r % 3].replay()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / (12 * G))
ts.sort()
print("triplet step %s: median %.2f us  min %.2f   loss %r" % (mode, ts[len(ts) // 2], ts[0], loss[:3, 0].tolist()))
