"""tools/rank_graph_probe.py -- dev probe: MAP + MRR over n candidates with the results left on the device, hipGraph-replayed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mms_answer_selection_amd import capi
g = torch.Generator(device="cuda").manual_seed(1701)
for n in (600, 1517, 2048):
    sc = torch.rand(n, device="cuda", generator=g)
    prob = torch.stack([1 - sc, sc], 1).contiguous()
    lab = (torch.rand(n, device="cuda", generator=g) < 0.2).float()
    grp = torch.randint(0, 68, (n,), device="cuda", generator=g).float()
    res, eff = torch.zeros(16, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = capi.Workspace()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): capi.rank_map_mrr_device(prob, lab, grp, res, eff, ws=ws)
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(32): capi.rank_map_mrr_device(prob, lab, grp, res, eff, ws=ws)
        gph.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); gph.replay(); e1.record(s); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / 32)
    print("n=%d  MAP+MRR on the device: %.2f us per call (graph of 32)" % (n, sorted(ts)[2]))
