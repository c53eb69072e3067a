"""tools/rank_small_probe.py -- dev-only: MAP + MRR and AUC on TREC-QA-sized inputs, device results, graph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
g = torch.Generator(device="cuda").manual_seed(1)
for n in (50, 256, 512, 513, 1517):
    sc = torch.rand(n, device="cuda", generator=g)
    prob = torch.stack([1 - sc, sc], 1).contiguous()
    lab = (torch.rand(n, device="cuda", generator=g) < 0.2).float()
    grp = torch.sort(torch.randint(0, 68, (n,), device="cuda", generator=g).float()).values
    res, eff = torch.empty(2, device="cuda"), torch.empty(1, dtype=torch.int32, device="cuda")
    fn = lambda: capi.rank_map_mrr_device(prob, lab, grp, res, eff)
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
    print("MAP+MRR n=%d: %.2f us per call (device results)" % (n, e0.elapsed_time(e1) * 1e3 / 80))
