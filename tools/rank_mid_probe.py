"""tools/rank_mid_probe.py -- dev probe: phases of rank_mid_kernel (build with MMS_HIPCC_EXTRA=-DMMS_RANK_STAMPS)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mms_answer_selection_amd import capi
g = torch.Generator(device="cuda").manual_seed(1701)
for n in (600, 1517, 2048):
    sc = torch.rand(n, device="cuda", generator=g)
    prob = torch.stack([1 - sc, sc], 1).contiguous()
    lab = (torch.rand(n, device="cuda", generator=g) < 0.2).float()
    grp = torch.sort(torch.randint(0, 68, (n,), device="cuda", generator=g).float()).values
    res, eff = torch.zeros(16, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = capi.Workspace()
    for _ in range(3):
        capi.rank_map_mrr_device(prob, lab, grp, res, eff, ws=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        capi.rank_map_mrr_device(prob, lab, grp, res, eff, ws=ws)
    e1.record(); torch.cuda.synchronize()
    r = res.tolist()
    print("n=%d  %.1f us per call eager; phases (us): keys %.2f  sort %.2f  walks %.2f  fold %.2f" %
          (n, e0.elapsed_time(e1) * 10, r[4] / 100, r[5] / 100, r[6] / 100, r[7] / 100))
