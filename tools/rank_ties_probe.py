"""tools/rank_ties_probe.py -- dev-only: MAP + MRR and AUC on heavily tied scores, default tie rule vs MMS_RANK_TIES_LIBSTDCXX (us per call incl. the D2H of the results)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mms_answer_selection_amd import capi
r = np.random.default_rng(0)
for n, levels in ((1517, 4), (1517, 1), (20000, 8), (200000, 16)):
    score = (np.round(r.uniform(0, 1, n) * levels) / max(levels, 1)).astype(np.float32)
    label = (r.uniform(size=n) < 0.3).astype(np.float32)
    group = r.integers(0, max(1, n // 22), n).astype(np.float32)
    prob = torch.from_numpy(np.stack([1 - score, score], 1).astype(np.float32)).cuda()
    lab, grp = torch.from_numpy(label).cuda(), torch.from_numpy(group).cuda()
    for mode in ("input", "libstdcxx"):
        capi.set_rank_tie_mode(mode)
        res = []
        for fn in (lambda: capi.rank_map_mrr(prob, lab, grp), lambda: capi.rank_auc(prob, lab)):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 200)
        print("n %6d levels %2d mode %-9s: map+mrr %9.1f us   auc %9.1f us" % (n, levels, mode, res[0], res[1]))
capi.set_rank_tie_mode("input")
