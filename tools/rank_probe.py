"""dev-only: ranking metric timings (MAP+MRR, AUC, RankAccuracy) at TREC-QA and larger sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mms_answer_selection_amd import capi
r = np.random.default_rng(0)
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for n, groups in ((1517, 68), (20000, 900), (200000, 9000)):
    g = np.sort(r.integers(0, groups, n)).astype(np.float32)
    lab = (r.uniform(size=n) < 0.2).astype(np.float32)
    s = r.uniform(size=n).astype(np.float32)
    prob = torch.from_numpy(np.stack([1 - s, s], 1)).cuda(); L = torch.from_numpy(lab).cuda(); G = torch.from_numpy(g).cuda()
    a = torch.from_numpy(s).cuda(); b = torch.from_numpy(r.uniform(size=n).astype(np.float32)).cuda()
    print("n %d: map+mrr %.1f us (incl. D2H of 3 scalars)  auc %.1f us  rank_accuracy %.1f us" % (
        n, t(lambda: capi.rank_map_mrr(prob, L, G)), t(lambda: capi.rank_auc(prob, L)), t(lambda: capi.rank_accuracy(a, b, L))))
