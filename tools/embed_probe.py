"""dev-only: Embed forward / backward timing at TREC-QA-like batch shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mms_answer_selection_amd import capi
r = np.random.default_rng(0)
for M, N, K, padfrac in ((50 * 80, 50, 3000, 0.6), (50 * 80, 300, 20000, 0.6), (1517 * 80, 50, 20000, 0.6), (4096 * 40, 300, 50000, 0.3)):
    idx = r.integers(0, K, M); idx[r.uniform(size=M) < padfrac] = K - 1
    index = torch.from_numpy(idx.astype(np.float32)).cuda()
    w = torch.randn(K, N, device="cuda"); top = torch.empty(M, N, device="cuda"); dT = torch.randn(M, N, device="cuda"); wd = torch.zeros(K, N, device="cuda")
    for _ in range(2):
        capi.embed_forward(index, w, top); capi.embed_backward(index, dT, wd)
    torch.cuda.synchronize()
    res = []
    for fn in (lambda: capi.embed_forward(index, w, top), lambda: capi.embed_backward(index, dT, wd)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100)
    print("M %d N %d K %d pad %.1f: fwd %.1f us  bwd %.1f us" % (M, N, K, padfrac, res[0], res[1]))
