"""tools/soak_f16_lanewalk.py -- dev-only: the fp16-storage ordered distance (one lane walks a pair's image) against a numpy
restatement of the reference's loop (d-ascending fp32 running sum, sim_cross_layer.cpp:100-111: np.cumsum on float32 adds in
order), bit for bit, on random batch sizes and widths D in (400, 2048], D % 8 == 0; Gaussian, heavy-tailed, tiny and
near-identical pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mms_answer_selection_amd import capi

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 150
r = np.random.default_rng(77)
pairs = mism = 0
for it in range(runs):
    N = int(r.integers(1, 400))
    D = 8 * int(r.integers(51, 257))
    kind = it % 4
    q = r.standard_normal((N, D)); a = r.standard_normal((N, D))
    if kind == 1: q, a = r.standard_t(2.2, (N, D)), r.standard_t(2.2, (N, D))
    if kind == 2: q, a = q * 1e-3, a * 1e-3
    if kind == 3: a = q + 1e-2 * r.standard_normal((N, D))
    qh, ah = (0.4 * q).astype(np.float16), (0.4 * a).astype(np.float16)
    d = qh.astype(np.float32) - ah.astype(np.float32)
    sq = d * d
    dist = np.cumsum(sq, axis=1, dtype=np.float32)[:, -1]
    ref = (np.float32(1) / (np.float32(1) + np.sqrt(dist))).astype(np.float32)
    top = torch.empty(N, 1, 1, 1, device="cuda")
    capi.simcross_euclid_forward_f16(torch.from_numpy(qh).cuda().view(N, 1, D), torch.from_numpy(ah).cuda().view(N, 1, D), top)
    got = top.cpu().numpy().ravel()
    mism += int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    pairs += N
print("soak_f16_lanewalk: %d runs, %d pairs, D in (400, 2048]: %d mismatching score words" % (runs, pairs, mism))
sys.exit(1 if mism else 0)
