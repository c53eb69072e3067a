"""dev-only soak: the specialised pair kernel against the oracle on many random batches and input
distributions (scores and reference-mode gradients bit for bit; counts any mismatch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mms_answer_selection_amd import capi
from oracle import cpu_oracle as O
capi.set_euclid_backward_mode("reference")
bad = 0; total = 0; t0 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    r = np.random.default_rng(seed)
    N = int(r.choice([4096, 4095, 1234, 8192]))
    D = int(r.choice([300, 300, 200, 100]))
    kind = seed % 5
    if kind == 0: q = r.standard_normal((N, 1, D)) * 0.4; a = r.standard_normal((N, 1, D)) * 0.4
    elif kind == 1: q = r.uniform(-1, 1, (N, 1, D)); a = r.uniform(-1, 1, (N, 1, D))
    elif kind == 2: q = r.standard_t(2, (N, 1, D)); a = r.standard_t(2, (N, 1, D))            # heavy tails
    elif kind == 3: q = r.standard_normal((N, 1, D)) * 1e-3; a = q + r.standard_normal((N, 1, D)) * 1e-5   # near-identical
    else: q = np.abs(r.standard_normal((N, 1, D))) * r.choice([1e-6, 1.0, 30.0], (N, 1, 1)); a = -q * r.uniform(0, 1, (N, 1, 1))
    q = q.astype(np.float32); a = a.astype(np.float32)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    top_ref, _, _ = O.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = O.simcross_backward(1, q, a, top_ref, dT)
    qd, ad, dTd = (torch.from_numpy(x).cuda() for x in (q, a, dT))
    top = torch.empty(N, 1, 1, 1, device="cuda"); dq = torch.empty_like(qd); da = torch.empty_like(ad)
    if seed % 2:      # the Layer-API pair of launches (row-aligned forward, workgroup-dense backward) ...
        capi.simcross_forward(1, qd, ad, top)
        capi.simcross_backward(1, qd, ad, top, dTd, dq, da)
    else:             # ... or the one-launch variant
        capi.simcross_forward_backward(1, qd, ad, dTd, top, dq, da)
    t, g, h = top.cpu().numpy(), dq.cpu().numpy(), da.cpu().numpy()
    m = int((t.view(np.uint32) != top_ref.view(np.uint32)).sum() + (g.view(np.uint32) != dq_ref.view(np.uint32)).sum()
            + (h.view(np.uint32) != da_ref.view(np.uint32)).sum())
    bad += m; total += N
    if m: print("seed", seed, "kind", kind, "N", N, "D", D, "mismatching words", m)
print("pairs checked %d, mismatching words %d, %.1f s" % (total, bad, time.time() - t0))
sys.exit(1 if bad else 0)
