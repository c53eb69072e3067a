"""Worker of tests/test_gpu_layouts.py: run with MMS_EUCLID_LAYOUT_{FWD,BWD,FUSED} set to the layouts that
are NOT the library's defaults (the switch is read once per process) and compare the three kinds of launch of
the GloVe-width Euclidean path with the CPU oracle, bit for bit (reference rounding of the backward term)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import qa, rng  # noqa: E402
from oracle import cpu_oracle  # noqa: E402
from mms_answer_selection_amd import capi  # noqa: E402

cpu_oracle.build()
capi.set_euclid_backward_mode(capi.EUCLID_BWD_REFERENCE)
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
nan = lambda s: torch.full(s, float("nan"), dtype=torch.float32, device="cuda")
bad = 0
for D in (100, 200, 300):
    for N in (1, 2, 3, 15, 16, 17, 33, 255, 1000, 4096):
        r = rng(1000 * D + N)
        q, a = qa(r, N, 1, 1, D)
        if N >= 4:
            a[1, 0] = q[1, 0]                        # T = 1, divisor 1e-9
        dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
        top_ref, _, _ = cpu_oracle.simcross_forward(1, q, a)
        dq_ref, da_ref, _, _ = cpu_oracle.simcross_backward(1, q, a, top_ref, dT)
        qd, ad, dTd = dev(q), dev(a), dev(dT)
        top, gq, ga = nan(top_ref.shape), nan(q.shape), nan(a.shape)
        capi.simcross_forward(1, qd, ad, top)
        capi.simcross_backward(1, qd, ad, top, dTd, gq, ga)
        top2, gq2, ga2 = nan(top_ref.shape), nan(q.shape), nan(a.shape)
        capi.simcross_forward_backward(1, qd, ad, dTd, top2, gq2, ga2)
        for name, got, ref in (("top", top, top_ref), ("dq", gq, dq_ref), ("da", ga, da_ref),
                               ("fused top", top2, top_ref), ("fused dq", gq2, dq_ref), ("fused da", ga2, da_ref)):
            g = got.cpu().numpy()
            if not np.array_equal(g.view(np.uint32), ref.view(np.uint32)):
                bad += 1
                print("MISMATCH D=%d N=%d %s: %d words" % (D, N, name, int((g.view(np.uint32) != ref.view(np.uint32)).sum())))
print("layouts fwd=%s bwd=%s fused=%s: mismatching outputs %d" % (
    os.environ.get("MMS_EUCLID_LAYOUT_FWD"), os.environ.get("MMS_EUCLID_LAYOUT_BWD"),
    os.environ.get("MMS_EUCLID_LAYOUT_FUSED"), bad))
sys.exit(1 if bad else 0)
