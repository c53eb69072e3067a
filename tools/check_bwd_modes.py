import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mms_answer_selection_amd import capi
from oracle import cpu_oracle as O
r = np.random.default_rng(0)
for N, D in ((4096, 300), (4097, 300), (1, 300), (333, 200), (1000, 100)):
    q = (r.standard_normal((N, 1, D)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N, 1, D)) * 0.4).astype(np.float32)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    top_ref, _, _ = O.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = O.simcross_backward(1, q, a, top_ref, dT)
    qd, ad, dTd = (torch.from_numpy(x).cuda() for x in (q, a, dT))
    for mode in ("fp32", "reference"):
        capi.set_euclid_backward_mode(mode)
        top = torch.empty(N, 1, 1, 1, device="cuda"); dq = torch.full_like(qd, 7.0); da = torch.full_like(ad, 7.0)
        capi.simcross_forward_backward(1, qd, ad, dTd, top, dq, da)
        torch.cuda.synchronize()
        t, g, h = top.cpu().numpy(), dq.cpu().numpy(), da.cpu().numpy()
        top_ok = (t.view(np.uint32) == top_ref.view(np.uint32)).all()
        ulp = np.abs(g.view(np.int32).astype(np.int64) - dq_ref.view(np.int32).astype(np.int64))
        rel = np.max(np.abs(g - dq_ref) / np.maximum(np.abs(dq_ref), 1e-30))
        print(N, D, mode, "top bit-exact", top_ok, "dq max ulp", ulp.max(), "frac differing", (ulp > 0).mean(), "max rel", rel,
              "da==-dq", (h == -g).all())
