"""dev-only soak: the fused (q, a+, a-) step (two triplets per wave, loss summed inside the launch) against the
layer-by-layer oracle on many random batches and input distributions: scores and reference-mode gradients bit
for bit; the loss scalar against the EXACT (float64) mean of the oracle's per-triplet terms to 2e-6, and against
the oracle's own float32 running sum to 1e-5 where that sum is itself within 1e-5 of the exact mean (for
near-constant terms of a few thousand triplets it is not: every add of ~2 to a sum in [4096, 8192) rounds the
same way, and the reference's CPU loss drifts by 1-2e-5 -- its GPU path, a cuBLAS reduction, does not share
that drift either); both finish modes; counts any mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mms_answer_selection_amd import capi
from oracle import cpu_oracle as O
capi.set_euclid_backward_mode("reference")
bad = 0; total = 0; t0 = time.time()
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    r = np.random.default_rng(1000 + seed)
    N = int(r.choice([4096, 4095, 1233, 8192, 2, 1]))
    D = int(r.choice([300, 300, 200, 100]))
    kind = seed % 4
    if kind == 0: q = r.standard_normal((N, 1, D)) * 0.4; ap = q + 0.1 * r.standard_normal((N, 1, D)); an = r.standard_normal((N, 1, D)) * 0.4
    elif kind == 1: q = r.uniform(-1, 1, (N, 1, D)); ap = r.uniform(-1, 1, (N, 1, D)); an = r.uniform(-1, 1, (N, 1, D))
    elif kind == 2: q = r.standard_t(2, (N, 1, D)); ap = r.standard_t(2, (N, 1, D)); an = r.standard_t(2, (N, 1, D))
    else: q = r.standard_normal((N, 1, D)) * 1e-3; ap = q + r.standard_normal((N, 1, D)) * 1e-5; an = q + r.standard_normal((N, 1, D)) * 1e-4
    q, ap, an = (x.astype(np.float32) for x in (q, ap, an))
    y = (r.uniform(size=(N, 1)) < 0.8).astype(np.float32)
    margin = float(r.choice([0.05, 0.5, 2.0]))
    sp, _, _ = O.simcross_forward(1, q, ap)
    sn, _, _ = O.simcross_forward(1, q, an)
    loss_ref, o, s = O.pairrank_forward(sp.reshape(N, 1), sn.reshape(N, 1), y, margin)
    gsp, gsn = O.pairrank_backward(y, o, s, top_diff=1.0)
    dq_p, dap_ref, _, _ = O.simcross_backward(1, q, ap, sp, gsp.reshape(sp.shape))
    dq_n, dan_ref, _, _ = O.simcross_backward(1, q, an, sn, gsn.reshape(sn.shape))
    dq_ref = dq_p + dq_n
    o64, s64 = o.astype(np.float64), s.astype(np.float64)
    exact = float((np.maximum(0.0, o64) + np.abs((1.0 - y.astype(np.float64)) * s64)).mean())   # pair_rank_loss_layer.cpp:43-49 in float64
    ref_ok = abs(loss_ref - exact) <= 1e-5 * max(1.0, abs(exact))
    drifted = globals().get("drifted", 0) + (0 if ref_ok else 1)
    dev = lambda x: torch.from_numpy(x).cuda()
    for mode in ("inlaunch", "launch"):
        capi.set_triplet_finish_mode(mode)
        out = dict(s_pos=torch.empty(N, 1, device="cuda"), s_neg=torch.empty(N, 1, device="cuda"), loss=torch.full((1,), float("nan"), device="cuda"),
                   dq=torch.empty(N, 1, D, device="cuda"), da_pos=torch.empty(N, 1, D, device="cuda"), da_neg=torch.empty(N, 1, D, device="cuda"))
        capi.triplet_euclid_step(dev(q), dev(ap), dev(an), dev(y), margin=margin, **out)
        h = {k: v.cpu().numpy() for k, v in out.items()}
        eq = lambda a, b: (a.view(np.uint32) == b.reshape(a.shape).view(np.uint32)).all()
        ok = (eq(h["s_pos"], sp) and eq(h["s_neg"], sn) and eq(h["dq"], dq_ref) and eq(h["da_pos"], dap_ref) and eq(h["da_neg"], dan_ref)
              and abs(h["loss"][0] - exact) <= 2e-6 * max(1.0, abs(exact))
              and (not ref_ok or abs(h["loss"][0] - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "mode", mode, "N", N, "D", D, "kind", kind, "loss", h["loss"][0], loss_ref)
        total += N
    # MMS_LOSS_SUM_REFERENCE: the oracle's running fp32 sum, bit for bit
    capi.set_loss_sum_mode("reference")
    lo = torch.full((1,), float("nan"), device="cuda")
    out["loss"] = lo
    capi.triplet_euclid_step(dev(q), dev(ap), dev(an), dev(y), margin=margin, **out)
    capi.set_loss_sum_mode("fast")
    if lo.cpu().numpy().view(np.uint32)[0] != np.float32(loss_ref).view(np.uint32):
        bad += 1
        print("MISMATCH (reference loss sum) seed", seed, "N", N, "D", D, lo.item(), loss_ref)
capi.set_triplet_finish_mode("inlaunch")
print("triplets checked (both modes): %d, mismatching batches: %d, batches where the oracle's float32 running sum is itself > 1e-5 from the exact mean: %d, %.1f s" % (total, bad, drifted, time.time() - t0))
