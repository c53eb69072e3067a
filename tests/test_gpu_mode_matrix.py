"""The arithmetic switches of the C ABI (include/mms.h: mms_set_*_mode) in COMBINATION: 2^7 = 128 settings.  A switch
changes one documented thing and nothing else, so under every combination each entry point must still land inside its
own bar against the oracle, and the outputs a switch does not own must keep the bits they have under the defaults."""
import itertools

import numpy as np
import pytest
import torch

from util import rng, qa, assert_close, assert_bitexact, TOL

pytestmark = pytest.mark.gpu

SWITCHES = [("euclid_backward", ("fp32", "reference")), ("pairrank_hinge", ("cpu", "gpu")), ("f16_distance", ("ordered", "tree")),
            ("rank_tie", ("input", "libstdcxx")), ("loss_sum", ("fast", "reference")), ("triplet_finish", ("inlaunch", "launch")),
            ("matrix", ("bf16x3", "fp32"))]


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def nan_like(shape, dtype=torch.float32):
    return torch.full(tuple(shape), float("nan"), dtype=dtype, device="cuda")


def _apply(capi, combo):
    for (name, _), v in zip(SWITCHES, combo):
        getattr(capi, "set_%s_mode" % name)(v)


def test_every_combination_of_the_mode_switches(oracle, hiplib):
    from mms_answer_selection_amd import capi
    r = rng(2024)
    # one small case per family of entry points the switches touch
    N, D = 96, 300
    q, a = qa(r, N, 1, 1, D)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    qd, ad, dTd = dev(q), dev(a), dev(dT)
    # PairRankLoss on scores with exact ties at the hinge (the cpu / gpu modes differ there, by design)
    sa = r.standard_normal((257, 1)).astype(np.float32)
    sb = sa.copy(); sb[::3] -= 0.5; sb[1::3] += 0.25
    y = (r.uniform(size=(257, 1)) < 0.7).astype(np.float32)
    loss_ref, o_ref, s_ref = oracle.pairrank_forward(sa, sb, y, 0.5)
    # SimMatrix at a size the bf16 pipe takes
    Nm, K = 2176, 64
    qm = (r.standard_normal((Nm, K)) * 0.4).astype(np.float32)
    am = (r.standard_normal((Nm, K)) * 0.4).astype(np.float32)
    Wm = r.uniform(-0.08, 0.08, (K, K)).astype(np.float32)
    tm_ref, _ = oracle.simmatrix_forward(qm, am, Wm)
    qmd, amd, Wmd = dev(qm), dev(am), dev(Wm)
    # fp16-storage Euclid
    q16, a16 = qa(r, 40, 1, 1, 304)                                     # (D % 8 == 0)
    qh, ah = q16.astype(np.float16), a16.astype(np.float16)
    th_ref, _, _ = oracle.simcross_forward(1, qh.astype(np.float32), ah.astype(np.float32))
    qhd, ahd = dev(qh), dev(ah)
    # ranking: distinct scores (the tie mode must not matter)
    n = 300
    p1 = r.permutation(n).astype(np.float32) / n
    prob2 = np.stack([1.0 - p1, p1], axis=1).astype(np.float32)          # (n, 2): channel fixed_axis = 1 is the score
    grp = np.sort(r.integers(0, 23, n)).astype(np.float32).reshape(n, 1)
    lab = (r.uniform(size=(n, 1)) < 0.3).astype(np.float32)
    map_ref, _ = oracle.map_score(prob2, lab, grp)
    mrr_ref, _ = oracle.mrr_score(prob2, lab, grp)

    base = {}
    try:
        for combo in itertools.product(*[vals for _, vals in SWITCHES]):
            _apply(capi, combo)
            tag = "/".join(combo)
            top, gq, ga = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
            capi.simcross_forward(1, qd, ad, top)
            capi.simcross_backward(1, qd, ad, top, dTd, gq, ga)
            assert_bitexact(host(top), top_ref, "Euclid scores [%s]" % tag)
            if combo[0] == "reference":
                assert_bitexact(host(gq), dq_ref, "Euclid dq [%s]" % tag)
            else:
                assert_close(host(gq), dq_ref, TOL, "Euclid dq [%s]" % tag)
            po, ps, pl = nan_like(sa.shape), nan_like(sa.shape), nan_like((1,))
            capi.pairrank_forward(dev(sa), dev(sb), dev(y), po, ps, pl, margin=0.5)
            assert_bitexact(host(po), o_ref, "PairRankLoss ordered_diff [%s]" % tag)
            assert_close(host(pl)[0], loss_ref, TOL, "loss [%s]" % tag)
            if combo[4] == "reference":
                assert host(pl)[0] == np.float32(loss_ref), "reference loss sum is the reference's bits [%s]" % tag
            tm, scr = nan_like((Nm, 1)), nan_like((Nm, K))
            capi.simmatrix_forward(qmd, amd, Wmd, tm, scr)
            assert_close(host(tm), tm_ref, TOL, "SimMatrix scores [%s]" % tag)
            th = nan_like(th_ref.shape)
            capi.simcross_euclid_forward_f16(qhd, ahd, th)
            if combo[2] == "ordered":
                assert_bitexact(host(th), th_ref, "fp16-storage scores [%s]" % tag)
            else:
                assert_close(host(th), th_ref, TOL, "fp16-storage scores, tree [%s]" % tag)
            mres, rres, _ = capi.rank_map_mrr(dev(prob2), dev(lab), dev(grp))
            mo, ro = np.float32([mres]), np.float32([rres])
            # outputs a switch does not own keep the bits of the first combination that shares their own switch
            own = {"top": (), "po": (1,), "tm": (6,), "th": (2,), "map": (), "mrr": ()}
            got = {"top": host(top), "po": host(po), "tm": host(tm), "th": host(th), "map": mo, "mrr": ro}
            for k, idx in own.items():
                key = (k,) + tuple(combo[i] for i in idx)
                if key in base:
                    assert_bitexact(got[k], base[key], "%s changed with a switch that does not own it [%s]" % (k, tag))
                else:
                    base[key] = got[k]
            assert mo[0] == np.float32(map_ref) and ro[0] == np.float32(mrr_ref), "MAP / MRR [%s]" % tag
    finally:
        _apply(capi, tuple(vals[0] for _, vals in SWITCHES))
