"""CPU suite, part 1: pin the oracle (oracle/) as far as it can be pinned.

The reference holds NO golden vector, fixture or test for SimCross, SimMatrix,
PairRankLoss or the ranking layers and cannot be compiled
here, so parity is UNPINNED by the reference.  What these tests do instead:
  * float64 closed forms written independently in numpy (forward values);
  * central finite differences of the oracle's forward against its backward,
    the method of the reference's GradientChecker
    (include/caffe/test/test_gradient_check_util.hpp:148-175: stepsize 1e-2,
    threshold 1e-2 relative with scale floor 1; kinks skipped);
  * the reference-visible quirks listed in SURVEY.md Appendix A.
"""
import numpy as np
import pytest

from util import assert_bitexact, assert_close, qa, rng


SHAPES = [(8, 1, 1, 300), (4, 5, 7, 300), (2, 40, 40, 50)]   # SURVEY 8c golden shapes


@pytest.mark.parametrize("shape", SHAPES)
def test_euclid_closed_form(shape, oracle):
    N, W1, W2, D = shape
    q, a = qa(rng(), N, W1, W2, D)
    top, _, _ = oracle.simcross_forward(1, q, a)
    q6, a6 = q.astype(np.float64), a.astype(np.float64)
    dist = np.sqrt(((q6[:, :, None, :] - a6[:, None, :, :]) ** 2).sum(-1))
    assert_close(top[:, 0], 1 / (1 + dist), 1e-6)
    dT = rng(1).standard_normal(top.shape).astype(np.float32)
    dq, da, _, _ = oracle.simcross_backward(1, q, a, top, dT)
    g = -(dT[:, 0].astype(np.float64) / (1 + dist) ** 2 / dist)[..., None] * (
        q6[:, :, None, :] - a6[:, None, :, :])
    assert_close(dq, g.sum(2), 1e-5)
    assert_close(da, -g.sum(1), 1e-5)


def test_euclid_degenerate_pair(oracle):
    """q == a: T = 1 exactly, divisor (T-1+1e-9) = 1e-9, numerator 0 -> zero grads."""
    q, a = qa(rng(), 3, 1, 1, 16)
    a[1] = q[1]
    top, _, _ = oracle.simcross_forward(1, q, a)
    assert top[1, 0, 0, 0] == np.float32(1.0)
    dq, da, _, _ = oracle.simcross_backward(1, q, a, top, np.ones_like(top))
    assert (dq[1] == 0).all() and (da[1] == 0).all()
    assert np.isfinite(dq).all()


def test_euclid_double_divisor_is_observable(oracle):
    """The 1e-9 literal makes the division double (sim_cross_layer.cpp:217); an
    all-float evaluation differs in the last bit for some inputs."""
    q, a = qa(rng(7), 64, 1, 1, 300)
    top, _, _ = oracle.simcross_forward(1, q, a)
    dT = rng(8).standard_normal(top.shape).astype(np.float32)
    dq, _, _, _ = oracle.simcross_backward(1, q, a, top, dT)
    T = top.reshape(64, 1, 1)
    g = dT.reshape(64, 1, 1)
    num = (g * T * T * T * (q - a)).astype(np.float32)
    ref = (num.astype(np.float64) / ((T - np.float32(1)).astype(np.float64) + 1e-9)).astype(np.float32)
    assert_bitexact(dq, ref + np.float32(0))
    allfloat = num / ((T - np.float32(1)) + np.float32(1e-9))
    assert (allfloat.view(np.uint32) != dq.view(np.uint32)).any()


@pytest.mark.parametrize("shape", SHAPES)
def test_cosine_closed_form(shape, oracle):
    N, W1, W2, D = shape
    q, a = qa(rng(2), N, W1, W2, D)
    top, n0, n1 = oracle.simcross_forward(0, q, a)
    q6, a6 = q.astype(np.float64), a.astype(np.float64)
    nq, na = np.linalg.norm(q6, axis=-1), np.linalg.norm(a6, axis=-1)
    assert_close(n0, nq, 1e-6)       # the NORM is cached (cpp:118), not its square (cu:38)
    assert_close(n1, na, 1e-6)
    cos = np.einsum("njd,nkd->njk", q6, a6) / nq[:, :, None] / na[:, None, :]
    assert_close(top[:, 0], cos, 1e-6)


@pytest.mark.parametrize("shape", [(8, 1, 1, 300, 1), (4, 5, 7, 300, 2), (2, 40, 40, 50, 4)])
def test_bilinear_closed_form(shape, oracle):
    N, W1, W2, D, M = shape
    r = rng(3)
    q, a = qa(r, N, W1, W2, D)
    W = r.uniform(-0.08, 0.08, (M, D, D)).astype(np.float32)
    b = r.standard_normal((M, W1, W2)).astype(np.float32)
    top, _, _ = oracle.simcross_forward(2, q, a, W, b)
    q6, a6, W6 = q.astype(np.float64), a.astype(np.float64), W.astype(np.float64)
    assert_close(top, np.einsum("njd,mde,nke->nmjk", q6, W6, a6) + b, 1e-5)
    dT = r.standard_normal(top.shape).astype(np.float32)
    db0 = r.standard_normal(b.shape).astype(np.float32)
    dq, da, dW, db = oracle.simcross_backward(2, q, a, top, dT, W=W, bias_term=True, dbias_in=db0)
    assert_close(dq, np.einsum("nmjk,mde,nke->njd", dT, W6, a6), 1e-5)
    assert_close(da, np.einsum("nmjk,mde,njd->nke", dT, W6, q6), 1e-5)
    assert_close(dW, np.einsum("nmjk,njd,nke->mde", dT, q6, a6), 1e-5)   # zeroed first (:256)
    assert_close(db, db0 + dT.astype(np.float64).sum(0), 1e-5)           # accumulated (:301-304)


def _numeric_grad(f, x, eps=1e-2):
    g = np.zeros_like(x, dtype=np.float64)
    flat = x.reshape(-1)
    for i in range(flat.size):
        old = flat[i]
        flat[i] = old + eps
        fp = f()
        flat[i] = old - eps
        fm = f()
        flat[i] = old
        g.reshape(-1)[i] = (fp - fm) / (2 * eps)
    return g


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_simcross_gradient_check(mode, oracle):
    """GradientChecker-style: objective = sum(top * dT) in float64."""
    r = rng(10 + mode)
    N, W1, W2, D, M = 2, 3, 2, 5, (2 if mode == 2 else 1)
    q = r.standard_normal((N, W1, D))
    a = r.standard_normal((N, W2, D))
    W = r.uniform(-0.5, 0.5, (M, D, D)) if mode == 2 else None
    b = r.standard_normal((M, W1, W2)) if mode == 2 else None
    top, n0, n1 = oracle.simcross_forward(mode, q, a, W, b)
    dT = r.standard_normal(top.shape)
    dq, da, dW, db = oracle.simcross_backward(mode, q, a, top, dT, W=W, bias_term=mode == 2,
                                              norm0=n0, norm1=n1)
    f = lambda: float((oracle.simcross_forward(mode, q, a, W, b)[0] * dT).sum())
    thr = 1e-2
    for name, x, g in (("q", q, dq), ("a", a, da)) + ((("W", W, dW), ("bias", b, db)) if mode == 2 else ()):
        ng = _numeric_grad(f, x, 1e-4)
        scale = np.maximum(1.0, np.maximum(np.abs(ng), np.abs(g)))
        assert (np.abs(ng - g) <= thr * scale).all(), name


def test_simmatrix_closed_form_and_quirks(oracle):
    r = rng(4)
    N, K1, K2 = 16, 300, 300
    q = (r.standard_normal((N, K1)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N, K2)) * 0.4).astype(np.float32)
    W = r.uniform(-0.08, 0.08, (K1, K2)).astype(np.float32)
    top, scratch = oracle.simmatrix_forward(q, a, W)
    q6, a6, W6 = q.astype(np.float64), a.astype(np.float64), W.astype(np.float64)
    assert_close(scratch, q6 @ W6, 1e-5)          # left in bottom[1].diff (:58)
    assert_close(top, ((q6 @ W6) * a6).sum(1, keepdims=True), 1e-5)
    dT = r.standard_normal((N, 1)).astype(np.float32)
    dW0 = r.standard_normal((K1, K2)).astype(np.float32)
    dq, da, dW = oracle.simmatrix_backward(q, a, W, dT, dW_in=dW0)
    assert_close(dq, dT * (a6 @ W6.T), 1e-5)
    assert_close(da, dT * (q6 @ W6), 1e-5)
    assert_close(dW, dW0 + q6.T @ (dT * a6), 1e-5)   # accumulates (:73-80)


def test_pairrank_formula_and_kinks(oracle):
    r = rng(5)
    N = 64
    a = r.uniform(0, 1, (N, 1)).astype(np.float32)
    b = r.uniform(0, 1, (N, 1)).astype(np.float32)
    y = (r.uniform(size=(N, 1)) < 0.2).astype(np.float32)
    for margin in (1.0, 0.1):
        loss, o, s = oracle.pairrank_forward(a, b, y, margin)
        a6, b6, y6 = (x.astype(np.float64) for x in (a, b, y))
        ref = (np.maximum(0, margin - y6 * (a6 - b6)) + np.abs((1 - y6) * (a6 - b6))).mean()
        assert abs(loss - ref) < 1e-6                     # reference test tolerance
        da, db = oracle.pairrank_backward(y, o, s, top_diff=1.0)
        # y=0 rows contribute the constant `margin` to the loss (Appendix A.9)
        assert (o[y == 0] == np.float32(margin)).all()
        # finite differences away from the kinks (kink at ordered = 0 and similar = 0)
        eps = 1e-3
        for i in range(N):
            if abs(o[i, 0]) < 2 * eps or (y[i, 0] == 0 and abs(s[i, 0]) < 2 * eps):
                continue
            ap = a.copy(); ap[i] += eps
            am = a.copy(); am[i] -= eps
            num = (float(oracle.pairrank_forward(ap.astype(np.float64), b6, y6, margin)[0]) -
                   float(oracle.pairrank_forward(am.astype(np.float64), b6, y6, margin)[0])) / (2 * eps)
            assert abs(num - da[i, 0]) < 1e-4, (i, num, da[i, 0])
            assert db[i, 0] == -da[i, 0]
    # strict '>' of the CPU code (:76): ordered == 0 gives ordered_t = 0
    o = np.zeros((1, 1), np.float32); s = np.ones((1, 1), np.float32); y1 = np.ones((1, 1), np.float32)
    da, _ = oracle.pairrank_backward(y1, o, s, top_diff=1.0)
    assert da[0, 0] == 0.0


def test_map_mrr_auc(oracle):
    r = rng(6)
    n, groups = 1517, 68                         # TREC-QA test split (do_trec_qa_clean.py:39-41)
    group = np.sort(r.integers(0, groups, n)).astype(np.float32)
    label = (r.uniform(size=n) < 0.17).astype(np.float32)
    label[group == 3] = 1                         # an all-positive group is skipped
    label[group == 5] = 0                         # an all-negative group is skipped
    score = r.uniform(size=n).astype(np.float32)  # distinct with overwhelming probability
    prob = np.stack([1 - score, score], 1).astype(np.float32)
    m, eff = oracle.map_score(prob, label, group)
    rr, eff2 = oracle.mrr_score(prob, label, group)
    aps, rrs = [], []
    for gidx in np.unique(group):
        sel = group == gidx
        l = label[sel][np.argsort(-score[sel], kind="stable")]
        if l.sum() < 1 or l.sum() == l.size:
            continue
        hits = np.flatnonzero(l == 1)
        aps.append(np.mean([(i + 1) / (h + 1) for i, h in enumerate(hits)]))
        rrs.append(1.0 / (hits[0] + 1))
    assert eff == eff2 == len(aps)
    assert abs(m - np.mean(aps)) < 1e-5 and abs(rr - np.mean(rrs)) < 1e-5
    auc = oracle.auc_score(prob, label)
    pos, neg = score[label == 1], score[label == 0]
    ref = (pos[:, None] > neg[None, :]).mean()
    assert abs(auc - ref) < 1e-4


def test_oracle_double_instantiation(oracle):
    """INSTANTIATE_CLASS covers float and double (common.hpp:41-44)."""
    q, a = qa(rng(9), 3, 2, 2, 10, np.float64)
    top, _, _ = oracle.simcross_forward(1, q, a)
    assert top.dtype == np.float64
    dist = np.sqrt(((q[:, :, None, :] - a[:, None, :, :]) ** 2).sum(-1))
    assert_close(top[:, 0], 1 / (1 + dist), 1e-12)
