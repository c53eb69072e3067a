#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerate the committed golden vectors.

PROVENANCE: these vectors are produced by THIS repository's CPU oracle
(oracle/mms_oracle.c), not by the reference: the reference ships no fixture for
the MMS layers and cannot be built or imported in this environment (DESIGN.md
section 2).  They freeze the oracle (so an accidental change of the checker is
caught by tests/test_golden.py) and give the GPU tests fixed inputs/outputs that
do not depend on a numpy RNG implementation.  Shapes follow SURVEY.md 8(c).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cpu_oracle as O  # noqa: E402


def qa(r, N, W1, W2, D):
    q = (r.standard_normal((N, W1, D)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N, W2, D)) * 0.4).astype(np.float32)
    return q, a


def main():
    r = np.random.default_rng(1701)
    # SimCross, modes 0/1/2 at the three geometries
    for (N, W1, W2, D, M) in ((8, 1, 1, 300, 1), (4, 5, 7, 300, 2), (2, 40, 40, 50, 4)):
        q, a = qa(r, N, W1, W2, D)
        a[1, 0] = q[1, 0]                                  # degenerate pair (T = 1, divisor 1e-9)
        out = dict(q=q, a=a)
        for mode in (0, 1, 2):
            Wt = r.uniform(-0.08, 0.08, (M, D, D)).astype(np.float32) if mode == 2 else None
            bias = r.standard_normal((M, W1, W2)).astype(np.float32) if mode == 2 else None
            qq, aa = (q, a) if mode != 0 else (q, a + np.float32(0.01))   # avoid 0/0 rows for cosine
            top, n0, n1 = O.simcross_forward(mode, qq, aa, Wt, bias)
            dT = r.standard_normal(top.shape).astype(np.float32)
            db0 = r.standard_normal((M, W1, W2)).astype(np.float32) if mode == 2 else None
            dq, da, dW, db = O.simcross_backward(mode, qq, aa, top, dT, W=Wt, bias_term=mode == 2,
                                                 norm0=n0, norm1=n1, dbias_in=db0)
            p = "m%d_" % mode
            out.update({p + "top": top, p + "dT": dT, p + "dq": dq, p + "da": da})
            if mode == 0:
                out.update({p + "a": aa, p + "n0": n0, p + "n1": n1})
            if mode == 2:
                out.update({p + "W": Wt, p + "bias": bias, p + "dW": dW, p + "dbias_in": db0, p + "dbias": db})
        np.savez_compressed(os.path.join(HERE, "simcross_%d_%d_%d_%d_%d.npz" % (N, W1, W2, D, M)), **out)
    # SimMatrix
    for (N, K1, K2) in ((16, 300, 300), (5, 7, 3)):
        q = (r.standard_normal((N, K1)) * 0.4).astype(np.float32)
        a = (r.standard_normal((N, K2)) * 0.4).astype(np.float32)
        Wt = r.uniform(-0.08, 0.08, (K1, K2)).astype(np.float32)
        dT = r.standard_normal((N, 1)).astype(np.float32)
        dW0 = r.standard_normal((K1, K2)).astype(np.float32)
        top, scratch = O.simmatrix_forward(q, a, Wt)
        dq, da, dW = O.simmatrix_backward(q, a, Wt, dT, dW_in=dW0)
        np.savez_compressed(os.path.join(HERE, "simmatrix_%d_%d_%d.npz" % (N, K1, K2)), q=q, a=a, W=Wt, dT=dT,
                            dW_in=dW0, top=top, scratch=scratch, dq=dq, da=da, dW=dW)
    # PairRankLoss: scores on both sides of the margin, y ~ Bernoulli(0.2)
    for (N, C, margin) in ((64, 1, 1.0), (8, 5, 0.1)):
        a = r.uniform(0, 1, (N, C)).astype(np.float32)
        b = r.uniform(0, 1, (N, C)).astype(np.float32)
        y = (r.uniform(size=(N, C)) < 0.2).astype(np.float32)
        a[0] = b[0]
        loss, o, s = O.pairrank_forward(a, b, y, margin)
        da, db = O.pairrank_backward(y, o, s, top_diff=1.0)
        np.savez_compressed(os.path.join(HERE, "pairrank_%d_%d.npz" % (N, C)), a=a, b=b, y=y,
                            margin=np.float32(margin), loss=np.float32(loss), ordered=o, similar=s, da=da, db=db)
    # MAP / MRR / AUC: 1517 scores in 68 groups incl. all-positive / all-negative groups
    n, groups = 1517, 68
    group = np.sort(r.integers(0, groups, n)).astype(np.float32)
    label = (r.uniform(size=n) < 0.17).astype(np.float32)
    label[group == 3] = 1
    label[group == 5] = 0
    score = r.uniform(size=n).astype(np.float32)
    prob = np.stack([1 - score, score], 1).astype(np.float32)
    m, eff = O.map_score(prob, label, group)
    rr, _ = O.mrr_score(prob, label, group)
    auc = O.auc_score(prob, label)
    np.savez_compressed(os.path.join(HERE, "ranking_1517_68.npz"), prob=prob, label=label, group=group,
                        map=np.float32(m), mrr=np.float32(rr), auc=np.float32(auc), effective=np.int32(eff))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
