"""dev-only soak: the BLAS-ordered paths (cosine, bilinear, SimMatrix) and the word-grid Euclid paths against the
oracle over random geometries and input distributions -- 1e-5 * max(1, max|ref|) for the BLAS-ordered outputs,
bit for bit for the Euclid ones (reference-rounding mode); non-finite values must sit in the same places."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mms_answer_selection_amd import capi
from oracle import cpu_oracle as O
capi.set_euclid_backward_mode("reference")
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
bad = 0; checks = 0; t0 = time.time()
def close(x, ref, what, tol=1e-5):
    global bad, checks
    checks += 1
    x = x.cpu().numpy().astype(np.float64).reshape(ref.shape); r = ref.astype(np.float64)
    fin = np.isfinite(r)
    ok = (np.isfinite(x) == fin).all()
    if ok and fin.any():
        scale = max(1.0, float(np.abs(r[fin]).max()))
        ok = float(np.abs(x[fin] - r[fin]).max()) <= tol * scale
    if not ok:
        bad += 1
        print("MISMATCH", what)
def exact(x, ref, what):
    global bad, checks
    checks += 1
    x = x.cpu().numpy().reshape(ref.shape)
    same = (x.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(x) & np.isnan(ref))
    if not same.all():
        bad += 1
        print("MISMATCH (bits)", what, int((~same).sum()), "of", same.size)
def draw(r, kind, shape):
    if kind == 0: return r.standard_normal(shape) * 0.4
    if kind == 1: return r.uniform(-1, 1, shape)
    if kind == 2: return r.standard_t(2, shape)
    if kind == 3: return r.standard_normal(shape) * 1e-3
    x = r.standard_normal(shape) * 0.4
    x[r.uniform(size=shape[:-1]) < 0.05] = 0.0          # whole rows of zeros: zero norms, identical vectors
    return x
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    r = np.random.default_rng(7000 + seed)
    kind = seed % 5
    N = int(r.choice([1, 3, 50, 190, 600, 1517]))
    W1, W2 = (int(r.choice([1, 5, 8, 16, 33, 40, 48])) for _ in range(2))
    D = int(r.choice([9, 50, 52, 64, 100, 300]))
    if N * W1 * W2 * D > 6e7: N = 50
    q = draw(r, kind, (N, W1, D)).astype(np.float32); a = draw(r, kind, (N, W2, D)).astype(np.float32)
    tag = "seed %d kind %d %s" % (seed, kind, (N, W1, W2, D))
    for mode in (0, 1):
        top_ref, n0_ref, n1_ref = O.simcross_forward(mode, q, a)
        dT = r.standard_normal(top_ref.shape).astype(np.float32)
        top = torch.empty(top_ref.shape, device="cuda"); n0 = torch.empty(N, W1, device="cuda"); n1 = torch.empty(N, W2, device="cuda")
        qd, ad = dev(q), dev(a)
        capi.simcross_forward(mode, qd, ad, top, norm0=n0, norm1=n1)
        (exact if mode == 1 else close)(top, top_ref, tag + " mode %d top" % mode)
        with np.errstate(all="ignore"):
            dq_ref, da_ref, _, _ = O.simcross_backward(mode, q, a, top_ref, dT, norm0=n0_ref, norm1=n1_ref)
        dq, da = torch.empty_like(qd), torch.empty_like(ad)
        capi.simcross_backward(mode, qd, ad, dev(top_ref), dev(dT), dq, da, norm0=dev(n0_ref) if mode == 0 else None, norm1=dev(n1_ref) if mode == 0 else None)
        (exact if mode == 1 else close)(dq, dq_ref, tag + " mode %d dq" % mode)
        (exact if mode == 1 else close)(da, da_ref, tag + " mode %d da" % mode)
    # bilinear
    M = int(r.choice([1, 2, 4]))
    if D <= 100:
        Wt = (r.standard_normal((M, D, D)) * 0.05).astype(np.float32); bias = (r.standard_normal((M, W1, W2)) * 0.1).astype(np.float32)
        top_ref, _, _ = O.simcross_forward(2, q, a, W=Wt, bias=bias)
        top = torch.empty(top_ref.shape, device="cuda")
        capi.simcross_forward(2, dev(q), dev(a), top, W=dev(Wt), bias=dev(bias))
        close(top, top_ref, tag + " bilinear M=%d top" % M)
        dT = r.standard_normal(top_ref.shape).astype(np.float32)
        dq_ref, da_ref, dW_ref, db_ref = O.simcross_backward(2, q, a, top_ref, dT, W=Wt, bias_term=True, dbias_in=np.zeros_like(bias))
        dq, da = torch.empty(q.shape, device="cuda"), torch.empty(a.shape, device="cuda")
        dW, db = torch.empty(Wt.shape, device="cuda"), torch.zeros(bias.shape, device="cuda")
        capi.simcross_backward(2, dev(q), dev(a), dev(top_ref), dev(dT), dq, da, W=dev(Wt), bias_term=True, dW=dW, dbias=db)
        close(dq, dq_ref, tag + " bilinear dq"); close(da, da_ref, tag + " bilinear da")
        close(dW, dW_ref, tag + " bilinear dW"); close(db, db_ref, tag + " bilinear dbias")
print("checks %d, mismatches %d, %.1f s" % (checks, bad, time.time() - t0))
