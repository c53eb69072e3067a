"""Shared helpers for the parity tests."""
import numpy as np

SEED = 1701  # the reference's GradientChecker seed (test_gradient_check_util.hpp:25)
TOL = 1e-5   # BASELINE.json north_star: "within 1e-5 fp32"


def rng(extra=0):
    return np.random.default_rng(SEED + extra)


def qa(r, N, W1, W2, D, dtype=np.float32):
    """GloVe-like embeddings: N(0, 0.4^2) per coordinate."""
    q = (r.standard_normal((N, W1, D)) * 0.4).astype(dtype)
    a = (r.standard_normal((N, W2, D)) * 0.4).astype(dtype)
    return q, a


def assert_bitexact(x, y, what=""):
    x = np.ascontiguousarray(x)
    y = np.ascontiguousarray(y)
    assert x.shape == y.shape, (what, x.shape, y.shape)
    xb, yb = x.view(np.uint32), y.view(np.uint32)
    bad = np.flatnonzero(xb.ravel() != yb.ravel())
    # NaN payloads may differ legitimately; require NaN-ness to agree instead
    if bad.size:
        xf, yf = x.ravel()[bad], y.ravel()[bad]
        still = ~(np.isnan(xf) & np.isnan(yf))
        assert not still.any(), "%s: %d/%d elements differ bitwise, first at %d: %r vs %r" % (
            what, int(still.sum()), x.size, int(bad[still][0]), xf[still][0], yf[still][0])


def assert_close(x, y, tol=TOL, what=""):
    """|x-y| <= tol * max(1, max|y|): the north-star 1e-5 bound, scale-aware."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    assert x.shape == y.shape, (what, x.shape, y.shape)
    scale = max(1.0, float(np.max(np.abs(y))) if y.size else 1.0)
    err = float(np.max(np.abs(x - y))) if y.size else 0.0
    assert np.isfinite(err), "%s: non-finite difference" % what
    assert err <= tol * scale, "%s: max abs err %.3e > %.1e * %.3g" % (what, err, tol, scale)
