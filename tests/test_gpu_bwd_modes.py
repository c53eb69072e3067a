"""The two arithmetic modes of the Euclidean backward term (include/mms.h:
mms_set_euclid_backward_mode).  Forward scores are bit-identical to the oracle in both; the
default fp32 mode keeps every gradient element within 2 ulp of the reference's value (the
north-star bar is 1e-5 relative), the reference mode reproduces the bits."""
import numpy as np
import pytest
import torch

from mms_answer_selection_amd import capi

pytestmark = pytest.mark.gpu


def _case(r, N, D, scale=0.4):
    q = (r.standard_normal((N, 1, D)) * scale).astype(np.float32)
    a = (r.standard_normal((N, 1, D)) * scale).astype(np.float32)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    return q, a, dT


def _ulps(x, ref):
    return np.abs(x.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))


@pytest.mark.parametrize("N,D", [(4096, 300), (4097, 300), (1, 300), (7, 200), (333, 200), (1000, 100), (2, 100)])
def test_fp32_mode_is_within_two_ulp_and_forward_is_exact(N, D, oracle, hiplib):
    r = np.random.default_rng(N * 1000 + D)
    q, a, dT = _case(r, N, D)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    qd, ad, dTd = (torch.from_numpy(x).cuda() for x in (q, a, dT))
    capi.set_euclid_backward_mode("fp32")
    assert capi.get_euclid_backward_mode() == "fp32"
    top = torch.empty(N, 1, 1, 1, device="cuda")
    dq = torch.full_like(qd, 7.0)
    da = torch.full_like(ad, 7.0)
    capi.simcross_forward_backward(1, qd, ad, dTd, top, dq, da)
    t, g, h = top.cpu().numpy(), dq.cpu().numpy(), da.cpu().numpy()
    assert (t.view(np.uint32) == top_ref.view(np.uint32)).all()          # ranking-relevant output: exact
    assert _ulps(g, dq_ref).max() <= 2 and _ulps(h, da_ref).max() <= 2
    np.testing.assert_allclose(g, dq_ref, rtol=1e-5, atol=0)             # the stated bar, with room
    assert (h == -g).all()
    # the unfused entry points (Layer API: Forward then Backward) follow the same mode
    top2 = torch.empty_like(top)
    capi.simcross_forward(1, qd, ad, top2)
    dq2 = torch.empty_like(qd)
    da2 = torch.empty_like(ad)
    capi.simcross_backward(1, qd, ad, top2, dTd, dq2, da2)
    assert torch.equal(top2, top) and torch.equal(dq2, dq) and torch.equal(da2, da)
    # reference mode: the bits
    capi.set_euclid_backward_mode("reference")
    assert capi.get_euclid_backward_mode() == "reference"
    capi.simcross_forward_backward(1, qd, ad, dTd, top, dq, da)
    assert (dq.cpu().numpy().view(np.uint32) == dq_ref.view(np.uint32)).all()
    assert (da.cpu().numpy().view(np.uint32) == da_ref.view(np.uint32)).all()


def test_fp32_mode_near_identical_pairs_and_extreme_scales(oracle, hiplib):
    """T -> 1 (den -> 1e-9) and tiny / huge coordinates: still within 2 ulp of the reference
    expression wherever the reference value is a normal number."""
    r = np.random.default_rng(77)
    N, D = 64, 300
    q, a, dT = _case(r, N, D)
    a[:16] = q[:16]                                    # distance 0: T == 1, den == 1e-9
    a[16:32] = q[16:32] + np.float32(1e-6) * r.standard_normal((16, 1, D)).astype(np.float32)
    q[32:40] *= np.float32(1e-12)
    a[32:40] *= np.float32(1e-12)
    q[40:48] *= np.float32(1e6)
    a[40:48] *= np.float32(1e6)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, _, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    qd, ad, dTd = (torch.from_numpy(x).cuda() for x in (q, a, dT))
    capi.set_euclid_backward_mode("fp32")
    top = torch.empty(N, 1, 1, 1, device="cuda")
    dq = torch.empty_like(qd)
    da = torch.empty_like(ad)
    capi.simcross_forward_backward(1, qd, ad, dTd, top, dq, da)
    g = dq.cpu().numpy()
    assert (top.cpu().numpy().view(np.uint32) == top_ref.view(np.uint32)).all()
    normal = np.abs(dq_ref) >= np.float32(1.2e-38)
    assert _ulps(g, dq_ref)[normal].max() <= 2
    if (~normal).any():                                # subnormal / zero reference values: a few quanta at most
        assert np.abs(g[~normal].astype(np.float64) - dq_ref[~normal]).max() <= 1e-44


@pytest.mark.parametrize("N,D", [(4096, 300), (13, 300), (257, 200), (64, 100)])
def test_triplet_step_fp32_mode(N, D, oracle, hiplib):
    """Fused (q, a+, a-) step in the default arithmetic: scores exact, gradients within 3 ulp of
    the layer-by-layer oracle (dq is the sum of two branch terms, each within 1-2 ulp; compared
    against the magnitude of the larger branch term, since the sum may cancel)."""
    r = np.random.default_rng(N + D)
    q = (r.standard_normal((N, 1, D)) * 0.4).astype(np.float32)
    ap = (q + 0.04 * r.standard_normal((N, 1, D))).astype(np.float32)
    an = (r.standard_normal((N, 1, D)) * 0.4).astype(np.float32)
    y = (r.uniform(size=(N, 1)) < 0.8).astype(np.float32)
    margin, lw = 0.05, 1.0
    sp, _, _ = oracle.simcross_forward(1, q, ap)
    sn, _, _ = oracle.simcross_forward(1, q, an)
    loss_ref, o, s = oracle.pairrank_forward(sp.reshape(N, 1), sn.reshape(N, 1), y, margin)
    gsp, gsn = oracle.pairrank_backward(y, o, s, top_diff=lw)
    dq_p, dap_ref, _, _ = oracle.simcross_backward(1, q, ap, sp, gsp.reshape(sp.shape))
    dq_n, dan_ref, _, _ = oracle.simcross_backward(1, q, an, sn, gsn.reshape(sn.shape))
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    out = dict(s_pos=torch.empty(N, 1, device="cuda"), s_neg=torch.empty(N, 1, device="cuda"),
               loss=torch.empty(1, device="cuda"), dq=torch.empty(N, 1, D, device="cuda"),
               da_pos=torch.empty(N, 1, D, device="cuda"), da_neg=torch.empty(N, 1, D, device="cuda"))
    capi.set_euclid_backward_mode("fp32")
    capi.triplet_euclid_step(d(q), d(ap), d(an), d(y), margin=margin, loss_weight=lw, **out)
    h = {k: v.cpu().numpy() for k, v in out.items()}
    assert (h["s_pos"].ravel().view(np.uint32) == sp.ravel().view(np.uint32)).all()
    assert (h["s_neg"].ravel().view(np.uint32) == sn.ravel().view(np.uint32)).all()
    assert abs(h["loss"][0] - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
    assert _ulps(h["da_pos"], dap_ref).max() <= 2 and _ulps(h["da_neg"], dan_ref).max() <= 2
    scale = np.maximum(np.abs(dq_p), np.abs(dq_n))
    assert (np.abs(h["dq"] - (dq_p + dq_n)) <= 4 * np.spacing(scale)).all()


@pytest.mark.parametrize("shape", [(50, 40, 40, 50), (7, 40, 40, 300), (3, 5, 9, 33),
                                   (600, 40, 40, 50), (513, 7, 8, 33), (520, 47, 48, 64), (515, 9, 20, 50),
                                   (530, 33, 40, 50), (600, 1, 8, 16)])   # lane-per-column kernel (two waves per pair in fp32 mode: odd W1, W1 = 1)
def test_cross_geometry_fp32_mode(shape, oracle, hiplib):
    """Word-grid geometry (W1 x W2 scores per pair): scores exact; each gradient element is a sum of
    W terms that are each within 2 ulp, so it is held to the north-star bar against the largest
    magnitude, and to 4 ulp of the sum of the terms' magnitudes."""
    N, W1, W2, D = shape
    r = np.random.default_rng(sum(shape))
    q = (r.standard_normal((N, W1, D)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N, W2, D)) * 0.4).astype(np.float32)
    dT = r.standard_normal((N, 1, W1, W2)).astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    # magnitude budget: sum_k |tt| in float64
    T = top_ref.astype(np.float64)[:, 0]
    coef = np.abs(dT.astype(np.float64)[:, 0] * T ** 3 / (T - 1 + 1e-9))          # (N, W1, W2)
    diff = np.abs(q.astype(np.float64)[:, :, None, :] - a.astype(np.float64)[:, None, :, :])
    mag_q = (coef[..., None] * diff).sum(axis=2)
    mag_a = (coef[..., None] * diff).sum(axis=1)
    qd, ad, dTd = (torch.from_numpy(x).cuda() for x in (q, a, dT))
    capi.set_euclid_backward_mode("fp32")
    top = torch.empty(N, 1, W1, W2, device="cuda")
    capi.simcross_forward(1, qd, ad, top)
    assert (top.cpu().numpy().view(np.uint32) == top_ref.view(np.uint32)).all()
    dq, da = torch.empty_like(qd), torch.empty_like(ad)
    capi.simcross_backward(1, qd, ad, top, dTd, dq, da)
    g, h = dq.cpu().numpy(), da.cpu().numpy()
    eps = np.finfo(np.float32).eps
    assert (np.abs(g - dq_ref) <= 4 * eps * mag_q + 1e-30).all()
    assert (np.abs(h - da_ref) <= 4 * eps * mag_a + 1e-30).all()
    np.testing.assert_allclose(g, dq_ref, rtol=0, atol=1e-5 * max(1.0, np.abs(dq_ref).max()))
    capi.set_euclid_backward_mode("reference")
    capi.simcross_backward(1, qd, ad, top, dTd, dq, da)
    assert (dq.cpu().numpy().view(np.uint32) == dq_ref.view(np.uint32)).all()
    assert (da.cpu().numpy().view(np.uint32) == da_ref.view(np.uint32)).all()


def test_mode_setter_validates(hiplib):
    assert hiplib.mms_set_euclid_backward_mode(7) == 1            # MMS_ERR_INVALID_ARG
    capi.set_euclid_backward_mode("reference")
    assert hiplib.mms_get_euclid_backward_mode() == 1
    capi.set_euclid_backward_mode("fp32")
    assert hiplib.mms_get_euclid_backward_mode() == 0
