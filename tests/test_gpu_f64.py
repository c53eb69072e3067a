"""The double instantiation of the C ABI (include/mms.h *_f64, csrc/f64_paths.hip) against the
oracle's double instantiation (the reference instantiates float and double, common.hpp:41-44).
Order-defined results are compared bit for bit; BLAS-backed ones at 1e-12 relative."""
import numpy as np
import pytest
import torch

from mms_answer_selection_amd import capi

pytestmark = pytest.mark.gpu
RT = 1e-12


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(t):
    return t.cpu().numpy()


def bits(x):
    return np.ascontiguousarray(x).view(np.uint64)


def close(x, ref, name=""):
    scale = max(1.0, float(np.abs(ref).max())) if ref.size else 1.0
    assert np.abs(x - ref).max() <= RT * scale, name


def nanlike(shape):
    return torch.full(shape, float("nan"), dtype=torch.float64, device="cuda")


@pytest.mark.parametrize("shape", [(64, 1, 1, 300), (5, 3, 4, 7), (2, 40, 40, 50), (1, 1, 1, 1),
                                   (67, 1, 1, 301), (5, 1, 1, 15), (9, 1, 1, 48), (130, 1, 1, 1024)])   # tiled rows kernel: odd D (rows of alternating alignment), no whole tile, whole tiles only
def test_euclid_f64_bitexact(shape, oracle, hiplib):
    N, W1, W2, D = shape
    r = np.random.default_rng(sum(shape))
    q = r.standard_normal((N, W1, D)) * 0.4
    a = r.standard_normal((N, W2, D)) * 0.4
    dT = r.standard_normal((N, 1, W1, W2))
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    top = nanlike((N, 1, W1, W2))
    capi.simcross_forward_f64(1, dev(q), dev(a), top)
    assert (bits(host(top)) == bits(top_ref)).all()
    dq, da = nanlike(q.shape), nanlike(a.shape)
    capi.simcross_backward_f64(1, dev(q), dev(a), top, dev(dT), dq, da)
    assert (bits(host(dq)) == bits(dq_ref)).all() and (bits(host(da)) == bits(da_ref)).all()
    # :176-177: no propagate_down at all still zeroes both diffs
    capi.simcross_backward_f64(1, dev(q), dev(a), top, dev(dT), dq, da, propagate_down=(False, False))
    assert (host(dq) == 0).all() and (host(da) == 0).all()


@pytest.mark.parametrize("shape", [(33, 1, 1, 300), (4, 5, 3, 16), (70, 1, 1, 37)])
def test_cosine_f64(shape, oracle, hiplib):
    N, W1, W2, D = shape
    r = np.random.default_rng(7 + sum(shape))
    q = r.standard_normal((N, W1, D))
    a = r.standard_normal((N, W2, D))
    dT = r.standard_normal((N, 1, W1, W2))
    top_ref, n0_ref, n1_ref = oracle.simcross_forward(0, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(0, q, a, top_ref, dT, norm0=n0_ref, norm1=n1_ref)
    top, n0, n1 = nanlike((N, 1, W1, W2)), nanlike((N, W1)), nanlike((N, W2))
    capi.simcross_forward_f64(0, dev(q), dev(a), top, norm0=n0, norm1=n1)
    close(host(top), top_ref, "top")
    close(host(n0), n0_ref)
    close(host(n1), n1_ref)
    dq, da = nanlike(q.shape), nanlike(a.shape)
    capi.simcross_backward_f64(0, dev(q), dev(a), top, dev(dT), dq, da, norm0=n0, norm1=n1)
    close(host(dq), dq_ref, "dq")
    close(host(da), da_ref, "da")


@pytest.mark.parametrize("cfg", [(6, 5, 4, 12, 3, True), (3, 1, 1, 20, 1, False), (2, 7, 7, 9, 4, True),
                                 (64, 40, 33, 50, 3, True),      # several MFMA tiles, ragged edges; dW on the long-K kernel
                                 (5, 70, 65, 67, 2, False)])     # more than one 64 x 64 tile per product
def test_bilinear_f64(cfg, oracle, hiplib):
    N, W1, W2, D, M, bias_term = cfg
    r = np.random.default_rng(11 + sum(cfg[:5]))
    q = r.standard_normal((N, W1, D)) * 0.5
    a = r.standard_normal((N, W2, D)) * 0.5
    W = r.standard_normal((M, D, D)) * 0.2
    bias = r.standard_normal((M, W1, W2)) if bias_term else None
    dT = r.standard_normal((N, M, W1, W2))
    top_ref, _, _ = oracle.simcross_forward(2, q, a, W, bias)
    db_in = np.full((M, W1, W2), 0.25) if bias_term else None
    dq_ref, da_ref, dW_ref, db_ref = oracle.simcross_backward(2, q, a, top_ref, dT, W=W, bias_term=bias_term,
                                                              dbias_in=db_in)
    top = nanlike((N, M, W1, W2))
    capi.simcross_forward_f64(2, dev(q), dev(a), top, W=dev(W), bias=dev(bias) if bias_term else None)
    close(host(top), top_ref, "top")
    dq, da, dW = nanlike(q.shape), nanlike(a.shape), nanlike(W.shape)
    db = dev(db_in) if bias_term else None
    capi.simcross_backward_f64(2, dev(q), dev(a), dev(top_ref), dev(dT), dq, da, W=dev(W), bias_term=bias_term,
                               dW=dW, dbias=db)
    close(host(dq), dq_ref, "dq")
    close(host(da), da_ref, "da")
    close(host(dW), dW_ref, "dW")
    if bias_term:
        assert (bits(host(db)) == bits(db_ref)).all()          # n-ascending column sum: order-defined


@pytest.mark.parametrize("dims", [(37, 24, 19), (3000, 130, 70)])   # the second: many tiles, dW on the long-K kernel
def test_simmatrix_f64(dims, oracle, hiplib):
    r = np.random.default_rng(5)
    N, K1, K2 = dims
    q = r.standard_normal((N, K1))
    a = r.standard_normal((N, K2))
    W = r.standard_normal((K1, K2)) * 0.3
    dT = r.standard_normal((N, 1))
    top_ref, scr_ref = oracle.simmatrix_forward(q, a, W)
    dW_in = r.standard_normal((K1, K2))
    dq_ref, da_ref, dW_ref = oracle.simmatrix_backward(q, a, W, dT, dW_in=dW_in)
    top, scr = nanlike((N, 1)), nanlike((N, K2))
    capi.simmatrix_forward_f64(dev(q), dev(a), dev(W), top, scr)
    close(host(top), top_ref)
    close(host(scr), scr_ref)
    dq, da, dW = nanlike((N, K1)), nanlike((N, K2)), dev(dW_in)
    capi.simmatrix_backward_f64(dev(q), dev(a), dev(W), dev(dT), dq, da, dW)
    close(host(dq), dq_ref)
    close(host(da), da_ref)
    close(host(dW), dW_ref)
    # propagate flags: untouched outputs stay untouched
    dq2, dW2 = nanlike((N, K1)), dev(dW_in)
    capi.simmatrix_backward_f64(dev(q), dev(a), dev(W), dev(dT), dq2, None, dW2, param_propagate_down=False,
                                propagate_down=(True, False))
    close(host(dq2), dq_ref)
    assert (host(dW2) == dW_in).all()


@pytest.mark.parametrize("count", [1, 257, 4096])
def test_pairrank_f64_bitexact_including_loss(count, oracle, hiplib):
    r = np.random.default_rng(count)
    a = r.standard_normal((count, 1))
    b = r.standard_normal((count, 1))
    y = (r.uniform(size=(count, 1)) < 0.7).astype(np.float64)
    loss_ref, o_ref, s_ref = oracle.pairrank_forward(a, b, y, 0.3)
    da_ref, db_ref = oracle.pairrank_backward(y, o_ref, s_ref, top_diff=1.5)
    o, s, loss = nanlike((count, 1)), nanlike((count, 1)), nanlike((1,))
    capi.pairrank_forward_f64(dev(a), dev(b), dev(y), o, s, loss, margin=0.3)
    assert (bits(host(o)) == bits(o_ref)).all() and (bits(host(s)) == bits(s_ref)).all()
    assert bits(host(loss))[0] == bits(np.array([loss_ref]))[0]     # sequential sum reproduced
    da, db = nanlike((count, 1)), nanlike((count, 1))
    capi.pairrank_backward_f64(y=dev(y), ordered=o, similar=s, da=da, db=db, top_diff=1.5)
    assert (bits(host(da)) == bits(da_ref)).all() and (bits(host(db)) == bits(db_ref)).all()


def test_f64_argument_checks(hiplib):
    assert hiplib.mms_simcross_forward_f64(1, 4, 1, 1, 8, 1, None, None, None, None, None, None, None, None, 0, None) == 1
    assert hiplib.mms_pairrank_forward_f64(0, 1.0, None, None, None, None, None, None, None) == 1
    q = torch.zeros(2, 1, 4, dtype=torch.float64, device="cuda")
    top = torch.zeros(2, 2, 1, 1, dtype=torch.float64, device="cuda")
    W = torch.zeros(2, 4, 4, dtype=torch.float64, device="cuda")
    with pytest.raises(capi.MMSError, match="workspace"):
        capi.check(hiplib.mms_simcross_forward_f64(2, 2, 1, 1, 4, 2, q.data_ptr(), q.data_ptr(), W.data_ptr(), None,
                                                   top.data_ptr(), None, None, None, 0, None), "fwd")
