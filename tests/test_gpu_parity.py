"""GPU parity: the HIP path, called through the C ABI (include/mms.h), against the
CPU oracle (oracle/) on the same seeded inputs.

Bars (DESIGN.md "Numerics"):
  * bit-exact where the reference's own source fixes the evaluation order:
    Euclidean SimCross forward/backward (sim_cross_layer.cpp:96-111, 208-225),
    PairRankLoss cached terms and gradients (pair_rank_loss_layer.cpp:28-37, 61-82),
    dbias (sim_cross_layer.cpp:301-304);
  * TOL = 1e-5 (north_star) where the reference goes through CBLAS (cosine,
    bilinear, SimMatrix) or a long scalar running sum (the loss value).
"""
import numpy as np
import pytest
import torch

from util import TOL, assert_bitexact, assert_close, qa, rng

pytestmark = pytest.mark.gpu


def dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def nan_like(shape):
    return torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")


# --------------------------------------------------------------------------- #
# SimCross Euclidean (dist_mode 1): bit-exact
# --------------------------------------------------------------------------- #
EUCLID_SHAPES = [
    (8, 1, 1, 300),      # golden-vector shape
    (4096, 1, 1, 300),   # BASELINE cfg 2, full size
    (37, 1, 1, 301),     # D % 4 != 0 -> scalar path, ragged last workgroup
    (5, 1, 1, 4),
    (1, 1, 1, 1),
    (4, 5, 7, 300),      # W1 != W2
    (2, 40, 40, 50),     # the reference's network_v4 geometry (do_trec_qa_clean.py:468)
    (3, 41, 9, 33),      # tile remainders
    (2, 1, 6, 20),
    (33, 1, 1, 1024),    # BASELINE cfg 5 width: one pair per wave, +-31 ulp window
    (9, 1, 1, 400),      # last D with two pairs per wave
    (9, 1, 1, 404),      # first D with one pair per wave
    (3, 1, 1, 1028),     # beyond the wave kernel: generic rows kernel
    (1, 70, 60, 9),      # W1*W2 tables exceed LDS: generic cross backward
    (2, 1, 1, 2100),     # rows wider than every rows kernel: cross kernels with W = 1
    (600, 40, 40, 50),   # many narrow word grids: lane-per-column backward (cross_bwd_lane_kernel), W2 even
    (513, 7, 8, 33),     # ... narrowest instantiated width
    (520, 47, 48, 64),   # ... its largest geometry
    (515, 5, 24, 10),
    (530, 20, 20, 50),   # ... small enough for its reference-rounding variant
]


def _assert_grad(got, ref, what, mode, one_word):
    """reference mode: the reference's bits.  fp32 mode (the product default, what bench.py times): scores are still
    bit-exact; a gradient element of the one-word geometry is ONE term and stays within 2 ulp of the reference's
    (normal numbers); in a word grid it is a sum of W terms each that close, held to the north-star form
    1e-5 * max(1, max|ref|)."""
    if mode == "reference":
        assert_bitexact(got, ref, what)
        return
    if one_word:
        normal = np.abs(ref) >= np.float32(1.2e-38)
        ulps = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
        assert not normal.any() or ulps[normal].max() <= 2, "%s: %d ulp from the reference in fp32 mode" % (what, ulps[normal].max())
        if (~normal).any():
            assert np.abs(got[~normal].astype(np.float64) - ref[~normal]).max() <= 1e-37, what
    assert_close(got, ref, TOL, what)


@pytest.mark.parametrize("bwd_mode", ["reference", "fp32"])
@pytest.mark.parametrize("shape", EUCLID_SHAPES)
def test_euclid_forward_backward_bitexact(shape, bwd_mode, oracle, hiplib):
    from mms_answer_selection_amd import capi
    capi.set_euclid_backward_mode(bwd_mode)          # (the autouse fixture restores the default afterwards)
    N, W1, W2, D = shape
    one_word = W1 == 1 and W2 == 1
    r = rng(sum(shape))
    q, a = qa(r, N, W1, W2, D)
    if N >= 4:
        a[1, 0] = q[1, 0]                       # degenerate pair: T = 1, divisor 1e-9
        a[2, W2 - 1, : D // 2] = q[2, W1 - 1, : D // 2]
    dT = r.standard_normal((N, 1, W1, W2)).astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)

    dq_, da_, dT_ = dev(q), dev(a), dev(dT)
    top = nan_like(top_ref.shape)
    capi.simcross_forward(1, dq_, da_, top)
    assert_bitexact(host(top), top_ref, "top")

    gq, ga = nan_like(q.shape), nan_like(a.shape)
    capi.simcross_backward(1, dq_, da_, top, dT_, gq, ga)
    _assert_grad(host(gq), dq_ref, "dq", bwd_mode, one_word)
    _assert_grad(host(ga), da_ref, "da", bwd_mode, one_word)

    # one-launch forward+backward gives the same bits as the two launches, in either mode
    top2, gq2, ga2 = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
    capi.simcross_forward_backward(1, dq_, da_, dT_, top2, gq2, ga2)
    assert_bitexact(host(top2), top_ref, "fused top")
    _assert_grad(host(gq2), dq_ref, "fused dq", bwd_mode, one_word)
    _assert_grad(host(ga2), da_ref, "fused da", bwd_mode, one_word)
    if one_word:
        assert_bitexact(host(gq2), host(gq), "fused dq == Backward launch's")
        assert_bitexact(host(ga2), host(ga), "fused da == Backward launch's")


def test_euclid_shape_fuzz_bitexact(oracle, hiplib):
    """Many small random geometries through every dispatch branch (wave kernel with 1 or 2
    pairs per wave and NIT 1..4, generic rows kernel, cross tiles of every register-tile size,
    tiled and generic cross backward): everything Euclidean stays bit-exact."""
    from mms_answer_selection_amd import capi
    r = rng(2024)
    dims = [1, 2, 3, 4, 5, 8, 12, 16, 20, 36, 44, 52, 64, 100, 128, 132, 256, 300, 404, 512, 768, 1020, 1024]
    shapes = []
    for _ in range(28):
        D = int(r.choice(dims))
        if r.uniform() < 0.5:
            shapes.append((int(r.integers(1, 70)), 1, 1, D))
        else:
            shapes.append((int(r.integers(1, 6)), int(r.integers(1, 48)), int(r.integers(1, 48)), min(D, 132)))
    for (N, W1, W2, D) in shapes:
        q, a = qa(r, N, W1, W2, D)
        dT = r.standard_normal((N, 1, W1, W2)).astype(np.float32)
        top_ref, _, _ = oracle.simcross_forward(1, q, a)
        dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
        qd, ad, dTd = dev(q), dev(a), dev(dT)
        top, gq, ga = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
        capi.simcross_forward_backward(1, qd, ad, dTd, top, gq, ga)
        what = "shape %s" % ((N, W1, W2, D),)
        assert_bitexact(host(top), top_ref, what + " top")
        assert_bitexact(host(gq), dq_ref, what + " dq")
        assert_bitexact(host(ga), da_ref, what + " da")
        top2, gq2, ga2 = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
        capi.simcross_forward(1, qd, ad, top2)
        capi.simcross_backward(1, qd, ad, top2, dTd, gq2, ga2)
        assert_bitexact(host(top2), top_ref, what + " top (two calls)")
        assert_bitexact(host(gq2), dq_ref, what + " dq (two calls)")
        assert_bitexact(host(ga2), da_ref, what + " da (two calls)")


@pytest.mark.parametrize("D", [200, 300, 1024])
def test_euclid_speculation_miss_falls_back_exactly(D, oracle, hiplib):
    """Adversarial rows for the speculative chain (euclid_math.h): one large square
    followed by squares below half an ulp of the running sum.  The sequential fp32
    sum swallows every small term while a tree sum keeps them, so the prediction is
    tens of ulps off, the candidate window misses and the kernel must re-walk the
    segment -- still bit-identical to the reference order."""
    from mms_answer_selection_amd import capi
    N = 6
    q = np.zeros((N, 1, D), np.float32)
    a = np.zeros((N, 1, D), np.float32)
    tiny = np.float32(2.0 ** -12.5)
    q[:, 0, 0] = 1.0
    q[0:2, 0, 1:] = tiny                    # small terms in both segments
    q[2:4, 0, 1:D // 2] = tiny              # only in segment 0
    q[4, 0, 1:] = -tiny
    a[5, 0, :] = rng(3).standard_normal(D).astype(np.float32)   # an ordinary row alongside
    dT = rng(4).standard_normal((N, 1, 1, 1)).astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    # the construction really defeats the prediction: over the first segment (a third of the
    # row) tree and sequential sums differ by more than the candidate window (12 / 25 ulp)
    sq = ((q - a)[0, 0, : ((D // 4 + 2) // 3) * 4] ** 2).astype(np.float32)
    seq = np.float32(0)
    for v in sq:
        seq = np.float32(seq + v)
    tree = np.float32(sq.astype(np.float64).sum())
    assert abs(int(tree.view(np.int32)) - int(seq.view(np.int32))) > (25 if D > 400 else 12)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    top, gq, ga = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
    capi.simcross_forward_backward(1, dev(q), dev(a), dev(dT), top, gq, ga)
    assert_bitexact(host(top), top_ref, "top")
    assert_bitexact(host(gq), dq_ref, "dq")
    assert_bitexact(host(ga), da_ref, "da")
    # the fused triplet step shares the scheme
    y = np.ones((N, 1), np.float32)
    sp, _, _ = oracle.simcross_forward(1, q, a)
    out = dict(s_pos=nan_like((N, 1)), s_neg=nan_like((N, 1)), loss=nan_like((1,)),
               dq=nan_like(q.shape), da_pos=nan_like(q.shape), da_neg=nan_like(q.shape))
    capi.triplet_euclid_step(dev(q), dev(a), dev(a[::-1].copy()), dev(y), **out)
    assert_bitexact(host(out["s_pos"]).ravel(), sp.ravel(), "triplet s_pos")
    sn, _, _ = oracle.simcross_forward(1, q, a[::-1].copy())
    assert_bitexact(host(out["s_neg"]).ravel(), sn.ravel(), "triplet s_neg")


@pytest.mark.parametrize("D", [300, 200, 52])
def test_euclid_non_finite_inputs(D, oracle, hiplib):
    """Inf / NaN / huge coordinates: the speculative stitch must not turn them into something else
    (a window miss falls back to the sequential walk).  Compared with the oracle; NaNs match NaNs."""
    from mms_answer_selection_amd import capi
    N = 8
    r = rng(D)
    q, a = qa(r, N, 1, 1, D)
    q[0, 0, 5] = np.inf                       # dist = inf -> T = 0
    q[1, 0, D - 1] = np.nan                   # dist = NaN -> T = NaN
    q[2, 0, :] = 3e19                         # squares overflow to inf half-way through
    a[3, 0, 7] = -np.inf
    q[4, 0, 0] = 1e19
    a[4, 0, 0] = -1e19                        # a single huge square, finite sum
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    with np.errstate(all="ignore"):
        top_ref, _, _ = oracle.simcross_forward(1, q, a)
        dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    top, gq, ga = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
    capi.simcross_forward_backward(1, dev(q), dev(a), dev(dT), top, gq, ga)
    assert_bitexact(host(top), top_ref, "top")
    assert_bitexact(host(gq), dq_ref, "dq")
    assert_bitexact(host(ga), da_ref, "da")


def test_euclid_subnormal_squares(oracle, hiplib):
    """Squares in the fp32 subnormal range: the packed adds of the chain must not flush."""
    from mms_answer_selection_amd import capi
    N, D = 8, 300
    r = rng(77)
    q, a = qa(r, N, 1, 1, D)
    q *= np.float32(1e-20)
    a *= np.float32(1e-20)
    q[4:] *= np.float32(1e3)                 # mixes subnormal and tiny normal squares
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    sq = ((q - a) ** 2).astype(np.float32)
    assert (sq[0] < 1.1754944e-38).all() and (sq[0] > 0).any()
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    top, gq, ga = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
    capi.simcross_forward_backward(1, dev(q), dev(a), dev(dT), top, gq, ga)
    assert_bitexact(host(top), top_ref, "top")
    assert_bitexact(host(gq), dq_ref, "dq")
    assert_bitexact(host(ga), da_ref, "da")


@pytest.mark.parametrize("cfg", [(8192, 1024), (33, 1024), (5, 8), (17, 2048), (9, 304), (21, 520), (5, 408), (1, 1024), (4099, 512)])
def test_euclid_fp16_storage(cfg, oracle, hiplib):
    """BASELINE cfg 5: half in HBM, fp32 reference arithmetic.  Against the fp32 oracle run on
    the widened inputs: scores bit-exact, gradients equal to the oracle's rounded to half."""
    from mms_answer_selection_amd import capi
    N, D = cfg
    r = rng(N + D)
    qf, af = qa(r, N, 1, 1, D)
    qh, ah = qf.astype(np.float16), af.astype(np.float16)
    if N > 4:
        ah[1] = qh[1]
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    q32, a32 = qh.astype(np.float32), ah.astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q32, a32)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q32, a32, top_ref, dT)
    qd = torch.from_numpy(qh).cuda()
    ad = torch.from_numpy(ah).cuda()
    top = nan_like(top_ref.shape)
    capi.simcross_euclid_forward_f16(qd, ad, top)
    assert_bitexact(host(top), top_ref, "top (forward only)")
    top2 = nan_like(top_ref.shape)
    gq = torch.full(qh.shape, float("nan"), dtype=torch.float16, device="cuda")
    ga = torch.full(qh.shape, float("nan"), dtype=torch.float16, device="cuda")
    capi.simcross_euclid_forward_backward_f16(qd, ad, dev(dT), top2, gq, ga)
    assert_bitexact(host(top2), top_ref, "top")
    assert (host(gq).view(np.uint16) == dq_ref.astype(np.float16).view(np.uint16)).all(), "dq halves"
    assert (host(ga).view(np.uint16) == da_ref.astype(np.float16).view(np.uint16)).all(), "da halves"


@pytest.mark.parametrize("cfg", [(8192, 1024), (33, 1024), (7, 2048), (5, 8), (9, 304), (1, 512)])
def test_cosine_fp16_storage(cfg, oracle, hiplib):
    """fp16-STORAGE cosine sentence vectors (cfg 5's dtype with dist_mode 0): fp32 arithmetic on the widened halves,
    against the fp32 oracle run on the same fp16-rounded inputs.  Scores and cached norms within 1e-5 (the dot
    products are cblas_sdot in the reference, no defined order); gradients are stored as halves, so they are held to
    the configuration's 1e-3 bar and, tighter, to one half ulp-pair of the fp32 gradient."""
    from mms_answer_selection_amd import capi
    N, D = cfg
    r = rng(11 * N + D)
    qf, af = qa(r, N, 1, 1, D)
    qh, ah = qf.astype(np.float16), af.astype(np.float16)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    q32, a32 = qh.astype(np.float32), ah.astype(np.float32)
    top_ref, n0_ref, n1_ref = oracle.simcross_forward(0, q32, a32)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(0, q32, a32, top_ref, dT, norm0=n0_ref, norm1=n1_ref)
    qd, ad = torch.from_numpy(qh).cuda(), torch.from_numpy(ah).cuda()
    mk16 = lambda: torch.full(qh.shape, float("nan"), dtype=torch.float16, device="cuda")
    top, n0, n1 = nan_like(top_ref.shape), nan_like(n0_ref.shape), nan_like(n1_ref.shape)
    capi.simcross_cosine_forward_f16(qd, ad, top, n0, n1)
    assert_close(host(top), top_ref, 1e-5, "top (forward only)")
    assert_close(host(n0), n0_ref, 1e-5, "norm0")
    assert_close(host(n1), n1_ref, 1e-5, "norm1")
    top2, gq, ga = nan_like(top_ref.shape), mk16(), mk16()
    capi.simcross_cosine_forward_backward_f16(qd, ad, dev(dT), top2, gq, ga)      # norms optional
    assert_bitexact(host(top2), host(top), "fused top == forward-only top")
    for got, ref, what in ((gq, dq_ref, "dq"), (ga, da_ref, "da")):
        g = host(got).astype(np.float32)
        assert_close(g, ref, 1e-3, what)
        # half rounding of a value within 1e-5 of the fp32 gradient: <= 2^-11 relative per element + the 1e-5 slack
        assert (np.abs(g - ref) <= np.abs(ref) * 2.0 ** -10 + 1e-5 * max(1.0, np.abs(ref).max()) + 2.0 ** -24).all(), what
    top3, gq3, ga3 = nan_like(top_ref.shape), mk16(), mk16()
    capi.simcross_cosine_forward_backward_f16(qd, ad, dev(dT), top3, gq3, ga3)
    assert (host(gq3).view(np.uint16) == host(gq).view(np.uint16)).all(), "deterministic"
    assert (host(ga3).view(np.uint16) == host(ga).view(np.uint16)).all(), "deterministic"


def test_cosine_fp16_storage_refuses_what_it_cannot_do(hiplib):
    from mms_answer_selection_amd import capi
    q = torch.zeros((4, 1, 12), dtype=torch.float16, device="cuda")           # D % 8 != 0
    top = torch.zeros((4, 1, 1, 1), device="cuda")
    with pytest.raises(capi.MMSError):
        capi.simcross_cosine_forward_f16(q, q, top)
    q = torch.zeros((4, 1, 4096), dtype=torch.float16, device="cuda")          # D > 2048
    with pytest.raises(capi.MMSError):
        capi.simcross_cosine_forward_f16(q, q, top)
    capi.simcross_cosine_forward_f16(q[:0], q[:0], top[:0])                    # N == 0 is a no-op


@pytest.mark.parametrize("cfg", [(8192, 1024), (33, 1024), (17, 2048), (9, 304), (64, 512)])
def test_euclid_fp16_storage_tree_mode(cfg, oracle, hiplib):
    """The opt-in tree sum of the fp16-storage path (no ordered chain): SURVEY 8(d) holds cfg 5 to 1e-3 relative
    against the fp32 oracle on the fp16-rounded inputs; the tree sum is within 1e-5 of it, deterministic, and the
    default (ordered) mode is back to bit-exactness afterwards."""
    from mms_answer_selection_amd import capi
    N, D = cfg
    r = rng(N + 3 * D)
    qf, af = qa(r, N, 1, 1, D)
    qh, ah = qf.astype(np.float16), af.astype(np.float16)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    q32, a32 = qh.astype(np.float32), ah.astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q32, a32)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q32, a32, top_ref, dT)
    qd, ad = torch.from_numpy(qh).cuda(), torch.from_numpy(ah).cuda()
    mk16 = lambda: torch.full(qh.shape, float("nan"), dtype=torch.float16, device="cuda")
    capi.set_f16_distance_mode("tree")
    try:
        top, gq, ga = nan_like(top_ref.shape), mk16(), mk16()
        capi.simcross_euclid_forward_backward_f16(qd, ad, dev(dT), top, gq, ga)
        top_b, gq_b, ga_b = nan_like(top_ref.shape), mk16(), mk16()
        capi.simcross_euclid_forward_backward_f16(qd, ad, dev(dT), top_b, gq_b, ga_b)
        fwd = nan_like(top_ref.shape)
        capi.simcross_euclid_forward_f16(qd, ad, fwd)
    finally:
        capi.set_f16_distance_mode("ordered")
    assert_close(host(top), top_ref, 1e-5, "top (tree sum)")           # bar for this configuration: 1e-3
    assert_bitexact(host(top_b), host(top), "deterministic")
    assert_bitexact(host(fwd), host(top), "forward-only == fused")
    assert (host(gq_b).view(np.uint16) == host(gq).view(np.uint16)).all()
    g, ref = host(gq).astype(np.float32), dq_ref
    assert np.abs(g - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max()), "dq within the fp16 grid of the oracle's"
    g, ref = host(ga).astype(np.float32), da_ref
    assert np.abs(g - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max())
    top2 = nan_like(top_ref.shape)
    capi.simcross_euclid_forward_f16(qd, ad, top2)
    assert_bitexact(host(top2), top_ref, "ordered mode is the default again")


def test_euclid_unaligned_views_take_scalar_path(oracle, hiplib):
    """Pointers that are not 16-byte aligned must still give the same bits."""
    from mms_answer_selection_amd import capi
    N, D = 19, 300
    r = rng(5)
    q, a = qa(r, N, 1, 1, D)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)

    def shifted(x):
        buf = torch.empty(x.size + 1, dtype=torch.float32, device="cuda")
        v = buf[1:].view(x.shape)
        v.copy_(torch.from_numpy(x))
        assert v.data_ptr() % 16 != 0
        return v

    qd, ad = shifted(q), shifted(a)
    top, gq, ga = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
    capi.simcross_forward_backward(1, qd, ad, dev(dT), top, gq, ga)
    assert_bitexact(host(top), top_ref)
    assert_bitexact(host(gq), dq_ref)
    assert_bitexact(host(ga), da_ref)


def test_backward_without_propagate_down_zeroes_both(hiplib):
    """sim_cross_layer.cpp:176-177,201."""
    from mms_answer_selection_amd import capi
    N, W1, W2, D = 3, 2, 3, 8
    q, a = qa(rng(), N, W1, W2, D)
    top = torch.rand(N, 1, W1, W2, device="cuda")
    gq, ga = nan_like(q.shape), nan_like(a.shape)
    capi.simcross_backward(1, dev(q), dev(a), top, torch.ones_like(top), gq, ga,
                           propagate_down=(False, False))
    assert (host(gq) == 0).all() and (host(ga) == 0).all()


@pytest.mark.parametrize("shape", [(1517, 40, 40, 50), (1025, 16, 24, 50), (1031, 8, 40, 50), (1024, 40, 8, 50), (1030, 40, 40, 52)])
def test_euclid_pair_image_forward_bitexact(shape, oracle, hiplib):
    """Forward-only scoring of many pairs with small D (BASELINE cfg 4: the 1517 TREC-QA test
    candidates at 40 x 40 x 50): the whole-pair LDS image kernel, ragged last workgroup included.
    Cosine goes through the same kernel: 1e-5."""
    from mms_answer_selection_amd import capi
    N, W1, W2, D = shape
    r = rng(sum(shape) + 9)
    q, a = qa(r, N, W1, W2, D)
    a[1, 0] = q[1, 0]
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    qd, ad = dev(q), dev(a)
    top = nan_like(top_ref.shape)
    capi.simcross_forward(1, qd, ad, top)
    assert_bitexact(host(top), top_ref, "top")
    ctop_ref, n0_ref, n1_ref = oracle.simcross_forward(0, q, a)
    ctop, n0, n1 = nan_like(top_ref.shape), nan_like(n0_ref.shape), nan_like(n1_ref.shape)
    capi.simcross_forward(0, qd, ad, ctop, norm0=n0, norm1=n1)
    assert_close(host(ctop), ctop_ref, TOL, "cosine top")


# --------------------------------------------------------------------------- #
# SimCross cosine (dist_mode 0): 1e-5 (BLAS order in the reference)
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("shape", [(8, 1, 1, 300), (4096, 1, 1, 300), (9, 1, 1, 7), (4091, 1, 1, 200), (33, 1, 1, 100),
                                   (17, 1, 1, 304), (4, 5, 7, 300), (2, 40, 40, 50), (3, 41, 9, 33), (1, 70, 60, 9)])
def test_cosine_forward_backward(shape, oracle, hiplib):
    from mms_answer_selection_amd import capi
    N, W1, W2, D = shape
    r = rng(sum(shape) + 1)
    q, a = qa(r, N, W1, W2, D)
    dT = r.standard_normal((N, 1, W1, W2)).astype(np.float32)
    top_ref, n0_ref, n1_ref = oracle.simcross_forward(0, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(0, q, a, top_ref, dT, norm0=n0_ref, norm1=n1_ref)

    qd, ad, dTd = dev(q), dev(a), dev(dT)
    top, n0, n1 = nan_like(top_ref.shape), nan_like(n0_ref.shape), nan_like(n1_ref.shape)
    capi.simcross_forward(0, qd, ad, top, norm0=n0, norm1=n1)
    assert_close(host(top), top_ref, TOL, "top")
    assert_close(host(n0), n0_ref, TOL, "norm0")   # the NORM, as the CPU code caches it
    assert_close(host(n1), n1_ref, TOL, "norm1")
    gq, ga = nan_like(q.shape), nan_like(a.shape)
    capi.simcross_backward(0, qd, ad, top, dTd, gq, ga, norm0=n0, norm1=n1)
    assert_close(host(gq), dq_ref, TOL, "dq")
    assert_close(host(ga), da_ref, TOL, "da")

    top2, n02, n12 = nan_like(top_ref.shape), nan_like(n0_ref.shape), nan_like(n1_ref.shape)
    gq2, ga2 = nan_like(q.shape), nan_like(a.shape)
    capi.simcross_forward_backward(0, qd, ad, dTd, top2, gq2, ga2, norm0=n02, norm1=n12)
    assert_bitexact(host(top2), host(top), "fused top vs two-call")
    assert_bitexact(host(gq2), host(gq), "fused dq vs two-call")
    assert_bitexact(host(ga2), host(ga), "fused da vs two-call")


# --------------------------------------------------------------------------- #
# SimCross bilinear (dist_mode 2) on MFMA: 1e-5
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("cfg", [
    (8, 1, 1, 300, 1, False),
    (2000, 1, 1, 300, 1, True),   # cfg 3 written as a SimCross layer: routed to SimMatrix's panel-GEMM launches
    (2304, 1, 1, 300, 1, True),   # ... and, from 2048 pairs, to the bf16-pipe products (the bias rides in one column group's half)
    (2100, 1, 1, 64, 1, False),   # ... one column group
    (777, 1, 1, 52, 1, True),
    (130, 1, 1, 301, 1, False),   # ... and its generic fallback (odd width)
    (4, 5, 7, 300, 2, True),
    (2, 40, 40, 50, 4, True),     # network_v4: mesure_count 4, bias_term true
    (3, 9, 4, 33, 3, True),
    (300, 1, 1, 64, 2, True),
    (1, 130, 70, 20, 1, False),   # more than one output tile per pair
])
def test_bilinear_forward_backward(cfg, oracle, hiplib):
    from mms_answer_selection_amd import capi
    N, W1, W2, D, M, bias_term = cfg
    r = rng(sum(cfg[:5]) + 2)
    q, a = qa(r, N, W1, W2, D)
    W = r.uniform(-0.08, 0.08, (M, D, D)).astype(np.float32)
    bias = r.standard_normal((M, W1, W2)).astype(np.float32) if bias_term else None
    dT = r.standard_normal((N, M, W1, W2)).astype(np.float32)
    dbias0 = r.standard_normal((M, W1, W2)).astype(np.float32) if bias_term else None
    top_ref, _, _ = oracle.simcross_forward(2, q, a, W, bias)
    dq_ref, da_ref, dW_ref, db_ref = oracle.simcross_backward(
        2, q, a, top_ref, dT, W=W, bias_term=bias_term, dbias_in=dbias0)

    qd, ad, Wd, bd, dTd = dev(q), dev(a), dev(W), dev(bias), dev(dT)
    top = nan_like(top_ref.shape)
    capi.simcross_forward(2, qd, ad, top, W=Wd, bias=bd)
    assert_close(host(top), top_ref, TOL, "top")
    gq, ga, gW = nan_like(q.shape), nan_like(a.shape), nan_like(W.shape)
    gb = dev(dbias0)
    capi.simcross_backward(2, qd, ad, top, dTd, gq, ga, W=Wd, bias_term=bias_term, dW=gW, dbias=gb)
    assert_close(host(gq), dq_ref, TOL, "dq")
    assert_close(host(ga), da_ref, TOL, "da")
    assert_close(host(gW), dW_ref, TOL, "dW (overwritten: W.diff zeroed at :256)")
    if bias_term:
        assert_bitexact(host(gb), db_ref, "dbias (accumulated, n ascending)")


def test_bilinear_shape_fuzz(oracle, hiplib):
    """Random geometries through every GEMM dispatch branch: 16-byte rows (D % 4 == 0), 8-byte rows
    (D % 4 == 2: the driver's default D = 50), odd D (stride-generic kernel), one or several measures
    (stacked split-K over m), W1 != W2, batch sizes on both sides of the deep-k-tile threshold."""
    from mms_answer_selection_amd import capi
    r = rng(77)
    shapes = [(50, 40, 40, 50, 4), (32, 40, 40, 300, 4), (70, 24, 40, 50, 3), (600, 1, 1, 50, 2),
              # >= 512 pairs of word grids: the fused one-launch forward (Q_n W_m kept in LDS)
              (520, 40, 40, 50, 4), (513, 17, 33, 64, 2), (600, 48, 48, 52, 1), (512, 5, 7, 9, 3)]
    for _ in range(10):
        shapes.append((int(r.integers(1, 40)), int(r.integers(1, 45)), int(r.integers(1, 45)),
                       int(r.choice([6, 10, 12, 18, 20, 33, 50, 64, 100])), int(r.integers(1, 5))))
    for (N, W1, W2, D, M) in shapes:
        q, a = qa(r, N, W1, W2, D)
        W = r.uniform(-0.08, 0.08, (M, D, D)).astype(np.float32)
        bias = r.standard_normal((M, W1, W2)).astype(np.float32)
        dT = r.standard_normal((N, M, W1, W2)).astype(np.float32)
        db0 = r.standard_normal((M, W1, W2)).astype(np.float32)
        top_ref, _, _ = oracle.simcross_forward(2, q, a, W, bias)
        dq_ref, da_ref, dW_ref, db_ref = oracle.simcross_backward(2, q, a, top_ref, dT, W=W, bias_term=True,
                                                                  dbias_in=db0)
        qd, ad, Wd, bd, dTd = dev(q), dev(a), dev(W), dev(bias), dev(dT)
        top, gq, ga, gW, gb = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape), nan_like(W.shape), dev(db0)
        capi.simcross_forward(2, qd, ad, top, W=Wd, bias=bd)
        capi.simcross_backward(2, qd, ad, top, dTd, gq, ga, W=Wd, bias_term=True, dW=gW, dbias=gb)
        what = "shape %s " % ((N, W1, W2, D, M),)
        assert_close(host(top), top_ref, TOL, what + "top")
        assert_close(host(gq), dq_ref, TOL, what + "dq")
        assert_close(host(ga), da_ref, TOL, what + "da")
        assert_close(host(gW), dW_ref, TOL, what + "dW")
        assert_bitexact(host(gb), db_ref, what + "dbias")


def test_bilinear_fused_forward_without_bias_and_ragged_tiles(oracle, hiplib):
    """The one-launch forward for large batches of word grids (>= 512 pairs): no bias, tile remainders in
    every dimension, and the largest geometry it accepts."""
    from mms_answer_selection_amd import capi
    r = rng(91)
    for (N, W1, W2, D, M) in [(512, 8, 8, 16, 1), (515, 17, 3, 5, 2), (512, 48, 48, 64, 1), (640, 1, 9, 7, 5)]:
        q, a = qa(r, N, W1, W2, D)
        W = r.uniform(-0.1, 0.1, (M, D, D)).astype(np.float32)
        top_ref, _, _ = oracle.simcross_forward(2, q, a, W, None)
        top = nan_like(top_ref.shape)
        capi.simcross_forward(2, dev(q), dev(a), top, W=dev(W), bias=None)
        assert_close(host(top), top_ref, TOL, "top %s" % ((N, W1, W2, D, M),))


# --------------------------------------------------------------------------- #
# SimMatrix: 1e-5
# --------------------------------------------------------------------------- #
@pytest.fixture(params=["bf16x3", "fp32"])
def matrix_mode(request, hiplib):
    """Both matrix pipes behind the learned-metric products (include/mms.h: mms_set_matrix_mode): the default -- exact
    three-way bf16 splits on the bf16 pipe, taken for N >= 2048 by calls that own a workspace -- and fp32 MFMA."""
    from mms_answer_selection_amd import capi
    capi.set_matrix_mode(request.param)
    yield request.param
    capi.set_matrix_mode("bf16x3")


# N >= 2048: the shapes the split-bf16 kernel takes (one and two column groups, ragged last row panel, K not a multiple
# of the 16-deep k-step, a single k-step, the 8-column minimum of the side job, N and K at the kernel's 320 limit)
@pytest.mark.parametrize("shape", [(16, 300, 300), (5, 7, 3), (700, 64, 48), (1, 1, 1), (900, 50, 50), (130, 33, 18), (2100, 12, 300),
                                   (2125, 300, 300), (4096, 64, 160), (2500, 52, 304), (3000, 300, 8), (2333, 20, 36),
                                   (2048, 320, 320), (2050, 16, 164), (2049, 33, 18)])
def test_simmatrix(shape, matrix_mode, oracle, hiplib):
    from mms_answer_selection_amd import capi
    N, K1, K2 = shape
    r = rng(sum(shape) + 3)
    q = (r.standard_normal((N, K1)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N, K2)) * 0.4).astype(np.float32)
    W = r.uniform(-0.08, 0.08, (K1, K2)).astype(np.float32)
    dT = r.standard_normal((N, 1)).astype(np.float32)
    dW0 = r.standard_normal((K1, K2)).astype(np.float32)
    top_ref, scratch_ref = oracle.simmatrix_forward(q, a, W)
    dq_ref, da_ref, dW_ref = oracle.simmatrix_backward(q, a, W, dT, dW_in=dW0)

    qd, ad, Wd, dTd = dev(q), dev(a), dev(W), dev(dT)
    top, scratch = nan_like((N, 1)), nan_like((N, K2))
    capi.simmatrix_forward(qd, ad, Wd, top, scratch)
    assert_close(host(top), top_ref, TOL, "top")
    assert_close(host(scratch), scratch_ref, TOL, "Q*W left in bottom[1].diff (:58)")
    gq, ga, gW = nan_like((N, K1)), nan_like((N, K2)), dev(dW0)
    capi.simmatrix_backward(qd, ad, Wd, dTd, gq, ga, gW)
    assert_close(host(gq), dq_ref, TOL, "dq")
    assert_close(host(ga), da_ref, TOL, "da")
    assert_close(host(gW), dW_ref, TOL, "dW (accumulated)")

    # propagate_down flags: untouched outputs stay untouched (:81-93)
    gq2, ga2, gW2 = nan_like((N, K1)), nan_like((N, K2)), dev(dW0)
    capi.simmatrix_backward(qd, ad, Wd, dTd, gq2, ga2, gW2, param_propagate_down=False,
                            propagate_down=(False, True))
    assert np.isnan(host(gq2)).all()
    assert_bitexact(host(gW2), dW0, "dW untouched")
    assert_bitexact(host(ga2), host(ga), "da")

    # cached entry point: the forward's Q*W is scaled instead of recomputed -- same bits, also in place
    gq3, ga3, gW3 = nan_like((N, K1)), nan_like((N, K2)), dev(dW0)
    capi.simmatrix_backward(qd, ad, Wd, dTd, gq3, ga3, gW3, qw=scratch)
    assert_bitexact(host(gq3), host(gq), "dq (cached call)")
    assert_bitexact(host(ga3), host(ga), "da from the cached Q*W")
    assert_bitexact(host(gW3), host(gW), "dW (cached call)")
    inplace = scratch.clone()                         # the reference's layout: Q*W sits in bottom[1].diff
    capi.simmatrix_backward(qd, ad, Wd, dTd, None, inplace, None, param_propagate_down=False,
                            propagate_down=(False, True), qw=inplace)
    assert_bitexact(host(inplace), host(ga), "da scaled in place")


@pytest.mark.parametrize("shape", [(16384, 304, 300), (2125, 64, 160), (100, 8, 4), (3000, 320, 320), (4097, 16, 36), (1, 24, 8), (2300, 1024, 64)])
def test_simmatrix_scoring_fp16_storage(shape, oracle, hiplib):
    """mms_simmatrix_forward_f16: q, a stored as halves, W fp32 -> fp32 scores, against the fp32 oracle run on the same
    fp16-rounded inputs at the layer's 1e-5 (the kernel forms every fp16 x fp32 product exactly: two bf16 planes x three)."""
    from mms_answer_selection_amd import capi
    N, K1, K2 = shape
    r = rng(3 * N + K1 + 5 * K2)
    qh = (r.standard_normal((N, K1)) * 0.4).astype(np.float16)
    ah = (r.standard_normal((N, K2)) * 0.4).astype(np.float16)
    W = r.uniform(-0.08, 0.08, (K1, K2)).astype(np.float32)
    top_ref, _ = oracle.simmatrix_forward(qh.astype(np.float32), ah.astype(np.float32), W)
    top = nan_like((N, 1))
    capi.simmatrix_forward_f16(torch.from_numpy(qh).cuda(), torch.from_numpy(ah).cuda(), dev(W), top)
    assert_close(host(top), top_ref, TOL, "scores")
    top2 = nan_like((N, 1))
    capi.set_matrix_mode("fp32")                        # the fp16-storage call does not depend on the matrix mode
    try:
        capi.simmatrix_forward_f16(torch.from_numpy(qh).cuda(), torch.from_numpy(ah).cuda(), dev(W), top2)
    finally:
        capi.set_matrix_mode("bf16x3")
    assert_bitexact(host(top2), host(top), "deterministic, mode-independent")


@pytest.mark.parametrize("shape", [(16384, 304, 304), (2125, 64, 160), (100, 8, 8), (3000, 320, 320), (4097, 16, 40), (1, 24, 8)])
def test_simmatrix_training_fp16_storage(shape, oracle, hiplib):
    """mms_simmatrix_forward_train_f16 / mms_simmatrix_backward_f16: q, a, dq, da stored as halves; W, dW, scores, top_diff
    and the Q.W scratch fp32.  Against the fp32 oracle on the fp16-rounded inputs: scores, Q.W and dW at 1e-5 (products of
    widened halves are formed exactly), dq / da at half precision."""
    from mms_answer_selection_amd import capi
    N, K1, K2 = shape
    r = rng(5 * N + K1 + 3 * K2)
    qh = (r.standard_normal((N, K1)) * 0.4).astype(np.float16)
    ah = (r.standard_normal((N, K2)) * 0.4).astype(np.float16)
    W = r.uniform(-0.08, 0.08, (K1, K2)).astype(np.float32)
    dT = r.standard_normal((N, 1)).astype(np.float32)
    dW0 = r.standard_normal((K1, K2)).astype(np.float32)
    q32, a32 = qh.astype(np.float32), ah.astype(np.float32)
    top_ref, qw_ref = oracle.simmatrix_forward(q32, a32, W)
    dq_ref, da_ref, dW_ref = oracle.simmatrix_backward(q32, a32, W, dT, dW_in=dW0)
    qd, ad, Wd, dTd = torch.from_numpy(qh).cuda(), torch.from_numpy(ah).cuda(), dev(W), dev(dT)
    top, qw = nan_like((N, 1)), nan_like((N, K2))
    capi.simmatrix_forward_train_f16(qd, ad, Wd, top, qw)
    assert_close(host(top), top_ref, TOL, "scores")
    assert_close(host(qw), qw_ref, TOL, "Q*W scratch")
    mk = lambda shp: torch.full(shp, float("nan"), dtype=torch.float16, device="cuda")
    gq, ga, gW = mk((N, K1)), mk((N, K2)), dev(dW0)
    capi.simmatrix_backward_f16(qd, ad, Wd, qw, dTd, gq, ga, gW)
    assert_close(host(gW), dW_ref, 2e-5 if N > 8192 else TOL, "dW (accumulated)")
    for got, ref, what in ((gq, dq_ref, "dq"), (ga, da_ref, "da")):
        g = host(got).astype(np.float32)
        assert np.isfinite(g).all(), what
        assert (np.abs(g - ref) <= np.abs(ref) * 2.0 ** -10 + 1e-5 * max(1.0, np.abs(ref).max()) + 2.0 ** -24).all(), what
    # propagate flags: NULL outputs are left alone, dW only when asked
    gq2, gW2 = mk((N, K1)), dev(dW0)
    capi.simmatrix_backward_f16(qd, ad, Wd, qw, dTd, gq2, None, None)
    assert (host(gq2).view(np.uint16) == host(gq).view(np.uint16)).all(), "dq alone"
    capi.simmatrix_backward_f16(qd, ad, Wd, None, dTd, None, None, gW2)
    assert_bitexact(host(gW2), host(gW), "dW alone")


def test_simmatrix_scoring_fp16_storage_refuses_what_it_cannot_do(hiplib):
    from mms_answer_selection_amd import capi
    q = torch.zeros((64, 12), dtype=torch.float16, device="cuda")            # K1 % 8 != 0
    a = torch.zeros((64, 8), dtype=torch.float16, device="cuda")
    top = torch.zeros((64, 1), device="cuda")
    with pytest.raises(capi.MMSError):
        capi.simmatrix_forward_f16(q, a, torch.zeros((12, 8), device="cuda"), top)
    q = torch.zeros((64, 16), dtype=torch.float16, device="cuda")
    a = torch.zeros((64, 324), dtype=torch.float16, device="cuda")           # K2 > 320
    with pytest.raises(capi.MMSError):
        capi.simmatrix_forward_f16(q, a, torch.zeros((16, 324), device="cuda"), top)


def test_simmatrix_matrix_pipes_agree_and_mode_errors(hiplib):
    """The two pipes give the same product to fp32 rounding (both far inside 1e-5), the workspace-less entry point stays on
    the fp32 pipe, a bad mode is refused, and the documented edge of the split (inf -> NaN) is what happens."""
    from mms_answer_selection_amd import capi
    N, K = 4096, 300
    r = rng(77)
    q = dev((r.standard_normal((N, K)) * 0.4).astype(np.float32))
    a = dev((r.standard_normal((N, K)) * 0.4).astype(np.float32))
    W = dev(r.uniform(-0.08, 0.08, (K, K)).astype(np.float32))
    out = {}
    for mode in ("bf16x3", "fp32"):
        capi.set_matrix_mode(mode)
        assert capi.get_matrix_mode() == mode
        top, qw = nan_like((N, 1)), nan_like((N, K))
        capi.simmatrix_forward(q, a, W, top, qw)
        out[mode] = (host(top), host(qw))
    capi.set_matrix_mode("bf16x3")
    top0, qw0 = nan_like((N, 1)), nan_like((N, K))
    capi.simmatrix_forward(q, a, W, top0, qw0, use_workspace=False)
    assert_bitexact(host(qw0), out["fp32"][1], "no workspace: the fp32 pipe")
    ref = host(q).astype(np.float64) @ host(W).astype(np.float64)
    e3, e32 = np.abs(out["bf16x3"][1] - ref).max(), np.abs(out["fp32"][1] - ref).max()
    assert e3 <= 2e-6 and e32 <= 2e-6, (e3, e32)
    assert e3 <= 2.0 * e32 + 1e-7, "the split product is as accurate as the fp32 one: %g vs %g" % (e3, e32)
    assert hiplib.mms_set_matrix_mode(2) == 1 and capi.get_matrix_mode() == "bf16x3"
    qi = q.clone(); qi[5, 7] = float("inf")
    top, qw = nan_like((N, 1)), nan_like((N, K))
    capi.simmatrix_forward(qi, a, W, top, qw)
    h = host(qw)
    assert np.isnan(h[5]).all() and np.isfinite(np.delete(h, 5, axis=0)).all(), "an infinity poisons its own row only"


# --------------------------------------------------------------------------- #
# PairRankLoss
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("cfg", [(64, 1, 1.0), (8, 5, 0.1), (4096, 1, 1.0), (100003, 1, 0.5)])
def test_pairrank(cfg, oracle, hiplib):
    from mms_answer_selection_amd import capi
    N, Cc, margin = cfg
    r = rng(N + Cc)
    a = r.uniform(0, 1, (N, Cc)).astype(np.float32)
    b = r.uniform(0, 1, (N, Cc)).astype(np.float32)
    y = (r.uniform(size=(N, Cc)) < 0.2).astype(np.float32)   # TREC-QA positive rate
    a[0] = b[0]                                     # similar == 0 exactly
    if N > 3:
        a[1, 0], b[1, 0], y[1, 0] = 0.25, 0.25 + margin, 1.0   # tie at the hinge
        a[2, 0], b[2, 0], y[2, 0] = 0.75, 0.25, 0.0            # (1-y)*similar > 0
        a[3, 0], b[3, 0], y[3, 0] = 0.25, 0.75, 0.0            # (1-y)*similar < 0
    loss_ref, ord_ref, sim_ref = oracle.pairrank_forward(a, b, y, margin)
    da_ref, db_ref = oracle.pairrank_backward(y, ord_ref, sim_ref, top_diff=2.0)

    ad, bd, yd = dev(a), dev(b), dev(y)
    o, s, loss = nan_like(a.shape), nan_like(a.shape), nan_like((1,))
    capi.pairrank_forward(ad, bd, yd, o, s, loss, margin=margin)
    assert_bitexact(host(o), ord_ref, "ordered_diff_")
    assert_bitexact(host(s), sim_ref, "similar_diff_")
    assert_close(host(loss)[0], loss_ref, TOL, "loss")
    ga, gb = nan_like(a.shape), nan_like(a.shape)
    capi.pairrank_backward(yd, o, s, ga, gb, top_diff=2.0)
    assert_bitexact(host(ga), da_ref, "da")
    assert_bitexact(host(gb), db_ref, "db")
    # only bottom[1]
    ga2, gb2 = nan_like(a.shape), nan_like(a.shape)
    capi.pairrank_backward(yd, o, s, ga2, gb2, top_diff=2.0, propagate_down=(False, True))
    assert np.isnan(host(ga2)).all()
    assert_bitexact(host(gb2), db_ref)


@pytest.mark.parametrize("N", [1, 77, 4095, 8192, 8193, 20001])
def test_loss_sum_reference_mode_is_bit_identical(N, oracle, hiplib):
    """MMS_LOSS_SUM_REFERENCE reproduces Forward_cpu's running fp32 sum (pair_rank_loss_layer.cpp:41-49) bit for bit,
    on data where that sum drifts from the exact mean by more than 1e-5 (near-constant terms of about 2): the default
    order-free sum stays within 2e-6 of the exact mean instead.  PairRankLoss alone and the fused triplet step."""
    from mms_answer_selection_amd import capi
    r = rng(N)
    a = (0.5 + 1e-3 * r.standard_normal((N, 1))).astype(np.float32)
    b = (0.5 + 1e-3 * r.standard_normal((N, 1))).astype(np.float32)
    y = (r.uniform(size=(N, 1)) < 0.8).astype(np.float32)
    margin = 2.0
    loss_ref, ord_ref, sim_ref = oracle.pairrank_forward(a, b, y, margin)
    exact = float((np.maximum(0.0, ord_ref.astype(np.float64)) + np.abs((1.0 - y.astype(np.float64)) * sim_ref.astype(np.float64))).mean())
    o, s_, loss = nan_like(a.shape), nan_like(a.shape), nan_like((1,))
    capi.pairrank_forward(dev(a), dev(b), dev(y), o, s_, loss, margin=margin)
    assert abs(host(loss)[0] - exact) <= 2e-6 * max(1.0, abs(exact))
    capi.set_loss_sum_mode("reference")
    try:
        loss2 = nan_like((1,))
        capi.pairrank_forward(dev(a), dev(b), dev(y), o, s_, loss2, margin=margin)
        assert_bitexact(host(loss2), np.array([loss_ref], np.float32), "PairRankLoss loss, reference sum")
        assert_bitexact(host(o), ord_ref)
        # the fused step: widths of the specialised kernels and of the generic ones
        for D in ((300, 52) if N <= 8193 else (100,)):
            q = (r.standard_normal((N, 1, D)) * 1e-3).astype(np.float32)
            ap = (q + r.standard_normal((N, 1, D)) * 1e-5).astype(np.float32)
            an = (q + r.standard_normal((N, 1, D)) * 1e-4).astype(np.float32)
            sp, _, _ = oracle.simcross_forward(1, q, ap)
            sn, _, _ = oracle.simcross_forward(1, q, an)
            lref, _, _ = oracle.pairrank_forward(sp.reshape(N, 1), sn.reshape(N, 1), y, margin)
            out = dict(s_pos=nan_like((N, 1)), s_neg=nan_like((N, 1)), loss=nan_like((1,)),
                       dq=nan_like(q.shape), da_pos=nan_like(q.shape), da_neg=nan_like(q.shape))
            capi.triplet_euclid_step(dev(q), dev(ap), dev(an), dev(y), margin=margin, **out)
            assert_bitexact(host(out["loss"]), np.array([lref], np.float32), "triplet loss, reference sum, D=%d" % D)
            assert_bitexact(host(out["s_pos"]).ravel(), sp.ravel())
    finally:
        capi.set_loss_sum_mode("fast")
    assert hiplib.mms_set_loss_sum_mode(5) == 1                   # MMS_ERR_INVALID_ARG
    assert hiplib.mms_get_loss_sum_mode() == 0


def test_pairrank_hinge_comparison_modes(oracle, hiplib):
    """Where margin - y*(a-b) is EXACTLY zero the reference's two backward implementations disagree:
    Backward_cpu gates the hinge term with `ordered > 0` (pair_rank_loss_layer.cpp:76, the default here and the
    oracle), the .cu kernel with `>= 0` (pair_rank_loss_layer.cu:51).  Both are selectable; they differ only there."""
    from mms_answer_selection_amd import capi
    N, margin = 64, 0.5
    r = rng(77)
    a = r.uniform(0, 1, (N, 1)).astype(np.float32)
    b = r.uniform(0, 1, (N, 1)).astype(np.float32)
    y = (r.uniform(size=(N, 1)) < 0.5).astype(np.float32)
    a[:4, 0], b[:4, 0], y[:4, 0] = 0.75, 0.25, 1.0          # diff = margin, y = 1: ordered == 0 exactly
    loss_ref, ord_ref, sim_ref = oracle.pairrank_forward(a, b, y, margin)
    assert (ord_ref[:4] == 0).all()
    da_ref, db_ref = oracle.pairrank_backward(y, ord_ref, sim_ref, top_diff=1.0)
    o, s, loss = nan_like(a.shape), nan_like(a.shape), nan_like((1,))
    capi.pairrank_forward(dev(a), dev(b), dev(y), o, s, loss, margin=margin)
    ga, gb = nan_like(a.shape), nan_like(a.shape)
    assert capi.get_pairrank_hinge_mode() == "cpu"
    capi.pairrank_backward(dev(y), o, s, ga, gb)
    assert_bitexact(host(ga), da_ref, "da, Backward_cpu comparison")
    try:
        capi.set_pairrank_hinge_mode("gpu")
        capi.pairrank_backward(dev(y), o, s, ga, gb)
    finally:
        capi.set_pairrank_hinge_mode("cpu")
    g = host(ga)
    assert_bitexact(g[4:], da_ref[4:], "elements off the tie are untouched by the mode")
    scale = np.float32(1.0) / np.float32(N)
    assert_bitexact(g[:4], np.full((4, 1), -scale, np.float32), "at the tie the .cu comparison lets the hinge term through")
    assert (da_ref[:4] == 0).all()


def test_triplet_step_many_launches_and_graph_replay(hiplib):
    """The loss of the fused step is summed inside the launch through arrival words that the launch hands back
    zeroed: thousands of launches (more than there are ticket slots) and replays of one captured graph must all
    return the first launch's bits."""
    from mms_answer_selection_amd import capi
    N, D = 4096, 300
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(N, 1, D, device="cuda", generator=g) * 0.4
    ap = q + 0.1 * torch.randn(N, 1, D, device="cuda", generator=g)
    an = torch.randn(N, 1, D, device="cuda", generator=g) * 0.4
    y = (torch.rand(N, 1, device="cuda", generator=g) < 0.8).float()
    mk = lambda: dict(s_pos=nan_like((N, 1)), s_neg=nan_like((N, 1)), loss=nan_like((1,)),
                      dq=nan_like(q.shape), da_pos=nan_like(q.shape), da_neg=nan_like(q.shape))
    ws = capi.TripletWorkspace()
    two = mk()
    capi.set_triplet_finish_mode("launch")
    try:
        capi.triplet_euclid_step(q, ap, an, y, margin=0.05, ws=ws, **two)      # loss summed by a second launch
    finally:
        capi.set_triplet_finish_mode("inlaunch")                               # the default
    first = mk()
    _many_launches(capi, q, ap, an, y, ws, mk, first, two)


def test_triplet_step_without_the_loss_scalar(oracle, hiplib):
    """loss = NULL: scores and gradients as always, no reduction of the display scalar (include/mms.h)."""
    from mms_answer_selection_amd import capi
    for (N, D) in ((4095, 300), (77, 52)):
        g = torch.Generator(device="cuda").manual_seed(N)
        q = torch.randn(N, 1, D, device="cuda", generator=g) * 0.4
        ap = q + 0.1 * torch.randn(N, 1, D, device="cuda", generator=g)
        an = torch.randn(N, 1, D, device="cuda", generator=g) * 0.4
        y = torch.ones(N, 1, device="cuda")
        mk = lambda: dict(s_pos=nan_like((N, 1)), s_neg=nan_like((N, 1)), dq=nan_like((N, 1, D)),
                          da_pos=nan_like((N, 1, D)), da_neg=nan_like((N, 1, D)))
        a, b = mk(), mk()
        capi.triplet_euclid_step(q, ap, an, y, loss=nan_like((1,)), margin=0.05, **a)
        capi.triplet_euclid_step(q, ap, an, y, loss=None, margin=0.05, **b)
        for k in a:
            assert_bitexact(host(b[k]), host(a[k]), k)


def test_triplet_step_inlaunch_loss_domain(hiplib):
    """In-launch mode sums the terms as integers in units of 2^-S: a term of 2^10 or more (a margin or labels in
    the hundreds) cannot be carried and the loss must come out NaN, not wrong; scores and gradients are
    unaffected, the two-launch mode has no such limit, and a batch beyond what a ticket slot covers falls back
    to the second launch by itself."""
    from mms_answer_selection_amd import capi
    N, D = 520, 300
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn(N, 1, D, device="cuda", generator=g) * 0.4
    ap = q + 0.1 * torch.randn(N, 1, D, device="cuda", generator=g)
    an = torch.randn(N, 1, D, device="cuda", generator=g) * 0.4
    y = torch.ones(N, 1, device="cuda")
    mk = lambda n=N: dict(s_pos=nan_like((n, 1)), s_neg=nan_like((n, 1)), loss=nan_like((1,)),
                          dq=nan_like((n, 1, D)), da_pos=nan_like((n, 1, D)), da_neg=nan_like((n, 1, D)))
    big, ref = mk(), mk()
    capi.triplet_euclid_step(q, ap, an, y, margin=5000.0, **big)
    capi.set_triplet_finish_mode("launch")
    try:
        capi.triplet_euclid_step(q, ap, an, y, margin=5000.0, **ref)
    finally:
        capi.set_triplet_finish_mode("inlaunch")
    assert np.isnan(host(big["loss"])[0])
    assert np.isfinite(host(ref["loss"])[0]) and host(ref["loss"])[0] > 4000
    assert_bitexact(host(big["dq"]), host(ref["dq"]))
    assert_bitexact(host(big["s_pos"]), host(ref["s_pos"]))
    # the words are handed back clean: the next in-domain launch is unaffected
    ok = mk()
    capi.triplet_euclid_step(q, ap, an, y, margin=0.05, **ok)
    capi.set_triplet_finish_mode("launch")
    try:
        capi.triplet_euclid_step(q, ap, an, y, margin=0.05, **ref)
    finally:
        capi.set_triplet_finish_mode("inlaunch")
    assert_close(host(ok["loss"])[0], host(ref["loss"])[0], TOL, "after a poisoned launch")
    # a term just inside the domain
    edge = mk()
    capi.triplet_euclid_step(q, ap, an, y, margin=1022.0, **edge)
    assert abs(host(edge["loss"])[0] - 1022.0) < 1.5
    # beyond a slot (131072 triplets): second launch, same entry point
    n2 = 131072 + 24
    q2 = torch.randn(n2, 1, 100, device="cuda", generator=g) * 0.4
    a2 = torch.randn(n2, 1, 100, device="cuda", generator=g) * 0.4
    y2 = torch.ones(n2, 1, device="cuda")
    o1 = dict(s_pos=nan_like((n2, 1)), s_neg=nan_like((n2, 1)), loss=nan_like((1,)),
              dq=nan_like(q2.shape), da_pos=nan_like(q2.shape), da_neg=nan_like(q2.shape))
    capi.triplet_euclid_step(q2, q2 + 0.05 * a2, a2, y2, margin=0.05, **o1)
    sp, sn = host(o1["s_pos"]).astype(np.float64), host(o1["s_neg"]).astype(np.float64)
    want = np.maximum(0.0, 0.05 - (sp - sn)).mean()
    assert_close(host(o1["loss"])[0], want, TOL, "loss of a batch beyond one slot")


def _many_launches(capi, q, ap, an, y, ws, mk, first, two):
    if True:
        capi.triplet_euclid_step(q, ap, an, y, margin=0.05, ws=ws, **first)
        torch.cuda.synchronize()
        ref = host(first["loss"]).copy()
        assert np.isfinite(ref).all()
        assert_close(ref[0], host(two["loss"])[0], TOL, "in-launch loss vs two-launch loss")
        assert_bitexact(host(first["dq"]), host(two["dq"]))
        out = mk()
        for i in range(1500):
            capi.triplet_euclid_step(q, ap, an, y, margin=0.05, ws=ws, **out)
            if i % 250 == 0:
                assert_bitexact(host(out["loss"]), ref, "launch %d" % i)
        assert_bitexact(host(out["loss"]), ref)
        assert_bitexact(host(out["dq"]), host(first["dq"]))
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=cap):
                for _ in range(4):
                    capi.triplet_euclid_step(q, ap, an, y, margin=0.05, ws=ws, **out)
        torch.cuda.current_stream().wait_stream(cap)
        for _ in range(50):
            out["loss"].fill_(float("nan"))
            gph.replay()
        torch.cuda.synchronize()
        assert_bitexact(host(out["loss"]), ref, "graph replay")


# --------------------------------------------------------------------------- #
# Fused (q, a+, a-) step == layer-by-layer oracle
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("cfg", [(8, 300), (4096, 300), (4091, 300), (1, 300), (77, 301), (5, 4), (19, 1024), (7, 400),
                                 (33, 200), (9, 100), (1000, 100)])
@pytest.mark.parametrize("finish", ["inlaunch", "launch"])
def test_triplet_step(cfg, finish, oracle, hiplib):
    from mms_answer_selection_amd import capi
    N, D = cfg
    margin, lw = 0.05, 1.0
    r = rng(N + D)
    q, ap = qa(r, N, 1, 1, D)
    _, an = qa(r, N, 1, 1, D)
    ap = (q + 0.1 * ap).astype(np.float32)          # positives closer than negatives, mostly
    y = (r.uniform(size=(N, 1)) < 0.8).astype(np.float32)
    sp, _, _ = oracle.simcross_forward(1, q, ap)
    sn, _, _ = oracle.simcross_forward(1, q, an)
    loss_ref, o, s = oracle.pairrank_forward(sp.reshape(N, 1), sn.reshape(N, 1), y, margin)
    gsp, gsn = oracle.pairrank_backward(y, o, s, top_diff=lw)
    dq_p, dap_ref, _, _ = oracle.simcross_backward(1, q, ap, sp, gsp.reshape(sp.shape))
    dq_n, dan_ref, _, _ = oracle.simcross_backward(1, q, an, sn, gsn.reshape(sn.shape))
    dq_ref = dq_p + dq_n                              # Split layer backward

    out = dict(s_pos=nan_like((N, 1)), s_neg=nan_like((N, 1)), loss=nan_like((1,)),
               dq=nan_like(q.shape), da_pos=nan_like(q.shape), da_neg=nan_like(q.shape))
    capi.set_triplet_finish_mode(finish)
    try:
        capi.triplet_euclid_step(dev(q), dev(ap), dev(an), dev(y), margin=margin, loss_weight=lw, **out)
    finally:
        capi.set_triplet_finish_mode("inlaunch")
    assert_bitexact(host(out["s_pos"]).ravel(), sp.ravel(), "s_pos")
    assert_bitexact(host(out["s_neg"]).ravel(), sn.ravel(), "s_neg")
    assert_close(host(out["loss"])[0], loss_ref, TOL, "loss")
    assert_bitexact(host(out["dq"]), dq_ref, "dq")
    assert_bitexact(host(out["da_pos"]), dap_ref, "da_pos")
    assert_bitexact(host(out["da_neg"]), dan_ref, "da_neg")


@pytest.mark.parametrize("cfg", [(13, 50), (77, 301), (9, 1100), (1, 7), (8, 50), (4096, 300), (19, 1024)])
def test_triplet_step_stays_inside_its_buffers(cfg, oracle, hiplib):
    """Every output and the workspace are allocated at EXACTLY the size the ABI asks for, inside one arena with
    canary words between and after them: the step (all three kernel families: D = 100/200/300, D % 4 == 0, and
    the generic one-workgroup-per-8-triplets kernel for any other width) must leave every canary untouched and
    give the oracle's bits."""
    import ctypes as C
    from mms_answer_selection_amd import capi
    N, D = cfg
    r = rng(N * 31 + D)
    q, ap = qa(r, N, 1, 1, D)
    _, an = qa(r, N, 1, 1, D)
    y = (r.uniform(size=(N, 1)) < 0.8).astype(np.float32)
    sp, _, _ = oracle.simcross_forward(1, q, ap)
    sn, _, _ = oracle.simcross_forward(1, q, an)
    loss_ref, o, sres = oracle.pairrank_forward(sp.reshape(N, 1), sn.reshape(N, 1), y, 0.05)
    gsp, gsn = oracle.pairrank_backward(y, o, sres, top_diff=1.0)
    dq_p, dap_ref, _, _ = oracle.simcross_backward(1, q, ap, sp, gsp.reshape(sp.shape))
    dq_n, dan_ref, _, _ = oracle.simcross_backward(1, q, an, sn, gsn.reshape(sn.shape))
    wsb = hiplib.mms_triplet_workspace_bytes(N)
    assert wsb % 4 == 0
    sizes = dict(ws=wsb // 4, s_pos=N, s_neg=N, loss=1, dq=N * D, da_pos=N * D, da_neg=N * D)
    CAN = 64                                                   # canary floats after every buffer (256 B keeps 16-B alignment)
    total = sum(((n + 3) // 4 * 4) + CAN for n in sizes.values())
    arena = torch.full((total,), -777.25, dtype=torch.float32, device="cuda")
    off, view = 0, {}
    for k, n in sizes.items():
        view[k] = arena[off:off + n]
        off += (n + 3) // 4 * 4 + CAN
    lib = capi.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    capi.check(lib.mms_triplet_workspace_init(view["ws"].data_ptr(), wsb, st), "init")
    dq_, dqa, dqn, dy = dev(q), dev(ap), dev(an), dev(y)
    for finish in ("inlaunch", "launch"):
        capi.set_triplet_finish_mode(finish)
        try:
            capi.check(lib.mms_triplet_euclid_step_f32(
                N, D, 0.05, 1.0, dq_.data_ptr(), dqa.data_ptr(), dqn.data_ptr(), dy.data_ptr(),
                view["s_pos"].data_ptr(), view["s_neg"].data_ptr(), view["loss"].data_ptr(), view["dq"].data_ptr(),
                view["da_pos"].data_ptr(), view["da_neg"].data_ptr(), view["ws"].data_ptr(), wsb, st), "step")
        finally:
            capi.set_triplet_finish_mode("inlaunch")
        torch.cuda.synchronize()
        h = arena.cpu().numpy()
        mask = np.ones(total, bool)
        off = 0
        for k, n in sizes.items():
            mask[off:off + n] = False
            off += (n + 3) // 4 * 4 + CAN
        assert (h[mask] == np.float32(-777.25)).all(), "a canary word was overwritten (%s, finish=%s)" % (cfg, finish)
        assert_bitexact(host(view["s_pos"]), sp.ravel(), "s_pos")
        assert_bitexact(host(view["s_neg"]), sn.ravel(), "s_neg")
        assert_bitexact(host(view["dq"]).reshape(q.shape), dq_p + dq_n, "dq")
        assert_bitexact(host(view["da_pos"]).reshape(q.shape), dap_ref, "da_pos")
        assert_bitexact(host(view["da_neg"]).reshape(q.shape), dan_ref, "da_neg")
        assert_close(host(view["loss"])[0], loss_ref, TOL, "loss")
        # the arrival words are zero again after the launch
        words = view["ws"][:1056 * 2].view(torch.int32)
        assert int(words.abs().max().item()) == 0, "arrival words not reset"


def test_triplet_workspaces_are_independent_and_recoverable(hiplib):
    """Two graphs of 300 steps each, captured on their own workspaces and replayed CONCURRENTLY on two streams,
    return the single launch's loss (round 2 drew arrival slots from a process-wide table at capture time: such
    graphs shared slots).  A workspace whose arrival words were left non-zero (what a launch that died mid-way
    leaves) gives NaN-or-stale, never silently affects ANOTHER workspace, and is whole again after reset()."""
    from mms_answer_selection_amd import capi
    N, D = 4096, 300
    g = torch.Generator(device="cuda").manual_seed(11)
    mkin = lambda: (torch.randn(N, 1, D, device="cuda", generator=g) * 0.4,
                    torch.randn(N, 1, D, device="cuda", generator=g) * 0.4,
                    torch.randn(N, 1, D, device="cuda", generator=g) * 0.4,
                    (torch.rand(N, 1, device="cuda", generator=g) < 0.8).float())
    mk = lambda: dict(s_pos=nan_like((N, 1)), s_neg=nan_like((N, 1)), loss=nan_like((1,)),
                      dq=nan_like((N, 1, D)), da_pos=nan_like((N, 1, D)), da_neg=nan_like((N, 1, D)))
    ins = [mkin(), mkin()]
    refs = []
    for x in ins:
        o = mk()
        capi.triplet_euclid_step(*x, margin=0.05, **o)
        torch.cuda.synchronize()
        refs.append(host(o["loss"]).copy())
    assert np.isfinite(refs[0]).all() and np.isfinite(refs[1]).all() and refs[0][0] != refs[1][0]
    wss = [capi.TripletWorkspace(), capi.TripletWorkspace()]
    outs = [mk(), mk()]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    graphs = []
    for i in range(2):
        streams[i].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[i]):
            capi.triplet_euclid_step(*ins[i], margin=0.05, ws=wss[i], **outs[i])     # allocates + initialises
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph, stream=streams[i]):
                for _ in range(300):
                    capi.triplet_euclid_step(*ins[i], margin=0.05, ws=wss[i], **outs[i])
            graphs.append(gph)
    torch.cuda.synchronize()
    for rep in range(6):
        for i in range(2):
            outs[i]["loss"].fill_(float("nan"))
        torch.cuda.synchronize()
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
        torch.cuda.synchronize()
        for i in range(2):
            assert_bitexact(host(outs[i]["loss"]), refs[i], "concurrent graph %d, replay %d" % (i, rep))
    # poison workspace 0's arrival words; workspace 1 is unaffected; reset() repairs workspace 0
    wss[0].buf[:1056 * 8].view(torch.int64).fill_((5 << 52) | (1 << 40))   # stray arrivals and a stray partial sum
    o0, o1 = mk(), mk()
    capi.triplet_euclid_step(*ins[0], margin=0.05, ws=wss[0], **o0)
    capi.triplet_euclid_step(*ins[1], margin=0.05, ws=wss[1], **o1)
    torch.cuda.synchronize()
    assert_bitexact(host(o1["loss"]), refs[1], "a poisoned workspace must not leak into another one")
    l0 = host(o0["loss"])[0]
    assert np.isnan(l0) or l0 != refs[0][0]                    # garbage in, no promise -- but only for ITS owner
    wss[0].reset()
    o0 = mk()
    capi.triplet_euclid_step(*ins[0], margin=0.05, ws=wss[0], **o0)
    torch.cuda.synchronize()
    assert_bitexact(host(o0["loss"]), refs[0], "after mms_triplet_workspace_init the workspace is whole again")


# --------------------------------------------------------------------------- #
# Fused learned-metric (q, a+, a-) step == SimMatrix x 2 -> PairRankLoss chain of the oracle
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("cfg", [(4096, 300, 300), (6145, 300, 300), (8192, 100, 200),   # the panel kernel's fast path
                                 (333, 300, 300), (64, 7, 5), (2048, 302, 300)])          # the layers, one by one
def test_triplet_simmatrix_step(cfg, oracle, hiplib):
    """mms_triplet_simmatrix_step_f32 against the layer-by-layer chain (sim_matrix_layer.cpp:53-95 twice with W shared,
    pair_rank_loss_layer.cpp:26-84, Split sum of dq): scores / loss / gradients at 1e-5, and -- given the GPU's own
    scores -- PairRankLoss's per-row decisions bitwise: da_pos / da_neg must be EXACTLY g * (row of Q W) with the g the
    oracle derives from those scores."""
    from mms_answer_selection_amd import capi
    N, K1, K2 = cfg
    margin, lw = 0.3, 1.0
    r = rng(N + K1 + 7 * K2)
    q = (r.standard_normal((N, K1)) * 0.4).astype(np.float32)
    ap = (r.standard_normal((N, K2)) * 0.4).astype(np.float32)
    an = (r.standard_normal((N, K2)) * 0.4).astype(np.float32)
    W = r.uniform(-0.08, 0.08, (K1, K2)).astype(np.float32)
    y = (r.uniform(size=(N, 1)) < 0.8).astype(np.float32)
    dW0 = r.standard_normal((K1, K2)).astype(np.float32)          # param diffs accumulate
    sp, qw = oracle.simmatrix_forward(q, ap, W)
    sn, _ = oracle.simmatrix_forward(q, an, W)
    loss_ref, o, s = oracle.pairrank_forward(sp, sn, y, margin)
    gsp, gsn = oracle.pairrank_backward(y, o, s, top_diff=lw)
    dq_p, dap_ref, dW1 = oracle.simmatrix_backward(q, ap, W, gsp, dW_in=dW0.copy())
    dq_n, dan_ref, dW_ref = oracle.simmatrix_backward(q, an, W, gsn, dW_in=dW1)
    out = dict(s_pos=nan_like((N, 1)), s_neg=nan_like((N, 1)), loss=nan_like((1,)), dq=nan_like(q.shape),
               da_pos=nan_like(ap.shape), da_neg=nan_like(an.shape), dW=dev(dW0))
    capi.triplet_simmatrix_step(dev(q), dev(ap), dev(an), dev(y), dev(W), margin=margin, loss_weight=lw, **out)
    g = {k: host(v) for k, v in out.items()}
    assert_close(g["s_pos"], sp, TOL, "s_pos")
    assert_close(g["s_neg"], sn, TOL, "s_neg")
    assert_close(g["loss"][0], loss_ref, TOL, "loss")
    # rows whose hinge argument sits within rounding of zero may decide differently from the oracle's scores: judge the
    # gradients against the chain evaluated at the GPU's OWN scores (bit-exact PairRankLoss), and the products at 1e-5
    loss2, o2, s2 = oracle.pairrank_forward(g["s_pos"], g["s_neg"], y, margin)
    g2p, g2n = oracle.pairrank_backward(y, o2, s2, top_diff=lw)
    assert_close(g["loss"][0], loss2, TOL, "loss at the GPU's scores")
    dq2p, dap2, dW2a = oracle.simmatrix_backward(q, ap, W, g2p, dW_in=dW0.copy())
    dq2n, dan2, dW2 = oracle.simmatrix_backward(q, an, W, g2n, dW_in=dW2a)
    assert_close(g["da_pos"], dap2, TOL, "da_pos")
    assert_close(g["da_neg"], dan2, TOL, "da_neg")
    assert_close(g["dq"], dq2p + dq2n, TOL, "dq")
    assert_close(g["dW"], dW2, 2e-5, "dW")
    flips = int(((g2p != gsp) | (g2n != gsn)).sum())
    assert flips <= max(2, N // 500), "%d rows decide the hinge differently from the oracle's scores" % flips
    # a second call accumulates dW again and leaves everything else unchanged; loss = NULL is allowed
    first = {k: v.clone() for k, v in out.items()}
    out["loss"] = None
    capi.triplet_simmatrix_step(dev(q), dev(ap), dev(an), dev(y), dev(W), margin=margin, loss_weight=lw, **out)
    for k in ("s_pos", "s_neg", "dq", "da_pos", "da_neg"):
        assert torch.equal(out[k], first[k]), k
    assert_close(host(out["dW"]), 2 * dW2 - dW0, 4e-5, "dW accumulated twice")


# --------------------------------------------------------------------------- #
# Size-independent properties at full BASELINE sizes
# --------------------------------------------------------------------------- #
def test_properties_full_size(hiplib):
    from mms_answer_selection_amd import capi
    N, D = 4096, 300
    g = torch.Generator(device="cuda").manual_seed(1701)
    q = torch.randn(N, 1, D, device="cuda", generator=g) * 0.4
    a = torch.randn(N, 1, D, device="cuda", generator=g) * 0.4
    dT = torch.randn(N, 1, 1, 1, device="cuda", generator=g)
    top, gq, ga = nan_like((N, 1, 1, 1)), nan_like(q.shape), nan_like(a.shape)
    capi.simcross_forward_backward(1, q, a, dT, top, gq, ga)
    t = host(top)
    assert ((t > 0) & (t <= 1)).all()                        # range of 1/(1+dist)
    assert_bitexact(host(ga), -host(gq), "da == -dq for W1=W2=1")
    # symmetry: swapping q and a leaves T unchanged bitwise ((q-a)^2 == (a-q)^2)
    top_s = nan_like((N, 1, 1, 1))
    capi.simcross_forward(1, a, q, top_s)
    assert_bitexact(host(top_s), t, "T(q,a) == T(a,q)")
    # determinism: same bits on a second run
    top2, gq2, ga2 = nan_like((N, 1, 1, 1)), nan_like(q.shape), nan_like(a.shape)
    capi.simcross_forward_backward(1, q, a, dT, top2, gq2, ga2)
    assert_bitexact(host(top2), t)
    assert_bitexact(host(gq2), host(gq))
    # linearity of backward in top_diff: dq(2*dT) == 2*dq(dT) (power of two: exact)
    gq3, ga3 = nan_like(q.shape), nan_like(a.shape)
    capi.simcross_backward(1, q, a, top, 2 * dT, gq3, ga3)
    assert_bitexact(host(gq3), 2 * host(gq), "backward is linear in top_diff")


def test_cfg3_simmatrix_full_size_against_fp64(matrix_mode, hiplib):
    """BASELINE cfg 3 (16384 x 300 x 300): fp64 closed form on the GPU box's host, on either matrix pipe."""
    from mms_answer_selection_amd import capi
    N, K = 16384, 300
    r = rng(33)
    q = (r.standard_normal((N, K)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N, K)) * 0.4).astype(np.float32)
    W = r.uniform(-0.08, 0.08, (K, K)).astype(np.float32)
    dT = r.standard_normal((N, 1)).astype(np.float32)
    qd, ad, Wd, dTd = dev(q), dev(a), dev(W), dev(dT)
    top, scratch = nan_like((N, 1)), nan_like((N, K))
    capi.simmatrix_forward(qd, ad, Wd, top, scratch)
    q6, a6, W6, g6 = (x.astype(np.float64) for x in (q, a, W, dT))
    assert_close(host(top), ((q6 @ W6) * a6).sum(1, keepdims=True), TOL, "top")
    gq, ga = nan_like((N, K)), nan_like((N, K))
    gW = torch.zeros(K, K, device="cuda")
    capi.simmatrix_backward(qd, ad, Wd, dTd, gq, ga, gW)
    assert_close(host(gq), g6 * (a6 @ W6.T), TOL, "dq")
    assert_close(host(ga), g6 * (q6 @ W6), TOL, "da")
    assert_close(host(gW), q6.T @ (g6 * a6), 2e-5, "dW (16384-term sums)")


# --------------------------------------------------------------------------- #
# Error behaviour of the C ABI
# --------------------------------------------------------------------------- #
def test_error_codes(hiplib):
    from mms_answer_selection_amd import capi
    L = hiplib
    q = torch.zeros(2, 1, 4, device="cuda")
    top = torch.zeros(2, 1, 1, 1, device="cuda")
    p = q.data_ptr()
    assert L.mms_simcross_forward_f32(3, 2, 1, 1, 4, 1, p, p, None, None, top.data_ptr(), None, None, None, 0, None) == 1
    assert L.mms_simcross_forward_f32(1, 2, 1, 1, 4, 1, None, p, None, None, top.data_ptr(), None, None, None, 0, None) == 1
    assert L.mms_simcross_forward_f32(1, 2, 1, 1, 4, 2, p, p, None, None, top.data_ptr(), None, None, None, 0, None) == 1
    assert L.mms_simcross_forward_f32(0, 2, 1, 1, 4, 1, p, p, None, None, top.data_ptr(), None, None, None, 0, None) == 1  # norms required
    assert L.mms_simcross_forward_f32(1, 0, 1, 1, 4, 1, None, None, None, None, None, None, None, None, 0, None) == 0   # empty batch
    # bilinear without workspace
    W = torch.zeros(1, 4, 4, device="cuda")
    assert L.mms_simcross_forward_f32(2, 2, 1, 1, 4, 1, p, p, W.data_ptr(), None, top.data_ptr(), None, None, None, 0, None) == 3
    assert L.mms_pairrank_forward_f32(0, 1.0, p, p, p, p, p, p, None, 0, None) == 1
    assert L.mms_error_string(3).decode().startswith("workspace")
    with pytest.raises(capi.MMSError):
        capi.simcross_forward(1, q.cpu(), q.cpu(), top.cpu())     # host tensors are refused


def test_argument_block_entry_points_match_the_plain_calls(oracle, hiplib):
    """mms_simcross_forward_block_f32 / _backward_block_f32 (arguments in one struct, for FFI hosts) are the two
    SimCross calls: same bits, same error for a bad block."""
    import ctypes as C
    from mms_answer_selection_amd import capi
    r = rng(91)
    N, D = 37, 300
    q, a = qa(r, N, 1, 1, D)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    qd, ad, dTd = dev(q), dev(a), dev(dT)
    top, gq, ga = nan_like(top_ref.shape), nan_like(q.shape), nan_like(a.shape)
    blk = capi.SimCrossArgs()
    blk.dist_mode, blk.N, blk.W1, blk.W2, blk.D, blk.M = 1, N, 1, 1, D, 1
    blk.q, blk.a, blk.top = qd.data_ptr(), ad.data_ptr(), top.data_ptr()
    blk.top_diff, blk.dq, blk.da = dTd.data_ptr(), gq.data_ptr(), ga.data_ptr()
    blk.propagate_down0 = blk.propagate_down1 = 1
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert hiplib.mms_simcross_forward_block_f32(C.byref(blk), st) == 0
    assert hiplib.mms_simcross_backward_block_f32(C.byref(blk), st) == 0
    assert_bitexact(host(top), top_ref, "top")
    assert_bitexact(host(gq), dq_ref, "dq")
    assert_bitexact(host(ga), da_ref, "da")
    assert hiplib.mms_simcross_forward_block_f32(None, st) != 0
    blk.q = None
    assert hiplib.mms_simcross_forward_block_f32(C.byref(blk), st) != 0
