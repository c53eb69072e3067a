"""GPU parity at the drop-in boundary: the C++ Layer mirror (SetUp / Forward /
Backward over Blobs, created by type string from prototxt) against the oracle,
including the reference-visible quirks of SURVEY.md Appendix A."""
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from util import TOL, assert_bitexact, assert_close, qa, rng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L(hiplib):
    from mms_answer_selection_amd import layers
    layers.lib()
    layers.set_mode_gpu()
    return layers


def blob(L, x):
    b = L.Blob(x.shape)
    b.data[...] = x
    return b


def test_simcross_euclid_layer_default_mode(L, oracle):
    """dist_mode defaults to 1 (caffe.proto:472); top is (N,1,W1,W2); legacy height() = D."""
    N, W1, W2, D = 4, 5, 7, 300
    r = rng(1)
    q, a = qa(r, N, W1, W2, D)
    lay = L.Layer('layer { name: "s" type: "SimCross" bottom: "q" bottom: "a" top: "t" }')
    bq, ba, top = blob(L, q), blob(L, a), L.Blob()
    lay.SetUp([bq, ba], [top])
    assert top.shape == (N, 1, W1, W2) and lay.blobs == []
    loss = lay.Forward([bq, ba], [top])
    assert loss == 0.0
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    assert_bitexact(top.data, top_ref)
    dT = r.standard_normal(top_ref.shape).astype(np.float32)
    top.diff[...] = dT
    bq.diff[...] = 7.0                       # must be overwritten, not accumulated (:176-177)
    lay.Backward([top], [True, False], [bq, ba])   # either flag -> BOTH computed (:201)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    assert_bitexact(bq.diff, dq_ref)
    assert_bitexact(ba.diff, da_ref)
    lay.Backward([top], [False, False], [bq, ba])
    assert (bq.diff == 0).all() and (ba.diff == 0).all()


def test_simcross_bilinear_layer_network_v4(L, oracle):
    """The driver's layer: dist_mode=2, mesure_count=4, bias on, default constant-0 fillers
    (do_trec_qa_clean.py:468)."""
    N, Wd, D, M = 3, 40, 50, 4
    r = rng(2)
    q, a = qa(r, N, Wd, Wd, D)
    lay = L.SimCross(dist_mode=2, mesure_count=M)
    bq, ba, top = blob(L, q), blob(L, a), L.Blob()
    lay.SetUp([bq, ba], [top])
    Wb, bb = lay.blobs
    assert Wb.shape == (M, D, D) and bb.shape == (M, Wd, Wd)       # blob order [W, bias]
    assert (Wb.data == 0).all() and (bb.data == 0).all()            # FillerParameter default
    W = r.uniform(-0.08, 0.08, (M, D, D)).astype(np.float32)
    bias = r.standard_normal((M, Wd, Wd)).astype(np.float32)
    Wb.data[...] = W
    bb.data[...] = bias
    lay.Forward([bq, ba], [top])
    assert top.shape == (N, M, Wd, Wd)
    top_ref, _, _ = oracle.simcross_forward(2, q, a, W, bias)
    assert_close(top.data, top_ref, TOL)
    dT = r.standard_normal(top_ref.shape).astype(np.float32)
    top.diff[...] = dT
    Wb.diff[...] = 5.0        # zeroed inside Backward (:256)
    bb.diff[...] = 0.25       # accumulated into (:301-304)
    lay.Backward([top], [True, True], [bq, ba])
    dq_ref, da_ref, dW_ref, db_ref = oracle.simcross_backward(
        2, q, a, top_ref, dT, W=W, bias_term=True, dbias_in=np.full_like(bias, 0.25))
    assert_close(bq.diff, dq_ref, TOL)
    assert_close(ba.diff, da_ref, TOL)
    assert_close(Wb.diff, dW_ref, TOL)
    assert_bitexact(bb.diff, db_ref)


def test_simcross_setup_ignores_preloaded_blobs_and_checks_dims(L):
    """LayerSetUp re-creates blobs_ every time (no 'Skipping parameter initialization')."""
    lay = L.SimCross(dist_mode=2, mesure_count=2, weight_filler=dict(type="constant", value=3.0))
    q = blob(L, np.zeros((2, 3, 8), np.float32))
    a = blob(L, np.zeros((2, 4, 8), np.float32))
    top = L.Blob()
    lay.SetUp([q, a], [top])
    lay.blobs[0].data[...] = -1.0
    lay.SetUp([q, a], [top])
    assert (lay.blobs[0].data == 3.0).all()


def test_simcross_cosine_layer(L, oracle):
    N, W1, W2, D = 6, 1, 1, 300
    r = rng(3)
    q, a = qa(r, N, W1, W2, D)
    lay = L.SimCross(dist_mode=0)
    bq, ba, top = blob(L, q), blob(L, a), L.Blob()
    lay.SetUp([bq, ba], [top])
    lay.Forward([bq, ba], [top])
    top_ref, n0, n1 = oracle.simcross_forward(0, q, a)
    assert_close(top.data, top_ref, TOL)
    dT = r.standard_normal(top_ref.shape).astype(np.float32)
    top.diff[...] = dT
    lay.Backward([top], [True, True], [bq, ba])
    dq_ref, da_ref, _, _ = oracle.simcross_backward(0, q, a, top_ref, dT, norm0=n0, norm1=n1)
    assert_close(bq.diff, dq_ref, TOL)
    assert_close(ba.diff, da_ref, TOL)


def test_simmatrix_layer_quirks(L, oracle):
    N, K1, K2 = 16, 300, 300
    r = rng(4)
    q = (r.standard_normal((N, K1)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N, K2)) * 0.4).astype(np.float32)
    L.set_random_seed(1701)
    lay = L.SimMatrix(weight_filler=dict(type="uniform", min=-0.08, max=0.08))
    bq, ba, top = blob(L, q), blob(L, a), L.Blob()
    lay.SetUp([bq, ba], [top])
    W = lay.blobs[0].data.copy()
    assert W.shape == (K1, K2) and W.min() >= -0.08 and W.max() <= 0.08 and W.std() > 0.03
    ba.diff[...] = 7.0
    lay.Forward([bq, ba], [top])
    assert top.shape == (N, 1)
    top_ref, scratch_ref = oracle.simmatrix_forward(q, a, W)
    assert_close(top.data, top_ref, TOL)
    # the reference's forward leaves Q*W in bottom[1].diff (:58): an observable side effect, kept by default
    assert_close(ba.diff, scratch_ref, TOL)
    dT = r.standard_normal((N, 1)).astype(np.float32)
    top.diff[...] = dT
    lay.blobs[0].diff[...] = 1.5                      # accumulates (:73-80)
    lay.Backward([top], [True, True], [bq, ba])
    dq_ref, da_ref, dW_ref = oracle.simmatrix_backward(q, a, W, dT, dW_in=np.full_like(W, 1.5))
    assert_close(bq.diff, dq_ref, TOL)
    assert_close(ba.diff, da_ref, TOL)
    assert_close(lay.blobs[0].diff, dW_ref, TOL)
    # a second Backward without a Forward in between finds da, not Q*W, in that diff: it recomputes
    lay.blobs[0].diff[...] = 1.5
    lay.Backward([top], [True, True], [bq, ba])
    assert_close(ba.diff, da_ref, TOL)
    assert_close(lay.blobs[0].diff, dW_ref, TOL)
    # opt-out: a private copy of the product; then the bottom's diff is left alone by Forward and may be
    # scribbled on before Backward
    lay.set_option("private_qw", 1)
    ba.diff[...] = 7.0
    lay.Forward([bq, ba], [top])
    assert (ba.diff == 7.0).all()
    top.diff[...] = dT
    ba.diff[...] = np.nan
    lay.blobs[0].diff[...] = 1.5
    lay.Backward([top], [True, True], [bq, ba])
    assert_close(bq.diff, dq_ref, TOL)
    assert_close(ba.diff, da_ref, TOL)
    assert_close(lay.blobs[0].diff, dW_ref, TOL)
    with pytest.raises(KeyError):
        lay.set_option("no_such_option", 1)
    # pre-loaded blobs are honoured by SimMatrix (:18-20): a second SetUp keeps W
    lay.SetUp([bq, ba], [top])
    assert_bitexact(lay.blobs[0].data, W)


def test_pairrank_layer(L, oracle):
    N = 64
    r = rng(5)
    a = r.uniform(0, 1, (N, 1)).astype(np.float32)
    b = r.uniform(0, 1, (N, 1)).astype(np.float32)
    y = (r.uniform(size=(N, 1)) < 0.2).astype(np.float32)
    lay = L.PairRankLoss(margin=0.1)
    ba, bb, by, top = blob(L, a), blob(L, b), blob(L, y), L.Blob((3,))
    lay.SetUp([ba, bb, by], [top])
    assert top.shape == () and top.count == 1               # scalar top, 0 axes
    assert top.diff.reshape(-1)[0] == 1.0                   # default loss_weight 1 (loss_layer.cpp:11-13)
    loss = lay.Forward([ba, bb, by], [top])
    loss_ref, o, s = oracle.pairrank_forward(a, b, y, 0.1)
    assert abs(loss - loss_ref) <= TOL * max(1, abs(loss_ref))
    assert abs(top.data.reshape(-1)[0] - loss_ref) <= TOL
    lay.Backward([top], [True, True, False], [ba, bb, by])
    da_ref, db_ref = oracle.pairrank_backward(y, o, s, top_diff=1.0)
    assert_bitexact(ba.diff, da_ref)
    assert_bitexact(bb.diff, db_ref)
    # loss_weight: 2 scales both the reported loss and the gradients
    lay2 = L.PairRankLoss(margin=0.1, loss_weight=2.0)
    lay2.SetUp([ba, bb, by], [top])
    assert abs(lay2.Forward([ba, bb, by], [top]) - 2 * loss_ref) <= 2 * TOL
    lay2.Backward([top], [True, True, False], [ba, bb, by])
    da2, _ = oracle.pairrank_backward(y, o, s, top_diff=2.0)
    assert_bitexact(ba.diff, da2)


def test_metric_layers_on_network_v4_outputs(L, oracle):
    """MRR / MAP / AUC as the test net wires them: L.MRR(prob, label, group) etc.
    (do_trec_qa_clean.py:494-496), prob = softmax output (N,2)."""
    r = rng(6)
    n, groups = 1517, 68
    group = np.sort(r.integers(0, groups, n)).astype(np.float32)
    label = (r.uniform(size=n) < 0.17).astype(np.float32)
    score = ((r.permutation(n) + 0.5) / n).astype(np.float32)
    prob = np.stack([1 - score, score], 1).astype(np.float32)
    bp, bl, bg = blob(L, prob), blob(L, label), blob(L, group)
    for make, ref in ((L.MAP, oracle.map_score(prob, label, group)[0]),
                      (L.MRR, oracle.mrr_score(prob, label, group)[0])):
        lay, top = make(), L.Blob((7,))
        lay.SetUp([bp, bl, bg], [top])
        assert top.shape == ()
        lay.Forward([bp, bl, bg], [top])
        assert np.float32(top.data.reshape(-1)[0]).view(np.uint32) == np.float32(ref).view(np.uint32)
    lay, top = L.AUC(), L.Blob()
    lay.SetUp([bp, bl], [top])
    lay.Forward([bp, bl], [top])
    assert top.data.reshape(-1)[0] == oracle.auc_score(prob, label)
    a = r.uniform(size=(64, 1)).astype(np.float32)
    b = r.uniform(size=(64, 1)).astype(np.float32)
    y = r.choice([-1.0, 1.0], (64, 1)).astype(np.float32)
    lay, top = L.Layer('layer { type: "RankAccuracy" }'), L.Blob()
    ba, bb, by = blob(L, a), blob(L, b), blob(L, y)
    lay.SetUp([ba, bb, by], [top])
    lay.Forward([ba, bb, by], [top])
    assert top.data.reshape(-1)[0] == oracle.rank_accuracy(a, b, y)


def _run_snippet(code):
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n%s" % (ROOT, code)],
                          capture_output=True, text=True, timeout=300)


def test_fatal_errors_abort_like_caffe():
    """CHECK / LOG(FATAL) => message + abort (no exceptions, no status codes)."""
    pre = ("from mms_answer_selection_amd import layers as L\nimport numpy as np\n"
           "def blob(x):\n b = L.Blob(x.shape); b.data[...] = x; return b\n")
    # label propagation (pair_rank_loss_layer.cpp:57-60)
    r = _run_snippet(pre + "z = np.zeros((4,1), np.float32)\nl = L.PairRankLoss()\nt = L.Blob()\n"
                     "bs = [blob(z), blob(z), blob(z)]\nl.SetUp(bs, [t]); l.Forward(bs, [t])\n"
                     "l.Backward([t], [True, True, True], bs)\nprint('survived')")
    assert r.returncode != 0 and "cannot backpropagate to label inputs" in r.stderr and "survived" not in r.stdout
    # mismatched embedding dims (sim_cross_layer.cpp:14)
    r = _run_snippet(pre + "l = L.SimCross()\nl.SetUp([blob(np.zeros((2,3,8), np.float32)), "
                     "blob(np.zeros((2,3,9), np.float32))], [L.Blob()])\nprint('survived')")
    assert r.returncode != 0 and "height" in r.stderr and "survived" not in r.stdout
    # wrong bottom count (layer.hpp CheckBlobCounts)
    r = _run_snippet(pre + "l = L.SimCross()\nl.SetUp([blob(np.zeros((2,3,8), np.float32))], [L.Blob()])\nprint('survived')")
    assert r.returncode != 0 and "2 bottom blob" in r.stderr
    # CPU mode is not served by this library
    r = _run_snippet(pre + "L.set_mode_cpu()\nl = L.SimCross()\nq = blob(np.zeros((2,1,8), np.float32))\n"
                     "t = L.Blob()\nl.SetUp([q, q], [t]); l.Forward([q, q], [t])\nprint('survived')")
    assert r.returncode != 0 and "CPU mode" in r.stderr and "survived" not in r.stdout


def test_simcross_layer_pins_its_own_euclid_backward_mode(L, oracle):
    """The arithmetic of the Euclidean backward term is a per-thread setting of the C ABI; a layer can pin its
    own (mms_layer_set_option) and the thread's setting is untouched afterwards."""
    from mms_answer_selection_amd import capi
    N, D = 64, 300
    r = rng(31)
    q = (r.standard_normal((N, 1, D)) * 0.4).astype(np.float32)
    a = (r.standard_normal((N, 1, D)) * 0.4).astype(np.float32)
    dT = r.standard_normal((N, 1, 1, 1)).astype(np.float32)
    lay = L.SimCross(dist_mode=1)
    bq, ba, top = blob(L, q), blob(L, a), L.Blob()
    lay.SetUp([bq, ba], [top])
    lay.Forward([bq, ba], [top])
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    dq_ref, da_ref, _, _ = oracle.simcross_backward(1, q, a, top_ref, dT)
    capi.set_euclid_backward_mode("fp32")
    try:
        lay.set_option("euclid_backward_mode", 1)           # reference rounding for this layer only
        top.diff[...] = dT
        lay.Backward([top], [True, True], [bq, ba])
        assert_bitexact(bq.diff, dq_ref)
        assert_bitexact(ba.diff, da_ref)
        assert capi.get_euclid_backward_mode() == "fp32"    # the thread's mode was put back
        lay.set_option("euclid_backward_mode", -1)          # back to the thread's mode: <= 2 ulp, not bit-exact
        lay.Backward([top], [True, True], [bq, ba])
        ulps = np.abs(bq.diff.view(np.int32).astype(np.int64) - dq_ref.view(np.int32).astype(np.int64))
        assert ulps.max() <= 2
    finally:
        capi.set_euclid_backward_mode("reference")          # what conftest.py runs the GPU tests in
