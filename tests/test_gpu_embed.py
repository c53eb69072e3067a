"""GPU parity for the Embed layer: gather forward, ordered
scatter-add backward (bit-exact vs the CPU order), the weight_source loaders,
and the Embed -> SimCross chain the driver builds (do_trec_qa_clean.py:461-468)."""
import struct

import numpy as np
import pytest
import torch

from util import TOL, assert_bitexact, assert_close, rng

pytestmark = pytest.mark.gpu


def dev(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("cfg", [(50 * 80, 50, 3000, True), (37, 300, 11, False), (1, 4, 1, True),
                                 (1517 * 8, 50, 20000, True)])
def test_embed_forward_backward(cfg, oracle, hiplib):
    from mms_answer_selection_amd import capi
    M, N, K, use_bias = cfg
    r = rng(M + N)
    index = r.integers(0, K, M)
    index[r.uniform(size=M) < 0.6] = K - 1            # the zero-pad word id dominates a TREC-QA batch
    index = index.astype(np.float32)
    weight = r.uniform(-0.08, 0.08, (K, N)).astype(np.float32)
    bias = r.standard_normal(N).astype(np.float32) if use_bias else None
    top_ref = oracle.embed_forward(index, weight, bias)
    top = torch.full((M, N), float("nan"), device="cuda")
    capi.embed_forward(dev(index), dev(weight), top, bias=dev(bias))
    assert_bitexact(top.cpu().numpy(), top_ref, "top")
    dT = r.standard_normal((M, N)).astype(np.float32)
    wd0 = r.standard_normal((K, N)).astype(np.float32)   # diffs ACCUMULATE (embed_layer.cpp:170,177)
    bd0 = r.standard_normal(N).astype(np.float32) if use_bias else None
    wd_ref, bd_ref = oracle.embed_backward(index, dT, wd0, bd0)
    wd, bd = dev(wd0), dev(bd0)
    capi.embed_backward(dev(index), dev(dT), wd, bd)
    assert_bitexact(wd.cpu().numpy(), wd_ref, "weight_diff (n-ascending sums)")
    if use_bias:
        assert_close(bd.cpu().numpy(), bd_ref, TOL, "bias_diff")
    wd2 = dev(wd0)
    capi.embed_backward(dev(index), dev(dT), wd2, None)   # same bits on a second run (no atomics)
    assert_bitexact(wd2.cpu().numpy(), wd_ref)


@pytest.mark.parametrize("cfg", [(2000, 2000, 50, 20000), (2000, 1360, 50, 300), (3, 5, 7, 4), (6000, 7000, 50, 2000),
                                 (1, 1, 300, 1)])
def test_embed_backward_pair_is_the_two_backwards(cfg, oracle, hiplib):
    """mms_embed_backward_pair_f32: two Embed layers over one table in one pass == layer 0's Backward then layer 1's
    (embed_layer.cpp:155-180 twice into the same diffs): weight_diff bit for bit, bias_diff at 1e-5 (gemv)."""
    from mms_answer_selection_amd import capi
    M0, M1, N, K = cfg
    r = rng(M0 + 3 * M1 + N)
    idx = []
    for M in (M0, M1):
        i = r.integers(0, K, M)
        i[r.uniform(size=M) < 0.3] = K - 1              # zero-pad id: a long segment that spans both layers
        idx.append(i.astype(np.float32))
    d0 = r.standard_normal((M0, N)).astype(np.float32)
    d1 = r.standard_normal((M1, N)).astype(np.float32)
    wd0 = r.standard_normal((K, N)).astype(np.float32)
    bd0 = r.standard_normal(N).astype(np.float32)
    wd_a, bd_a = oracle.embed_backward(idx[0], d0, wd0, bd0)
    wd_ref, bd_ref = oracle.embed_backward(idx[1], d1, wd_a, bd_a)
    wd, bd = dev(wd0), dev(bd0)
    capi.embed_backward_pair(dev(idx[0]), dev(idx[1]), dev(d0), dev(d1), wd, bias_diff=bd)
    assert_bitexact(wd.cpu().numpy(), wd_ref, "weight_diff: layer 0's rows ascending, then layer 1's")
    assert_close(bd.cpu().numpy(), bd_ref, TOL, "bias_diff")
    # ... and equals the two single-layer calls of this library bit for bit
    wd2, bd2 = dev(wd0), dev(bd0)
    capi.embed_backward(dev(idx[0]), dev(d0), wd2, bd2)
    capi.embed_backward(dev(idx[1]), dev(d1), wd2, bd2)
    assert_bitexact(wd.cpu().numpy(), wd2.cpu().numpy(), "pair == two calls")
    assert_close(bd.cpu().numpy(), bd2.cpu().numpy(), TOL, "bias_diff pair vs two calls")


@pytest.mark.parametrize("cfg", [(2000, 2000, 50, 20000, True), (1200, 2896, 50, 300, True), (3, 5, 7, 4, False),
                                 (3000, 3000, 50, 2000, True)])
def test_embed_pair_forward_builds_the_index_the_backward_uses(cfg, oracle, hiplib):
    """mms_embed_forward_pair_f32 = the two forwards (bit for bit) + the inverted index;
    mms_embed_backward_pair_indexed_f32 = mms_embed_backward_pair_f32 minus the index build: same bits."""
    from mms_answer_selection_amd import capi
    M0, M1, N, K, use_bias = cfg
    r = rng(5 * M0 + M1 + N)
    idx = []
    for M in (M0, M1):
        i = r.integers(0, K, M)
        i[r.uniform(size=M) < 0.3] = K - 1
        idx.append(i.astype(np.float32))
    weight = r.uniform(-0.08, 0.08, (K, N)).astype(np.float32)
    bias = r.standard_normal(N).astype(np.float32) if use_bias else None
    t0 = torch.full((M0, N), float("nan"), device="cuda")
    t1 = torch.full((M1, N), float("nan"), device="cuda")
    index = capi.EmbedPairIndex()
    built = capi.embed_forward_pair(dev(idx[0]), dev(idx[1]), dev(weight), t0, t1, bias=dev(bias), index=index)
    assert built == (M0 + M1 <= 4096)
    assert_bitexact(t0.cpu().numpy(), oracle.embed_forward(idx[0], weight, bias), "top0")
    assert_bitexact(t1.cpu().numpy(), oracle.embed_forward(idx[1], weight, bias), "top1")
    if not built:
        with pytest.raises(capi.MMSError):
            capi.embed_backward_pair_indexed(dev(idx[0]), dev(idx[1]), t0, t1, dev(weight), index)
        return
    d0 = r.standard_normal((M0, N)).astype(np.float32)
    d1 = r.standard_normal((M1, N)).astype(np.float32)
    wd0 = r.standard_normal((K, N)).astype(np.float32)
    bd0 = r.standard_normal(N).astype(np.float32)
    wd_a, bd_a = oracle.embed_backward(idx[0], d0, wd0, bd0)
    wd_ref, bd_ref = oracle.embed_backward(idx[1], d1, wd_a, bd_a)
    wd, bd = dev(wd0), dev(bd0)
    capi.embed_backward_pair_indexed(dev(idx[0]), dev(idx[1]), dev(d0), dev(d1), wd, index, bias_diff=bd)
    assert_bitexact(wd.cpu().numpy(), wd_ref, "weight_diff from the forward's index")
    assert_close(bd.cpu().numpy(), bd_ref, TOL, "bias_diff")
    wd2 = dev(wd0)
    capi.embed_backward_pair_indexed(dev(idx[0]), dev(idx[1]), dev(d0), dev(d1), wd2, index)       # the index is read-only
    assert_bitexact(wd2.cpu().numpy(), wd_ref, "second use of the same index")


@pytest.mark.parametrize("cfg", [(1517, 40, 40, 50, 20000), (1100, 16, 24, 50, 300), (7, 40, 40, 50, 100),
                                 (5, 9, 13, 33, 40), (3, 40, 40, 300, 500), (600, 1, 1, 20, 64)])
def test_embed_fused_into_simcross_forward(cfg, oracle, hiplib):
    """mms_embed_simcross_forward_f32 == Embed (without and with its bias blob) followed by SimCross: Euclidean scores carry the
    CPU's bits; cosine 1e-5.  Covers the pair-image kernel (many pairs, D = 50) and the generic tiles."""
    from mms_answer_selection_amd import capi
    N, W1, W2, D, K = cfg
    r = rng(sum(cfg))
    weight = r.uniform(-0.5, 0.5, (K, D)).astype(np.float32)
    iq = r.integers(0, K, (N, W1)).astype(np.float32)
    ia = r.integers(0, K, (N, W2)).astype(np.float32)
    ia[r.uniform(size=ia.shape) < 0.4] = K - 1          # zero-pad id
    if N > 2:
        ia[1, 0] = iq[1, 0]                              # identical rows: T = 1
    q = oracle.embed_forward(iq.reshape(-1), weight, None).reshape(N, W1, D)
    a = oracle.embed_forward(ia.reshape(-1), weight, None).reshape(N, W2, D)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    top = torch.full(top_ref.shape, float("nan"), device="cuda")
    capi.embed_simcross_forward(1, dev(iq), dev(ia), dev(weight), top)
    assert_bitexact(top.cpu().numpy(), top_ref, "Euclid scores of the fused call")
    ctop_ref, n0_ref, n1_ref = oracle.simcross_forward(0, q, a)
    ctop = torch.full(top_ref.shape, float("nan"), device="cuda")
    n0 = torch.full(n0_ref.shape, float("nan"), device="cuda")
    n1 = torch.full(n1_ref.shape, float("nan"), device="cuda")
    capi.embed_simcross_forward(0, dev(iq), dev(ia), dev(weight), ctop, norm0=n0, norm1=n1)
    assert_close(ctop.cpu().numpy(), ctop_ref, TOL, "cosine scores of the fused call")
    assert_close(n0.cpu().numpy(), n0_ref, TOL, "norm0")
    # with the Embed layers' bias (the driver's layers have one): row = bias + table[id], one rounding
    eb = r.uniform(-0.3, 0.3, D).astype(np.float32)
    qb = oracle.embed_forward(iq.reshape(-1), weight, eb).reshape(N, W1, D)
    ab = oracle.embed_forward(ia.reshape(-1), weight, eb).reshape(N, W2, D)
    topb_ref, _, _ = oracle.simcross_forward(1, qb, ab)
    topb = torch.full(top_ref.shape, float("nan"), device="cuda")
    capi.embed_simcross_forward(1, dev(iq), dev(ia), dev(weight), topb, embed_bias=dev(eb))
    assert_bitexact(topb.cpu().numpy(), topb_ref, "Euclid scores of the fused call, Embed bias")
    cb_ref, n0b_ref, _ = oracle.simcross_forward(0, qb, ab)
    capi.embed_simcross_forward(0, dev(iq), dev(ia), dev(weight), ctop, norm0=n0, norm1=n1, embed_bias=dev(eb))
    assert_close(ctop.cpu().numpy(), cb_ref, TOL, "cosine scores of the fused call, Embed bias")
    assert_close(n0.cpu().numpy(), n0b_ref, TOL, "norm0, Embed bias")
    # ids outside the table are clamped like mms_embed_forward_f32
    iq2 = iq.copy(); iq2[0, 0] = -3.0; iq2[-1, -1] = K + 5.0
    iq2c = iq2.copy(); iq2c[0, 0] = 0.0; iq2c[-1, -1] = K - 1.0
    t1 = torch.empty_like(top); t2 = torch.empty_like(top)
    capi.embed_simcross_forward(1, dev(iq2), dev(ia), dev(weight), t1)
    capi.embed_simcross_forward(1, dev(iq2c), dev(ia), dev(weight), t2)
    assert_bitexact(t1.cpu().numpy(), t2.cpu().numpy(), "clamped ids")


@pytest.mark.parametrize("cfg", [(1517, 40, 40, 50, 4, 20000, True), (600, 16, 24, 64, 2, 300, False),
                                 (50, 40, 40, 50, 4, 1000, True), (7, 9, 13, 33, 3, 40, True)])
def test_embed_fused_into_bilinear_forward(cfg, oracle, hiplib):
    """mms_embed_simcross_bilinear_forward_f32 (dist_mode 2, the mode network_v4 scores with) == Embed x2 followed
    by SimCross: the same kernels on the same operand values, so the same BITS as the three separate calls; 1e-5
    against the oracle (BLAS order in the reference).  Covers both fused forward kernels (evaluation batches of
    >= 512 pairs, training batches of <= 256)."""
    from mms_answer_selection_amd import capi
    N, W1, W2, D, M, K, use_bias = cfg
    r = rng(sum(cfg[:6]))
    weight = r.uniform(-0.5, 0.5, (K, D)).astype(np.float32)
    iq = r.integers(0, K, (N, W1)).astype(np.float32)
    ia = r.integers(0, K, (N, W2)).astype(np.float32)
    ia[r.uniform(size=ia.shape) < 0.4] = K - 1          # zero-pad id
    iq[0, 0] = -2.0                                      # clamped like mms_embed_forward_f32
    ia[-1, -1] = K + 7.0
    Wm = (r.standard_normal((M, D, D)) * 0.1).astype(np.float32)
    bias = r.standard_normal((M, W1, W2)).astype(np.float32) if use_bias else None
    q = oracle.embed_forward(np.clip(iq, 0, K - 1).reshape(-1), weight, None).reshape(N, W1, D)
    a = oracle.embed_forward(np.clip(ia, 0, K - 1).reshape(-1), weight, None).reshape(N, W2, D)
    top_ref, _, _ = oracle.simcross_forward(2, q, a, W=Wm, bias=bias)
    top = torch.full(top_ref.shape, float("nan"), device="cuda")
    wd, Wd, bd = dev(weight), dev(Wm), (dev(bias) if use_bias else None)
    capi.embed_simcross_bilinear_forward(dev(iq), dev(ia), wd, Wd, bd, top)
    assert_close(top.cpu().numpy(), top_ref, TOL, "bilinear scores of the fused call")
    qd = torch.empty(N, W1, D, device="cuda"); ad = torch.empty(N, W2, D, device="cuda")
    capi.embed_forward(dev(iq).view(-1), wd, qd.view(N * W1, D))
    capi.embed_forward(dev(ia).view(-1), wd, ad.view(N * W2, D))
    top2 = torch.full(top_ref.shape, float("nan"), device="cuda")
    capi.simcross_forward(2, qd, ad, top2, W=Wd, bias=bd)
    assert_bitexact(top.cpu().numpy(), top2.cpu().numpy(), "fused == Embed, Embed, SimCross")
    # with the Embed layers' bias blob
    eb = dev(r.uniform(-0.3, 0.3, D).astype(np.float32))
    capi.embed_forward(dev(iq).view(-1), wd, qd.view(N * W1, D), bias=eb)
    capi.embed_forward(dev(ia).view(-1), wd, ad.view(N * W2, D), bias=eb)
    capi.simcross_forward(2, qd, ad, top2, W=Wd, bias=bd)
    capi.embed_simcross_bilinear_forward(dev(iq), dev(ia), wd, Wd, bd, top, embed_bias=eb)
    assert_bitexact(top.cpu().numpy(), top2.cpu().numpy(), "fused == Embed, Embed, SimCross (Embed bias)")


def test_embed_fused_bilinear_refuses_other_geometries(hiplib):
    from mms_answer_selection_amd import capi
    N, W1, W2, D, M, K = 3, 40, 40, 300, 2, 50                # D beyond the fused forward kernels
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(capi.MMSError):
        capi.embed_simcross_bilinear_forward(z(N, W1), z(N, W2), z(K, D), z(M, D, D), None, z(N, M, W1, W2))


def test_embed_layer_weight_sources_and_chain(tmp_path, oracle, hiplib):
    from mms_answer_selection_amd import layers as L
    L.lib(); L.set_mode_gpu()
    r = rng(8)
    V, Dw = 12, 50
    vecs = r.uniform(-1, 1, (V, Dw)).astype(np.float32)
    # 1. GloVe-style text file: "<word> v1 ... vD" (do_trec_qa_clean.py:283-289)
    txt = tmp_path / "wiki_dict.txt"
    txt.write_text("".join("w%d %s\n" % (i, " ".join("%.6f" % v for v in vecs[i])) for i in range(V)))
    # 2. the ".all" format: header "<float> <K-1> <N-1>", rows "<id> v... <word>"
    allf = tmp_path / "dict.all"
    allf.write_text("0.5 %d %d\n" % (V + 2 - 1, Dw - 1) +
                    "".join("%d %s w%d\n" % (i, " ".join("%.6f" % v for v in vecs[i]), i) for i in range(V)))
    # 3. word2vec binary
    binf = tmp_path / "vectors.bin"
    with open(binf, "wb") as f:
        f.write(b"%d %d\n" % (V, Dw))
        for i in range(V):
            f.write(b"w%d " % i + struct.pack("%df" % Dw, *vecs[i]) + b"\n")
    expect_txt = np.array([[float("%.6f" % v) for v in row] for row in vecs], np.float32)
    for path, expect in ((txt, expect_txt), (allf, expect_txt), (binf, vecs)):
        lay = L.Embed(input_dim=V + 2, num_output=Dw, bias_term=False, weight_source=str(path),
                      weight_filler=dict(type="constant", value=7.0))
        idx = L.Blob((3, 5))
        top = L.Blob()
        lay.SetUp([idx], [top])
        W = lay.blobs[0].data
        assert W.shape == (V + 2, Dw)
        assert_bitexact(W[:V], expect, str(path))
        assert (W[V:] == 7.0).all()                     # +2 rows (unknown, zero-pad) keep the filler
        assert top.shape == (3, 5, Dw)
    # Embed(question), Embed(answer) -> SimCross, as network_v4 wires them
    q_idx = r.integers(0, V + 2, (4, 6)).astype(np.float32)
    a_idx = r.integers(0, V + 2, (4, 6)).astype(np.float32)
    bq, ba, tq, ta, ts = L.Blob((4, 6)), L.Blob((4, 6)), L.Blob(), L.Blob(), L.Blob()
    bq.data[...] = q_idx
    ba.data[...] = a_idx
    lay.SetUp([bq], [tq]); lay.Forward([bq], [tq])
    lay.Forward([ba], [ta])                             # shared weights ('w2v-weights')
    sim = L.SimCross()
    sim.SetUp([tq, ta], [ts]); sim.Forward([tq, ta], [ts])
    Wnp = lay.blobs[0].data.copy()
    top_ref, _, _ = oracle.simcross_forward(1, Wnp[q_idx.astype(int)], Wnp[a_idx.astype(int)])
    assert_bitexact(ts.data, top_ref, "SimCross on embedded words")
    # backward through SimCross then Embed
    ts.diff[...] = r.standard_normal(ts.shape).astype(np.float32)
    sim.Backward([ts], [True, True], [tq, ta])
    lay.blobs[0].diff[...] = 0
    dq = tq.diff.copy()
    lay.Backward([tq], [False], [bq])
    wd_ref, _ = oracle.embed_backward(q_idx, dq.reshape(-1, Dw), np.zeros_like(Wnp))
    assert_bitexact(lay.blobs[0].diff, wd_ref)
