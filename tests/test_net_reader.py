"""a generated multi-layer net file runs unmodified.  The fixture is a hand-written
NetParameter text with network_v4's field values (examples/trec_qa_w2v_mms/do_trec_qa_clean.py:452-496; schema
src/caffe/proto/caffe.proto:63-110, 310-416).  CPU part: parsing, phase filtering, what is instantiated and what is
listed as skipped.  GPU part: the layers of the path run end to end out of the file -- HDF5Data -> Embed x2 (shared
table) -> SimCross (bilinear, M = 4, bias) forward and backward, and MRR / MAP / AUC on a host-filled `prob` --
against the oracle."""
import os

import numpy as np
import pytest

from conftest import ROOT
from util import TOL, assert_bitexact, assert_close, rng

FIXTURE = os.path.join(ROOT, "tests", "golden", "network_v4_test.prototxt")
PATH_TYPES = {"HDF5Data", "Embed", "SimCross", "SimMatrix", "PairRankLoss", "MAP", "MRR", "AUC", "RankAccuracy"}


@pytest.fixture(scope="module")
def L():
    from mms_answer_selection_amd import layers
    layers.lib()
    return layers


def test_whole_net_parses_and_lists_every_layer(L):
    text = open(FIXTURE).read()
    net = L.Net(text, phase="TEST")
    assert net.name == "qa-test-net"
    lay = net.layers
    names = [n for n, *_ in lay]
    # the TRAIN-only layer is filtered out in the TEST phase (net.cpp:272-296), everything else is listed in file order
    assert "train_only_probe" not in names and names[0] == "question" and names[-1] == "auc" and len(lay) == 24
    for n, t, sup, runnable, why in lay:
        assert sup == (t in PATH_TYPES), (n, t)
        assert not runnable                                  # nothing has been set up yet
    skipped = sorted({t for _, t, sup, _, _ in lay if not sup})
    assert skipped == ["BN", "Concat", "Convolution", "Dropout", "Flatten", "InnerProduct", "Pooling", "Softmax",
                       "SoftmaxWithLoss", "TanH"]
    # in-place tops share their bottom's blob; every named blob exists exactly once
    bl = net.blob_names
    assert len(bl) == len(set(bl)) and bl.count("pool0") == 1 and {"question", "w2v_q", "sim_cross", "prob", "map"} <= set(bl)
    train = L.Net(text, phase="TRAIN")
    assert [n for n, *_ in train.layers][-1] == "train_only_probe"


def test_net_reader_rejects_malformed_text(L):
    for bad in ("layer { name: \"x\" }",                          # no type
                "layer { name: \"x\" type: \"SimCross\" sim_cross_param { no_such_field: 1 } }",
                "layer { type: \"SimCross\" bottom: }",
                "no_such_net_field: 3",
                "layer { type: \"SimCross\" "):
        with pytest.raises(ValueError):
            L.Net(bad)


@pytest.mark.gpu
def test_network_v4_path_layers_run_from_the_net_file(L, oracle, tmp_path):
    B, Wd, D, V, groups = 12, 40, 50, 64, 3
    r = rng(44)
    question = r.integers(0, V, (2 * B, Wd)).astype(np.float64)
    answer = r.integers(0, V, (2 * B, Wd)).astype(np.float64)
    label = (r.uniform(size=(2 * B,)) < 0.3).astype(np.float64)
    label[:3] = [1, 0, 1]
    group = np.sort(r.integers(0, groups, 2 * B)).astype(np.float64)
    overlap = r.uniform(size=(2 * B, 2))
    h5 = tmp_path / "test.h5"
    L.write_h5(h5, dict(question=question, answer=answer, label=label, group=group, overlap_feat=overlap))
    src = tmp_path / "source.txt"
    src.write_text(str(h5) + "\n")
    net = L.Net(open(FIXTURE).read().replace("__SOURCE__", str(src)), phase="TEST")
    L.set_mode_gpu()
    L.set_random_seed(1701)

    # `prob` is produced by layers outside the library (fc2 -> Softmax): the host supplies it
    s = r.uniform(0.05, 0.95, B).astype(np.float32)
    prob = np.stack([1 - s, s], 1).astype(np.float32)
    pb = net.blob("prob")
    pb.reshape(B, 2)
    pb.data[...] = prob
    n_run = net.SetUp()
    state = {n: (run, why) for n, _, _, run, why in net.layers}
    for n in ("question", "w2v_q", "w2v_a", "sim_cross", "mrr", "map", "auc"):
        assert state[n][0], (n, state[n][1])
    assert n_run == 7
    assert not state["conv0"][0] and "not implemented" in state["conv0"][1]

    # the two Embed layers share `w2v-weights` / `w2v-bias` by name; SimCross's W and bias start at zero
    # (constant filler, do_trec_qa_clean.py:468 gives none): give them values so the products mean something
    eq, ea, sc = net.layer("w2v_q"), net.layer("w2v_a"), net.layer("sim_cross")
    table = eq.blobs[0].data.copy()
    assert table.shape == (V, D) and np.abs(table).max() <= 0.08 and table.std() > 0.03
    eq.blobs[0].data[3, :] = 0.5
    assert (ea.blobs[0].data[3, :] == 0.5).all(), "the answer Embed reads the SAME table"
    table = eq.blobs[0].data.copy()
    Wm = r.uniform(-0.08, 0.08, sc.blobs[0].shape).astype(np.float32)
    bias = r.uniform(-0.1, 0.1, sc.blobs[1].shape).astype(np.float32)
    assert Wm.shape == (4, D, D) and bias.shape == (4, Wd, Wd)
    sc.blobs[0].data[...] = Wm
    sc.blobs[1].data[...] = bias

    loss = net.Forward()
    assert loss == 0.0                                       # no loss layer of the library in this net
    # batch 0 of the file, in order (shuffle: false)
    assert_bitexact(net.blob("question").data, question[:B].astype(np.float32))
    assert_bitexact(net.blob("group").data.ravel(), group[:B].astype(np.float32))
    q_ref = table[question[:B].astype(int)]                  # bias is the constant 0
    a_ref = table[answer[:B].astype(int)]
    assert_bitexact(net.blob("w2v_q").data, q_ref)
    assert_bitexact(net.blob("w2v_a").data, a_ref)
    top_ref, _, _ = oracle.simcross_forward(2, q_ref, a_ref, Wm, bias)
    assert net.blob("sim_cross").shape == (B, 4, Wd, Wd)
    assert_close(net.blob("sim_cross").data, top_ref, TOL, "sim_cross")
    # the evaluation layers of the TEST net on the host-filled prob
    lab32, grp32 = label[:B].astype(np.float32), group[:B].astype(np.float32)
    m_ref, _ = oracle.map_score(prob, lab32, grp32)
    rr_ref, _ = oracle.mrr_score(prob, lab32, grp32)
    auc_ref = oracle.auc_score(prob, lab32)
    same = lambda x, y: np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32)
    assert same(net.blob("map").data.ravel()[0], m_ref)
    assert same(net.blob("mrr").data.ravel()[0], rr_ref)
    assert same(net.blob("auc").data.ravel()[0], auc_ref)

    # the same scores straight from the word ids ("fuse_embed_scoring": one launch, w2v_q / w2v_a never written),
    # with a non-zero Embed bias as after training: the bits of Embed, Embed, SimCross
    eq.blobs[1].data[...] = r.uniform(-0.2, 0.2, eq.blobs[1].shape).astype(np.float32)
    net2 = L.Net(open(FIXTURE).read().replace("__SOURCE__", str(src)), phase="TEST")
    pb2 = net2.blob("prob"); pb2.reshape(B, 2); pb2.data[...] = prob
    net2.set_option("fuse_embed_scoring", 1)
    assert net2.SetUp() == 7 and net2.num_fused == 1
    for name, bi in (("w2v_q", 0), ("w2v_q", 1), ("sim_cross", 0), ("sim_cross", 1)):
        net2.layer(name).blobs[bi].data[...] = net.layer(name).blobs[bi].data
    net2.blob("w2v_q").data[...] = -7.0
    net2.Forward()                                           # batch 0 of the file again (a fresh HDF5Data layer)
    assert (net2.blob("w2v_q").data == -7.0).all(), "the Embed top must not be written"
    net3 = L.Net(open(FIXTURE).read().replace("__SOURCE__", str(src)), phase="TEST")
    pb3 = net3.blob("prob"); pb3.reshape(B, 2); pb3.data[...] = prob
    assert net3.SetUp() == 7 and net3.num_fused == 0
    for name, bi in (("w2v_q", 0), ("w2v_q", 1), ("sim_cross", 0), ("sim_cross", 1)):
        net3.layer(name).blobs[bi].data[...] = net.layer(name).blobs[bi].data
    net3.Forward()
    assert np.abs(net3.blob("sim_cross").data - net.blob("sim_cross").data).max() > 1e-3   # the bias matters
    assert_bitexact(net2.blob("sim_cross").data, net3.blob("sim_cross").data)
    with pytest.raises(KeyError):
        net2.set_option("no_such_option", 1)
    eq.blobs[1].data[...] = 0

    # all forwards, then all backwards (net.cpp:581-591): dW / dbias of SimCross and the shared table's gradient
    dT = r.standard_normal(top_ref.shape).astype(np.float32)
    net.blob("sim_cross").diff[...] = dT
    for b in (sc.blobs[0], sc.blobs[1], eq.blobs[0], eq.blobs[1]):
        b.diff[...] = 0                                      # Net::ClearParamDiffs (net.cpp:923-941)
    net.Backward()
    dq_ref, da_ref, dW_ref, db_ref = oracle.simcross_backward(2, q_ref, a_ref, top_ref, dT, W=Wm, bias_term=True)
    assert_close(sc.blobs[0].diff, dW_ref, TOL, "dW")
    assert_close(sc.blobs[1].diff, db_ref, TOL, "dbias")
    assert_close(net.blob("w2v_q").diff, dq_ref, TOL, "dq")
    assert_close(net.blob("w2v_a").diff, da_ref, TOL, "da")
    # both Embed layers scatter into the ONE shared table diff (embed_layer.cpp:155-180)
    dtab = np.zeros_like(table)
    np.add.at(dtab, question[:B].astype(int).ravel(), dq_ref.reshape(-1, D))
    np.add.at(dtab, answer[:B].astype(int).ravel(), da_ref.reshape(-1, D))
    assert_close(eq.blobs[0].diff, dtab, 2e-4, "shared table diff")
    # The two Embed layers ran as a PAIR (one forward launch, one backward pass from the index the forward built: the net's
    # default, option "pair_embed"): the same net with the option off runs them one after the other -- same bits.
    net4 = L.Net(open(FIXTURE).read().replace("__SOURCE__", str(src)), phase="TEST")
    pb4 = net4.blob("prob"); pb4.reshape(B, 2); pb4.data[...] = prob
    net4.set_option("pair_embed", 0)
    assert net4.SetUp() == 7
    for name, bi in (("w2v_q", 0), ("w2v_q", 1), ("sim_cross", 0), ("sim_cross", 1)):
        net4.layer(name).blobs[bi].data[...] = net.layer(name).blobs[bi].data
    net4.Forward()
    assert_bitexact(net4.blob("w2v_q").data, q_ref)
    assert_bitexact(net4.blob("w2v_a").data, a_ref)
    net4.blob("sim_cross").diff[...] = dT
    e4, s4 = net4.layer("w2v_q"), net4.layer("sim_cross")
    for b in (s4.blobs[0], s4.blobs[1], e4.blobs[0], e4.blobs[1]):
        b.diff[...] = 0
    net4.Backward()
    assert_bitexact(e4.blobs[0].diff, eq.blobs[0].diff, "table diff: pair == one after the other")
    # (the bias gradient is a gemv in the reference, no defined order: the pair sums it over the concatenated rows)
    assert_close(e4.blobs[1].diff, eq.blobs[1].diff, TOL, "Embed bias diff: pair vs one after the other")
    # the next Forward serves the file's second batch
    net.Forward()
    assert_bitexact(net.blob("question").data, question[B:].astype(np.float32))


TRIPLET_NET = """
name: "triplet"
input: "q"   input_shape { dim: 33 dim: 1 dim: 300 }
input: "ap"  input_shape { dim: 33 dim: 1 dim: 300 }
input: "an"  input_shape { dim: 33 dim: 1 dim: 300 }
input: "y"   input_shape { dim: 33 dim: 1 }
layer { name: "sp" type: "SimCross" bottom: "q" bottom: "ap" top: "s_pos" sim_cross_param { dist_mode: 1 } }
layer { name: "sn" type: "SimCross" bottom: "q" bottom: "an" top: "s_neg" sim_cross_param { dist_mode: 1 } }
layer { name: "loss" type: "PairRankLoss" bottom: "s_pos" bottom: "s_neg" bottom: "y" top: "loss"
        pair_rank_loss_param { margin: 0.05 } }
"""


@pytest.mark.gpu
def test_blob_feeding_two_layers_gets_split_semantics(L, oracle):
    """Net::Init inserts a Split layer wherever a blob feeds more than one layer (insert_splits.cpp:13-88): each
    consumer back-propagates into a diff of its own and SplitLayer::Backward sums them (split_layer.cpp:38-57).
    `q` feeds both SimCross layers of the triplet net: its gradient is dq(pos) + dq(neg), in that order -- not the
    last writer's."""
    N, D = 33, 300
    r = rng(91)
    q = (r.standard_normal((N, 1, D)) * 0.4).astype(np.float32)
    ap = (q + 0.1 * r.standard_normal((N, 1, D))).astype(np.float32)
    an = (r.standard_normal((N, 1, D)) * 0.4).astype(np.float32)
    y = (r.uniform(size=(N, 1)) < 0.8).astype(np.float32)
    L.set_mode_gpu()
    net = L.Net(TRIPLET_NET, phase="TRAIN")
    for name, v in (("q", q), ("ap", ap), ("an", an), ("y", y)):
        net.blob(name).data[...] = v
    assert net.SetUp() == 3
    assert net.num_splits == 1
    loss = net.Forward()
    sp, _, _ = oracle.simcross_forward(1, q, ap)
    sn, _, _ = oracle.simcross_forward(1, q, an)
    loss_ref, o, s = oracle.pairrank_forward(sp.reshape(N, 1), sn.reshape(N, 1), y, 0.05)
    assert_bitexact(net.blob("s_pos").data.ravel(), sp.ravel())
    assert_close(loss, loss_ref, TOL, "loss")
    net.Backward()
    gsp, gsn = oracle.pairrank_backward(y, o, s, top_diff=1.0)
    dq_p, dap_ref, _, _ = oracle.simcross_backward(1, q, ap, sp, gsp.reshape(sp.shape))
    dq_n, dan_ref, _, _ = oracle.simcross_backward(1, q, an, sn, gsn.reshape(sn.shape))
    assert np.abs(dq_n).max() > 0 and np.abs(dq_p).max() > 0
    assert_bitexact(net.blob("q").diff, dq_p + dq_n, "dq = Split backward of the two branches")
    assert_bitexact(net.blob("ap").diff, dap_ref, "da_pos")
    assert_bitexact(net.blob("an").diff, dan_ref, "da_neg")
    # a second iteration (Reshape + ShareData of the split tops every Forward) gives the same bits
    net.Forward()
    net.Backward()
    assert_bitexact(net.blob("q").diff, dq_p + dq_n, "second iteration")


@pytest.mark.gpu
def test_split_backward_abi_orders_its_sum(L):
    """mms_split_backward_f32: ((t0 + t1) + t2) + ..., any number of tops, in place on t0 allowed."""
    import ctypes as C
    import torch
    from mms_answer_selection_amd import capi
    r = rng(5)
    for ntop in (1, 2, 3, 11):
        tops = [(r.standard_normal(1000) * 10.0 ** r.integers(-3, 4)).astype(np.float32) for _ in range(ntop)]
        want = tops[0].copy()
        for t in tops[1:]:
            want = want + t
        dev = [torch.from_numpy(t).cuda() for t in tops]
        out = torch.full((1000,), float("nan"), device="cuda")
        arr = (C.c_void_p * ntop)(*[d.data_ptr() for d in dev])
        capi.check(capi.lib().mms_split_backward_f32(1000, ntop, arr, out.data_ptr(), None), "split")
        torch.cuda.synchronize()
        assert_bitexact(out.cpu().numpy(), want, "ntop=%d" % ntop)
    capi.check(capi.lib().mms_split_backward_f32(1000, ntop, arr, dev[0].data_ptr(), None), "split in place")
    torch.cuda.synchronize()
    assert_bitexact(dev[0].cpu().numpy(), want, "in place on top 0")
