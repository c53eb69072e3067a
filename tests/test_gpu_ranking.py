"""GPU parity for the ranking metrics: MAP / MRR / AUC /
RankAccuracy through the C ABI against the oracle's std::map + std::sort
restatement -- bit-exact, because the sequential float walks are reproduced --
and the end-to-end statement of BASELINE.json: ranking output computed from
GPU scores is identical to the one computed from CPU scores."""
import os

import numpy as np
import pytest
import torch

from util import assert_bitexact, qa, rng

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def split(r, n, groups, pos_rate=0.17):
    group = np.sort(r.integers(0, groups, n)).astype(np.float32)
    label = (r.uniform(size=n) < pos_rate).astype(np.float32)
    # DISTINCT scores: the reference's order among equal scores is implementation-defined
    # (unstable std::sort), so bit-exactness is only defined without cross-label ties
    score = ((r.permutation(n) + 0.5) / n).astype(np.float32)
    assert np.unique(score).size == n
    return group, label, score


def same_bits(x, y):
    return np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32) or (np.isnan(x) and np.isnan(y))


def test_golden_trec_split(oracle, hiplib):
    from mms_answer_selection_amd import capi
    g = dict(np.load(os.path.join(GOLD, "ranking_1517_68.npz")))
    m, rr, eff = capi.rank_map_mrr(dev(g["prob"]), dev(g["label"]), dev(g["group"]))
    assert same_bits(m, g["map"]) and same_bits(rr, g["mrr"]) and eff == int(g["effective"])
    assert same_bits(capi.rank_auc(dev(g["prob"]), dev(g["label"])), g["auc"])


@pytest.mark.parametrize("cfg", [(1517, 68), (1148, 65), (37, 5), (1, 1), (20000, 700),
                                 (512, 20), (513, 20), (64, 3), (65, 64)])   # either side of the one-workgroup path's limit and of its padded sizes
def test_map_mrr_auc_random(cfg, oracle, hiplib):
    from mms_answer_selection_amd import capi
    n, groups = cfg
    r = rng(n)
    group, label, score = split(r, n, groups)
    perm = r.permutation(n)                     # buckets arrive interleaved, as from a shuffled net
    group, label, score = group[perm], label[perm], score[perm]
    group -= 3                                   # negative group ids order like std::map<int>
    prob = np.stack([1 - score, score], 1).astype(np.float32)
    m_ref, eff_ref = oracle.map_score(prob, label, group)
    rr_ref, _ = oracle.mrr_score(prob, label, group)
    m, rr, eff = capi.rank_map_mrr(dev(prob), dev(label), dev(group))
    assert eff == eff_ref
    assert same_bits(m, m_ref), (m, m_ref)
    assert same_bits(rr, rr_ref), (rr, rr_ref)
    assert same_bits(capi.rank_auc(dev(prob), dev(label)), oracle.auc_score(prob, label))


@pytest.mark.parametrize("shape,axis", [((37, 2), 1), ((11, 2, 5), 1), ((6, 3, 4, 5), 1), ((2, 7, 3), 2), ((3, 4), 0)])
def test_auc_any_label_axis_and_ignore_label(shape, axis, oracle, hiplib):
    """AUCLayer with inner_num > 1 and a label axis other than 1 (auc_layer.cpp:26-33, 66-77), plus ignore_label
    (:69-71): the C ABI and the Layer mirror against the oracle's general indexing, bit for bit."""
    from mms_answer_selection_amd import capi
    r = rng(sum(shape) + axis)
    prob = r.uniform(0.01, 0.99, shape).astype(np.float32)          # distinct scores
    lshape = shape[:axis] + shape[axis + 1:]
    label = (r.uniform(size=lshape) < 0.4).astype(np.float32)
    label.reshape(-1)[:2] = [1, 0]
    C = shape[axis]
    for fixed_axis in sorted({0, C - 1}):
        ref = oracle.auc_score_nd(prob, label, axis=axis, fixed_axis=fixed_axis)
        got = capi.rank_auc_nd(dev(prob), dev(label), axis=axis, fixed_axis=fixed_axis)
        assert same_bits(got, ref), (got, ref)
    # ignore_label: labels of 2 are dropped before counting
    lab2 = label.copy()
    lab2.reshape(-1)[3::5] = 2
    ref = oracle.auc_score_nd(prob, lab2, axis=axis, fixed_axis=C - 1, ignore_label=2)
    got = capi.rank_auc_nd(dev(prob), dev(lab2), axis=axis, fixed_axis=C - 1, ignore_label=2)
    assert same_bits(got, ref), (got, ref)
    # through the Layer mirror (auc_param { axis fixed_axis ignore_label })
    from mms_answer_selection_amd import layers as L
    L.lib()
    L.set_mode_gpu()
    lay = L.AUC(axis=axis, fixed_axis=C - 1, ignore_label=2)
    bp, bl, top = L.Blob(shape), L.Blob(lshape), L.Blob()
    bp.data[...] = prob
    bl.data[...] = lab2
    lay.SetUp([bp, bl], [top])
    lay.Forward([bp, bl], [top])
    assert same_bits(top.data.reshape(-1)[0], ref)
    if len(shape) == 2 and axis == 1:                        # the (N, C) case agrees with the original entry point
        assert same_bits(capi.rank_auc(dev(prob), dev(label), fixed_axis=C - 1),
                         oracle.auc_score_nd(prob, label, axis=1, fixed_axis=C - 1))


def test_skipped_buckets_and_empty_result(oracle, hiplib):
    from mms_answer_selection_amd import capi
    # all-positive and all-negative buckets are skipped (map_layer.cpp:90-92); nothing left -> NaN
    group = np.array([0, 0, 1, 1, 1], np.float32)
    label = np.array([1, 1, 0, 0, 0], np.float32)
    prob = np.stack([np.zeros(5), np.linspace(0.1, 0.9, 5)], 1).astype(np.float32)
    m, rr, eff = capi.rank_map_mrr(dev(prob), dev(label), dev(group))
    assert eff == 0 and np.isnan(m) and np.isnan(rr)
    m_ref, eff_ref = oracle.map_score(prob, label, group)
    assert eff_ref == 0 and np.isnan(m_ref)
    # AUC with no positive is 0 (auc_layer.cpp:127-134)
    assert capi.rank_auc(dev(prob), dev(np.zeros(5, np.float32))) == 0.0


@pytest.mark.parametrize("seed", range(6))
def test_cross_label_ties_in_groups_of_at_most_16_follow_the_reference(seed, oracle, hiplib):
    """std::sort on at most 16 elements is libstdc++'s insertion sort, which is stable: for candidate groups that
    small the reference's order among EQUAL scores is the input order -- the library's own tie rule -- so MAP and
    MRR carry the reference's bits even when tied scores have different labels (map_layer.cpp:76, mrr_layer.cpp:57)."""
    from mms_answer_selection_amd import capi
    r = rng(400 + seed)
    ng = [1, 7, 60, 300, 900, 40][seed]
    sizes = r.integers(1, 17, ng)
    gid = np.repeat(np.arange(ng), sizes).astype(np.float32) - 2
    n = gid.size
    gid = gid[r.permutation(n)]
    label = (r.uniform(size=n) < 0.4).astype(np.float32)
    score = (np.round(r.uniform(0, 1, n) * 4) / 4).astype(np.float32)        # five score levels: ties everywhere
    prob = np.stack([1 - score, score], 1).astype(np.float32)
    m_ref, eff_ref = oracle.map_score(prob, label, gid)
    rr_ref, _ = oracle.mrr_score(prob, label, gid)
    m, rr, eff = capi.rank_map_mrr(dev(prob), dev(label), dev(gid))
    assert eff == eff_ref
    assert same_bits(m, m_ref) or (np.isnan(m) and np.isnan(m_ref)), (m, m_ref)
    assert same_bits(rr, rr_ref) or (np.isnan(rr) and np.isnan(rr_ref)), (rr, rr_ref)


@pytest.mark.parametrize("n,ngroups,levels", [(300, 6, 4), (1517, 70, 8), (1517, 5, 3), (6000, 40, 16), (700, 1, 5), (513, 30, 2)])
def test_cross_label_ties_libstdcxx_mode(n, ngroups, levels, oracle, hiplib):
    """MMS_RANK_TIES_LIBSTDCXX: candidate groups of MORE than 16 items with equal scores across labels -- MAP, MRR and
    AUC carry the bits of the oracle build (g++'s std::sort), where the default input-order rule does not."""
    from mms_answer_selection_amd import capi
    r = rng(n + ngroups + levels)
    group = r.integers(0, ngroups, n).astype(np.float32) - 1
    label = (r.uniform(size=n) < 0.35).astype(np.float32)
    score = (np.round(r.uniform(0, 1, n) * levels) / levels).astype(np.float32)
    prob = np.stack([1 - score, score], 1).astype(np.float32)
    m_ref, eff_ref = oracle.map_score(prob, label, group)
    rr_ref, _ = oracle.mrr_score(prob, label, group)
    auc_ref = oracle.auc_score(prob, label)
    lab_ign = label.copy(); lab_ign[r.uniform(size=n) < 0.2] = 7.0
    auc_ign_ref = oracle.auc_score_nd(prob, lab_ign, axis=1, fixed_axis=1, ignore_label=7)
    capi.set_rank_tie_mode("libstdcxx")
    try:
        m, rr, eff = capi.rank_map_mrr(dev(prob), dev(label), dev(group))
        auc = capi.rank_auc(dev(prob), dev(label))
        auc_ign = capi.rank_auc(dev(prob), dev(lab_ign), ignore_label=7)
    finally:
        capi.set_rank_tie_mode("input")
    assert eff == eff_ref
    assert same_bits(m, m_ref), (m, m_ref)
    assert same_bits(rr, rr_ref), (rr, rr_ref)
    assert same_bits(auc, auc_ref), (auc, auc_ref)
    assert same_bits(auc_ign, auc_ign_ref), (auc_ign, auc_ign_ref)
    assert hiplib.mms_set_rank_tie_mode(9) == 1 and hiplib.mms_get_rank_tie_mode() == 0
    # distinct scores: the mode changes nothing
    score2 = ((r.permutation(n) + 0.5) / n).astype(np.float32)
    prob2 = np.stack([1 - score2, score2], 1).astype(np.float32)
    base = capi.rank_map_mrr(dev(prob2), dev(label), dev(group))
    capi.set_rank_tie_mode("libstdcxx")
    try:
        again = capi.rank_map_mrr(dev(prob2), dev(label), dev(group))
    finally:
        capi.set_rank_tie_mode("input")
    assert same_bits(base[0], again[0]) and same_bits(base[1], again[1]) and base[2] == again[2]


def test_ties_with_equal_labels_are_order_independent(oracle, hiplib):
    """Equal scores: the reference's order is implementation-defined (unstable sort);
    when the tied items share a label every order gives the same metric."""
    from mms_answer_selection_amd import capi
    group = np.zeros(8, np.float32)
    score = np.array([.9, .5, .5, .5, .3, .3, .1, .05], np.float32)
    label = np.array([0, 1, 1, 1, 0, 0, 1, 0], np.float32)
    prob = np.stack([1 - score, score], 1).astype(np.float32)
    m, rr, _ = capi.rank_map_mrr(dev(prob), dev(label), dev(group))
    m_ref, _ = oracle.map_score(prob, label, group)
    rr_ref, _ = oracle.mrr_score(prob, label, group)
    assert same_bits(m, m_ref) and same_bits(rr, rr_ref)


def test_rank_accuracy(oracle, hiplib):
    from mms_answer_selection_amd import capi
    r = rng(11)
    n = 5000
    a = r.uniform(size=n).astype(np.float32)
    b = r.uniform(size=n).astype(np.float32)
    lab = r.choice([-1.0, 0.0, 1.0], n).astype(np.float32)
    assert same_bits(capi.rank_accuracy(dev(a), dev(b), dev(lab)), oracle.rank_accuracy(a, b, lab))


def test_end_to_end_ranking_identical_to_cpu(oracle, hiplib):
    """cfg 4 in miniature: score a TREC-QA-sized split (1517 candidates, 68 questions,
    sentence vectors) with the HIP SimCross, rank on the GPU, and compare with the CPU
    pipeline (oracle scores -> oracle metrics): scores, argsort, MAP, MRR, AUC all identical."""
    from mms_answer_selection_amd import capi
    n, groups, D = 1517, 68, 300
    r = rng(4)
    group, label, _ = split(r, n, groups)
    qvec = (r.standard_normal((groups, D)) * 0.4).astype(np.float32)
    q = qvec[group.astype(int)].reshape(n, 1, D)
    a = (q + r.standard_normal((n, 1, D)).astype(np.float32) * np.where(label, 0.2, 0.4).reshape(n, 1, 1)
         ).astype(np.float32)
    top_ref, _, _ = oracle.simcross_forward(1, q, a)
    top = torch.empty(n, 1, 1, 1, device="cuda")
    capi.simcross_forward(1, dev(q), dev(a), top)
    s_gpu = top.view(n)
    assert_bitexact(s_gpu.cpu().numpy(), top_ref.reshape(n), "scores")
    assert (np.argsort(-s_gpu.cpu().numpy(), kind="stable") == np.argsort(-top_ref.reshape(n), kind="stable")).all()
    prob_gpu = torch.stack([1 - s_gpu, s_gpu], 1).contiguous()
    prob_ref = np.stack([1 - top_ref.reshape(n), top_ref.reshape(n)], 1).astype(np.float32)
    m, rr, eff = capi.rank_map_mrr(prob_gpu, dev(label), dev(group))
    m_ref, eff_ref = oracle.map_score(prob_ref, label, group)
    rr_ref, _ = oracle.mrr_score(prob_ref, label, group)
    assert eff == eff_ref and same_bits(m, m_ref) and same_bits(rr, rr_ref)
    assert same_bits(capi.rank_auc(prob_gpu, dev(label)), oracle.auc_score(prob_ref, label))
    assert m > 0.5          # the synthetic positives really are closer
