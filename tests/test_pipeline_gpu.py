"""The path with its callers and data formats on either side, end to end through the Caffe layer
mirror, as the driver's TEST net wires them (do_trec_qa_clean.py:380, 461-468, 495):

    train_h5/*.h5 --HDF5Data--> question, answer, label, group
    question, answer --Embed (GloVe text table, shared by name in the driver)--> (B, L, Dw)
    --SimCross (Euclid)--> (B, 1, L, L) word-grid scores      [cfg 1 / cfg 4 geometry]
    and sentence vectors --SimCross--> (B, 1, 1, 1) --> prob --MAP / MRR layers--> ranking metrics

against the same chain composed from the oracle on the host.  Everything order-defined is compared
bit for bit: what the feed delivers, the gathered embeddings, every score, MAP and MRR."""
import numpy as np
import pytest

from util import assert_bitexact, rng

pytestmark = pytest.mark.gpu


def test_feed_embed_simcross_ranking_pipeline(tmp_path, oracle, hiplib):
    from mms_answer_selection_amd import layers as L
    L.lib()
    L.set_mode_gpu()
    r = rng(2024)
    V, Dw, Lw = 60, 50, 40                      # vocabulary, embedding dim, words per sentence
    n, groups = 150, 12                         # candidates, questions
    B = 50                                      # the reference's TRAIN batch size

    # --- data on disk, in the driver's formats -------------------------------------------------
    vecs = r.uniform(-1, 1, (V, Dw)).astype(np.float32)
    glove = tmp_path / "glove.txt"
    glove.write_text("".join("w%d %s\n" % (i, " ".join("%.6f" % v for v in vecs[i])) for i in range(V)))
    table = np.zeros((V + 2, Dw), np.float32)   # +2 rows: unknown word, zero pad (do_trec_qa_clean.py:297-299)
    table[:V] = np.array([[np.float32("%.6f" % v) for v in row] for row in vecs], np.float32)
    pad = V + 1
    group = np.sort(r.integers(0, groups, n)).astype(np.float64)
    label = (r.uniform(size=n) < 0.3).astype(np.float64)
    qwords = r.integers(0, V, (groups, Lw))
    question = qwords[group.astype(int)].astype(np.float64)
    answer = r.integers(0, V, (n, Lw)).astype(np.float64)
    answer[label > 0, :20] = question[label > 0, :20]          # positives share words with the question
    for row in (question, answer):
        row[:, 30:] = pad                                        # padded tails, as in a real batch
    files = []
    for i, (lo, hi) in enumerate(((0, 70), (70, 150))):         # two h5 "patches"; 70 is not a multiple of B
        p = tmp_path / ("data%d.h5" % i)
        L.write_h5(p, {"question": question[lo:hi], "answer": answer[lo:hi], "label": label[lo:hi],
                       "group": group[lo:hi]})
        files.append(str(p))
    src = tmp_path / "test.txt"
    src.write_text("\n".join(files) + "\n")

    # --- the layers ----------------------------------------------------------------------------
    feed = L.HDF5Data(top=["question", "answer", "label", "group"], batch_size=B, source=str(src), shuffle=0)
    tq, ta, tl, tg = L.Blob(), L.Blob(), L.Blob(), L.Blob()
    feed.SetUp([], [tq, ta, tl, tg])
    emb_q = L.Embed(input_dim=V + 2, num_output=Dw, bias_term=False, weight_source=str(glove))
    emb_a = L.Embed(input_dim=V + 2, num_output=Dw, bias_term=False, weight_source=str(glove))
    eq, ea = L.Blob(), L.Blob()
    emb_q.SetUp([tq], [eq])
    emb_a.SetUp([ta], [ea])
    assert_bitexact(emb_q.blobs[0].data, table, "Embed table loaded from the GloVe text file")
    grid = L.SimCross(dist_mode=1)
    tgrid = L.Blob()
    grid.SetUp([eq, ea], [tgrid])

    scores, labels, grps = [], [], []
    for it in range(3):                                          # 150 candidates = 3 batches of 50
        feed.Forward([], [tq, ta, tl, tg])
        lo = it * B
        assert_bitexact(tq.data, question[lo:lo + B].astype(np.float32), "fed question ids")
        assert_bitexact(tg.data, group[lo:lo + B].astype(np.float32), "fed group ids")
        emb_q.Forward([tq], [eq])
        emb_a.Forward([ta], [ea])
        q_ref = oracle.embed_forward(tq.data.copy(), table)
        a_ref = oracle.embed_forward(ta.data.copy(), table)
        assert_bitexact(eq.data, q_ref, "w2v_q")
        grid.Forward([eq, ea], [tgrid])
        grid_ref, _, _ = oracle.simcross_forward(1, q_ref, a_ref)
        assert tgrid.data.shape == (B, 1, Lw, Lw)
        assert_bitexact(tgrid.data, grid_ref, "word-grid scores")
        # the same grid straight from the fed word ids: Embed fused into SimCross's loads (C ABI)
        import torch
        from mms_answer_selection_amd import capi
        ids_q = torch.from_numpy(tq.data.reshape(B, Lw).copy()).cuda()
        ids_a = torch.from_numpy(ta.data.reshape(B, Lw).copy()).cuda()
        fused = torch.empty(B, 1, Lw, Lw, device="cuda")
        capi.embed_simcross_forward(1, ids_q, ids_a, torch.from_numpy(table).cuda(), fused)
        assert_bitexact(fused.cpu().numpy(), grid_ref, "word-grid scores from word ids (fused call)")
        # sentence vectors (sum of word vectors in d-order on the host) -> one score per candidate
        qs = q_ref.sum(axis=1, dtype=np.float32).reshape(B, 1, Dw)
        as_ = a_ref.sum(axis=1, dtype=np.float32).reshape(B, 1, Dw)
        sent = L.SimCross(dist_mode=1)
        bq, ba, bt = L.Blob(qs.shape), L.Blob(as_.shape), L.Blob()
        bq.data[...] = qs
        ba.data[...] = as_
        sent.SetUp([bq, ba], [bt])
        sent.Forward([bq, ba], [bt])
        s_ref, _, _ = oracle.simcross_forward(1, qs, as_)
        assert_bitexact(bt.data, s_ref, "sentence scores")
        scores.append(bt.data.reshape(B).copy())
        labels.append(tl.data.reshape(B).copy())
        grps.append(tg.data.reshape(B).copy())
    s = np.concatenate(scores)
    lab = np.concatenate(labels)
    grp = np.concatenate(grps)
    assert (lab == label.astype(np.float32)).all() and (grp == group.astype(np.float32)).all()

    # --- ranking layers on the accumulated scores (the driver evaluates the whole split) --------
    prob = np.stack([1 - s, s], 1).astype(np.float32)
    bp, bl, bg = L.Blob(prob.shape), L.Blob((n,)), L.Blob((n,))
    bp.data[...] = prob
    bl.data[...] = lab
    bg.data[...] = grp
    for make, ref in ((L.MAP, oracle.map_score), (L.MRR, oracle.mrr_score)):
        lay = make()
        out = L.Blob()
        lay.SetUp([bp, bl, bg], [out])
        lay.Forward([bp, bl, bg], [out])
        want, _ = ref(prob, lab, grp)
        assert np.float32(out.data.ravel()[0]).view(np.uint32) == np.float32(want).view(np.uint32)
    # a fourth batch wraps around to the first file (hdf5_data_layer.cpp:127-144)
    feed.Forward([], [tq, ta, tl, tg])
    assert_bitexact(tq.data, question[:B].astype(np.float32), "wrap-around")
