"""dev-only soak: MAP / MRR / AUC / RankAccuracy and the Embed forward / backward against the oracle, bit for bit,
over random sizes and group structures (groups without positives, single-candidate groups, shuffled bucket order,
negative group ids) with DISTINCT scores -- among EQUAL scores the reference's order is whatever its unstable
std::sort leaves (include/mms.h: here ties keep the original order), so bit-exactness is only defined without
cross-label ties; RankAccuracy (no sort) is checked with heavily tied inputs; Embed with repeated ids and a heavy
zero-pad id."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mms_answer_selection_amd import capi
from oracle import cpu_oracle as O
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
bits = lambda x: np.float32(x).view(np.uint32)
bad = 0; checks = 0; t0 = time.time()
def chk(ok, what):
    global bad, checks
    checks += 1
    if not ok:
        bad += 1
        print("MISMATCH", what)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    r = np.random.default_rng(9000 + seed)
    n = int(r.choice([1, 2, 7, 50, 511, 512, 513, 1517, 5000, 20000]))
    ngroups = max(1, int(n * r.choice([0.02, 0.1, 0.5, 1.0])))
    group = r.integers(0, ngroups, n).astype(np.float32) - 5
    label = (r.uniform(size=n) < r.choice([0.05, 0.3, 0.9])).astype(np.float32)
    q = int(r.choice([4, 16, 1000, 0]))                 # quantisation of the RankAccuracy inputs: ties
    score = ((r.permutation(n) + r.uniform(0.05, 0.95)) / n) ** float(r.choice([0.25, 1.0, 4.0]))   # distinct, skewed either way
    assert np.unique(score.astype(np.float32)).size == n
    prob = np.stack([1 - score, score], 1).astype(np.float32)
    tag = "seed %d n %d groups %d q %d" % (seed, n, ngroups, q)
    m_ref, eff_ref = O.map_score(prob, label, group)
    rr_ref, _ = O.mrr_score(prob, label, group)
    m, rr, eff = capi.rank_map_mrr(dev(prob), dev(label), dev(group))
    chk(eff == eff_ref and (bits(m) == bits(m_ref) or (np.isnan(m) and np.isnan(m_ref))) and
        (bits(rr) == bits(rr_ref) or (np.isnan(rr) and np.isnan(rr_ref))), tag + " map/mrr %r %r vs %r %r" % (m, rr, m_ref, rr_ref))
    au, au_ref = capi.rank_auc(dev(prob), dev(label)), O.auc_score(prob, label)
    chk(bits(au) == bits(au_ref) or (np.isnan(au) and np.isnan(au_ref)), tag + " auc %r vs %r" % (au, au_ref))
    a, b = r.uniform(0, 1, n).astype(np.float32), r.uniform(0, 1, n).astype(np.float32)
    if q: a = (np.round(a * q) / q).astype(np.float32); b = (np.round(b * q) / q).astype(np.float32)
    ra, ra_ref = capi.rank_accuracy(dev(a), dev(b), dev(label)), O.rank_accuracy(a, b, label)
    chk(bits(ra) == bits(ra_ref), tag + " rank_accuracy %r vs %r" % (ra, ra_ref))
    # TIED scores across labels, candidate groups of at most 16: libstdc++'s std::sort is then its (stable)
    # insertion sort, so the reference's order among equal scores IS the input order -- the library's tie rule
    ng = int(r.choice([1, 5, 100, 400]))
    sizes = r.integers(1, 17, ng)
    gid = np.repeat(np.arange(ng), sizes).astype(np.float32); n2 = gid.size
    perm2 = r.permutation(n2)
    gid = gid[perm2]
    lab2 = (r.uniform(size=n2) < 0.4).astype(np.float32)
    sc2 = (np.round(r.uniform(0, 1, n2) * int(r.choice([2, 4, 8]))) / 8).astype(np.float32)
    prob2 = np.stack([1 - sc2, sc2], 1).astype(np.float32)
    m_ref, eff_ref = O.map_score(prob2, lab2, gid)
    rr_ref, _ = O.mrr_score(prob2, lab2, gid)
    m, rr, eff = capi.rank_map_mrr(dev(prob2), dev(lab2), dev(gid))
    chk(eff == eff_ref and (bits(m) == bits(m_ref) or (np.isnan(m) and np.isnan(m_ref))) and
        (bits(rr) == bits(rr_ref) or (np.isnan(rr) and np.isnan(rr_ref))), "tied, groups <= 16: seed %d n %d: %r %r vs %r %r" % (seed, n2, m, rr, m_ref, rr_ref))
    # TIED scores across labels in groups of any size, MMS_RANK_TIES_LIBSTDCXX: the oracle build's order (g++'s std::sort)
    n3 = int(r.choice([40, 600, 1517, 5000])); g3 = r.integers(0, max(1, n3 // int(r.choice([5, 30, 300]))), n3).astype(np.float32)
    lab3 = (r.uniform(size=n3) < 0.4).astype(np.float32)
    sc3 = (np.round(r.uniform(0, 1, n3) * int(r.choice([1, 3, 10, 50]))) / 50).astype(np.float32)
    prob3 = np.stack([1 - sc3, sc3], 1).astype(np.float32)
    m_ref, eff_ref = O.map_score(prob3, lab3, g3); rr_ref, _ = O.mrr_score(prob3, lab3, g3); au_ref = O.auc_score(prob3, lab3)
    capi.set_rank_tie_mode("libstdcxx")
    m, rr, eff = capi.rank_map_mrr(dev(prob3), dev(lab3), dev(g3)); au = capi.rank_auc(dev(prob3), dev(lab3))
    capi.set_rank_tie_mode("input")
    same = lambda x, y: bits(x) == bits(y) or (np.isnan(x) and np.isnan(y))
    chk(eff == eff_ref and same(m, m_ref) and same(rr, rr_ref) and same(au, au_ref),
        "tied, libstdcxx mode: seed %d n %d: %r %r %r vs %r %r %r" % (seed, n3, m, rr, au, m_ref, rr_ref, au_ref))
    # Embed
    M = int(r.choice([1, 40, 2000, 4000, 4097, 30000])); K = int(r.choice([2, 50, 3000, 20000])); N = int(r.choice([1, 50, 64, 65, 300]))
    idx = r.integers(0, K, M)
    idx[r.uniform(size=M) < r.choice([0.0, 0.6])] = r.integers(0, K)       # a heavy id (the zero-pad word)
    index = idx.astype(np.float32)
    w = r.standard_normal((K, N)).astype(np.float32); bias = r.standard_normal(N).astype(np.float32)
    top_ref = O.embed_forward(index, w, bias)
    top = torch.empty(M, N, device="cuda")
    capi.embed_forward(dev(index), dev(w), top, bias=dev(bias))
    chk((top.cpu().numpy().view(np.uint32) == top_ref.reshape(M, N).view(np.uint32)).all(), "embed fwd seed %d" % seed)
    dT = r.standard_normal((M, N)).astype(np.float32)
    wd0 = r.standard_normal((K, N)).astype(np.float32)
    wd_ref, _ = O.embed_backward(index, dT, wd0)
    wd = dev(wd0.copy())
    capi.embed_backward(dev(index), dev(dT), wd)
    chk((wd.cpu().numpy().view(np.uint32) == wd_ref.view(np.uint32)).all(), "embed bwd seed %d M %d K %d N %d" % (seed, M, K, N))
print("checks %d, mismatches %d, %.1f s" % (checks, bad, time.time() - t0))
