"""Worker of tests/test_sharded_gpu.py (launched by torch.distributed.run, world 2, gloo):
cfg 4 in miniature -- every rank scores ITS shard of a TREC-QA-sized candidate set with the HIP
SimCross kernel, the per-pair scores are all-gathered, and every rank ranks the full set on its GPU;
rank 0 compares scores, MAP and MRR with the CPU chain (oracle) bit for bit.  Exits non-zero on
any mismatch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def bilinear_and_loss():
    """Each rank runs the HIP kernels on ITS shard: SimCross bilinear backward (dW, dbias all-reduced and compared
    with the unsharded oracle on rank 0) and the fused triplet step with the shard's loss weight
    (sharded.shard_loss_weight: gradients and loss must equal the unsharded HIP run)."""
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from mms_answer_selection_amd import capi, sharded
    from oracle import cpu_oracle as O
    ok = True
    # --- bilinear backward, pair-sharded ---
    n, W1, W2, D, M = 37, 5, 7, 52, 3               # ragged over 2 ranks
    r = np.random.default_rng(11)
    q = (r.standard_normal((n, W1, D)) * 0.4).astype(np.float32)
    a = (r.standard_normal((n, W2, D)) * 0.4).astype(np.float32)
    Wt = r.uniform(-0.08, 0.08, (M, D, D)).astype(np.float32)
    bias = r.uniform(-0.1, 0.1, (M, W1, W2)).astype(np.float32)
    dT = r.standard_normal((n, M, W1, W2)).astype(np.float32)
    lo, hi = sharded.shard_range(n, rank, world)
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    qd, ad, Wd, bd, dTd = d(q[lo:hi]), d(a[lo:hi]), d(Wt), d(bias), d(dT[lo:hi])
    top = torch.empty(hi - lo, M, W1, W2, device="cuda")
    capi.simcross_forward(2, qd, ad, top, W=Wd, bias=bd)
    dq, da = torch.empty_like(qd), torch.empty_like(ad)
    dW, db = torch.empty_like(Wd), torch.zeros_like(bd)
    capi.simcross_backward(2, qd, ad, top, dTd, dq, da, W=Wd, bias_term=True, dW=dW, dbias=db)
    dWh, dbh = dW.cpu(), db.cpu()                    # gloo moves host tensors
    sharded.all_reduce_param_grads([dWh, dbh])
    if rank == 0:
        top_ref, _, _ = O.simcross_forward(2, q, a, Wt, bias)
        dq_ref, da_ref, dW_ref, db_ref = O.simcross_backward(2, q, a, top_ref, dT, W=Wt, bias_term=True)
        tol = lambda x, y: np.abs(x - y).max() <= 1e-5 * max(1.0, np.abs(y).max())
        ok = ok and tol(dWh.numpy(), dW_ref) and tol(dbh.numpy(), db_ref)
        ok = ok and tol(dq.cpu().numpy(), dq_ref[lo:hi]) and tol(da.cpu().numpy(), da_ref[lo:hi])
        print("sharded bilinear backward: dW/dbias all-reduced == unsharded oracle: %s" % ok)
    # --- fused triplet step on shards with the shard's loss weight ---
    N, Dv = 513, 300
    g = torch.Generator(device="cuda").manual_seed(3)   # the same triplets on every rank
    tq = torch.randn(N, 1, Dv, device="cuda", generator=g) * 0.4
    tp = tq + 0.1 * torch.randn(N, 1, Dv, device="cuda", generator=g)
    tn = torch.randn(N, 1, Dv, device="cuda", generator=g) * 0.4
    ty = (torch.rand(N, 1, device="cuda", generator=g) < 0.8).float()
    def run(sl, lw):
        m = sl.stop - sl.start
        out = dict(s_pos=torch.empty(m, 1, device="cuda"), s_neg=torch.empty(m, 1, device="cuda"),
                   loss=torch.empty(1, device="cuda"), dq=torch.empty(m, 1, Dv, device="cuda"),
                   da_pos=torch.empty(m, 1, Dv, device="cuda"), da_neg=torch.empty(m, 1, Dv, device="cuda"))
        capi.triplet_euclid_step(tq[sl].contiguous(), tp[sl].contiguous(), tn[sl].contiguous(), ty[sl].contiguous(),
                                 margin=0.05, loss_weight=lw, **out)
        return out
    full = run(slice(0, N), 1.0)
    lo, hi = sharded.shard_range(N, rank, world)
    mine = run(slice(lo, hi), sharded.shard_loss_weight(1.0, hi - lo, N))
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
    ok_t = rel(mine["dq"], full["dq"][lo:hi]) < 2e-6 and rel(mine["da_pos"], full["da_pos"][lo:hi]) < 2e-6
    unscaled = run(slice(lo, hi), 1.0)                # the hazard: world_size times too large
    ok_t = ok_t and rel(unscaled["dq"], full["dq"][lo:hi]) > 0.5
    tot = sharded.all_reduce_shard_losses(mine["loss"].cpu(), hi - lo, N)
    ok_t = ok_t and abs(float(tot) - float(full["loss"].cpu())) < 1e-5
    if rank == 0:
        print("sharded triplet step with shard_loss_weight == unsharded: %s" % ok_t)
    flag = torch.tensor([1 if (ok and ok_t) else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "bilinear_and_loss":
        return bilinear_and_loss()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)                      # both ranks share the one GPU of the test box
    from mms_answer_selection_amd import capi, sharded
    n, groups, D = 1517, 68, 300                  # not divisible by the world size: ragged shards
    r = np.random.default_rng(4)                  # the same data on every rank
    group = np.sort(r.integers(0, groups, n)).astype(np.float32)
    label = (r.uniform(size=n) < 0.2).astype(np.float32)
    qvec = (r.standard_normal((groups, D)) * 0.4).astype(np.float32)
    q = qvec[group.astype(int)].reshape(n, 1, D)
    a = (q + r.standard_normal((n, 1, D)).astype(np.float32) * np.where(label, 0.2, 0.4).reshape(n, 1, 1)
         ).astype(np.float32)
    lo, hi = sharded.shard_range(n, rank, world)
    qd, ad = torch.from_numpy(q[lo:hi]).cuda(), torch.from_numpy(a[lo:hi]).cuda()
    top = torch.empty(hi - lo, 1, 1, 1, device="cuda")
    capi.simcross_forward(1, qd, ad, top)         # this rank's pairs only
    full = sharded.all_gather_scores(top.view(hi - lo).cpu(), n)       # gloo moves host tensors
    s = full.cuda()
    prob = torch.stack([1 - s, s], 1).contiguous()
    m, rr, eff = capi.rank_map_mrr(prob, torch.from_numpy(label).cuda(), torch.from_numpy(group).cuda())
    ok = True
    if rank == 0:
        from oracle import cpu_oracle as O
        top_ref, _, _ = O.simcross_forward(1, q, a)
        sr = top_ref.reshape(n)
        prob_ref = np.stack([1 - sr, sr], 1).astype(np.float32)
        m_ref, eff_ref = O.map_score(prob_ref, label, group)
        rr_ref, _ = O.mrr_score(prob_ref, label, group)
        same = lambda x, y: np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32)
        ok = bool((full.numpy().view(np.uint32) == sr.view(np.uint32)).all() and eff == eff_ref
                  and same(m, m_ref) and same(rr, rr_ref))
        print("sharded scoring: scores bit-identical %s, MAP %.6f (ref %.6f), MRR %.6f (ref %.6f)"
              % ((full.numpy().view(np.uint32) == sr.view(np.uint32)).all(), m, m_ref, rr, rr_ref))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
