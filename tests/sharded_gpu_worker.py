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


def main():
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
