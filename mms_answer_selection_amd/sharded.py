"""Pair-sharded execution over the GPUs of one node.

Every output of the hot path depends on one pair index n only
(sim_cross_layer.cpp:97-109), so the batch is split contiguously: rank g owns
pairs [g*N/W, (g+1)*N/W).  Forward exchanges only the per-pair SCORES
(all-gather); backward exchanges only the shared-parameter gradients dW/dbias
(all-reduce) -- dq/da never leave their GPU.  One process per GPU,
torch.distributed: backend "nccl" is RCCL over xGMI on ROCm; "gloo" runs the
same code on CPU tensors for the world_size-2 tests.

The reference's only multi-GPU path is P2PSync's tree of whole-parameter-buffer
copies (src/caffe/parallel.cpp:287-381); it never shards this path.  This module
replaces it for the MMS layers; it is not a translation of it.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world):
    """Contiguous partition: rank g owns [bounds[g], bounds[g+1])."""
    return [(g * n) // world for g in range(world + 1)]


def shard_range(n, rank, world):
    b = shard_bounds(n, world)
    return b[rank], b[rank + 1]


def shard(t, rank, world):
    """This rank's rows of a tensor whose axis 0 is the pair index."""
    lo, hi = shard_range(t.shape[0], rank, world)
    return t[lo:hi]


def all_gather_scores(local, n_total, group=None, out=None):
    """All-gather the per-pair scores (axis 0 = pairs of this rank's shard) into
    the full (n_total, ...) tensor on every rank, in pair order.

    Shards may be ragged (n_total % world != 0): each rank pads to the largest
    shard, one all_gather_into_tensor moves everything, the padding is dropped.
    With 4096 pairs per rank the message is 16 KiB -- latency-bound, so callers
    that score many batches should gather several batches per call (bench.py
    buckets GROUP steps per collective)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b = shard_bounds(n_total, world)
    lo, hi = b[rank], b[rank + 1]
    assert local.shape[0] == hi - lo, "rank %d holds %d pairs, expected %d" % (rank, local.shape[0], hi - lo)
    tail = tuple(local.shape[1:])
    mx = max(b[g + 1] - b[g] for g in range(world))
    if (hi - lo) == mx and n_total == mx * world:
        send = local.contiguous()
    else:
        send = local.new_zeros((mx,) + tail)
        send[: hi - lo] = local
    recv = local.new_empty((world * mx,) + tail)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    if n_total == mx * world:
        full = recv
    else:
        full = torch.cat([recv[g * mx: g * mx + (b[g + 1] - b[g])] for g in range(world)], 0)
    if out is not None:
        out.copy_(full)
        return out
    return full


def all_reduce_param_grads(grads, group=None):
    """Sum dW / dbias over ranks (SimCross dist_mode 2, SimMatrix).  Parameters
    are replicated, every rank saw a disjoint slice of the pairs, so the full
    gradient is the sum; dist_mode 0/1 have no parameters and never call this.
    One flattened bucket -> one collective (xGMI links are per-pair: few large
    messages beat many small ones)."""
    grads = [g for g in grads if g is not None]
    if not grads:
        return
    if len(grads) == 1:
        dist.all_reduce(grads[0], op=dist.ReduceOp.SUM, group=group)
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off: off + g.numel()].view_as(g))
        off += g.numel()


def all_reduce_loss(local_sum, n_total, group=None):
    """Mean loss over ALL pairs from per-rank SUMS of the per-pair terms
    (pair_rank_loss_layer.cpp:41-49 divides by the global count)."""
    t = local_sum.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t / float(n_total)


def shard_loss_weight(loss_weight, n_local, n_total):
    """The `top_diff` / `loss_weight` to hand a PairRankLoss backward (mms_pairrank_backward_f32) or the fused
    triplet step (mms_triplet_euclid_step_f32) that runs on ONE RANK'S SHARD of the batch.

    The reference scales every gradient element by top_diff / bottom[0]->count() -- the count of the WHOLE
    batch (pair_rank_loss_layer.cpp:62-64).  The C ABI divides by the count it is given, which on a shard is
    the local count, so an unscaled call would return gradients world_size times too large.  Passing
    loss_weight * n_local / n_total restores the reference's scale: (w * n_local / n_total) / n_local = w / n_total.
    The loss OUTPUT of those calls is the unweighted mean over the shard; all_reduce_shard_losses() turns the
    per-shard means into the batch loss."""
    if n_total <= 0 or n_local < 0 or n_local > n_total:
        raise ValueError("shard of %d pairs out of %d" % (n_local, n_total))
    return float(loss_weight) * float(n_local) / float(n_total)


def all_reduce_shard_losses(local_mean_loss, n_local, n_total, loss_weight=1.0, group=None):
    """The loss of the WHOLE batch from the per-shard losses.  The C ABI's loss outputs (mms_pairrank_forward_f32,
    mms_triplet_euclid_step_f32) are the UNWEIGHTED mean over the elements they were given, so a shard's share of
    the batch mean is n_local / n_total of it; the Layer's loss weight multiplies the result
    (layer.hpp:462-481)."""
    t = local_mean_loss.clone() * (float(loss_weight) * float(n_local) / float(n_total))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
