#!/usr/bin/env python3
"""tools/bench_configs.py -- secondary measurements (not the headline bench):
time the C-ABI entry points on the other BASELINE configs / geometries with HIP
events (cache-warm, median of 5 x `iters`).  Prints one JSON object per line."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi

capi.lib()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1701)


def rnd(*s, scale=0.4):
    return torch.randn(*s, device=dev, generator=g) * scale


def timeit(fn, iters=50, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / iters)
    return sorted(out)[len(out) // 2]


def simcross(mode, N, W1, W2, D, M=1, bias=True, iters=50):
    q, a = rnd(N, W1, D), rnd(N, W2, D)
    W = (torch.rand(M, D, D, device=dev, generator=g) * 0.16 - 0.08) if mode == 2 else None
    b = rnd(M, W1, W2, scale=1.0) if (mode == 2 and bias) else None
    top = torch.empty(N, M, W1, W2, device=dev)
    dT = rnd(N, M, W1, W2, scale=1.0)
    n0, n1 = torch.empty(N, W1, device=dev), torch.empty(N, W2, device=dev)
    dq, da = torch.empty_like(q), torch.empty_like(a)
    dW = torch.empty_like(W) if mode == 2 else None
    db = torch.zeros_like(b) if b is not None else None
    f = lambda: capi.simcross_forward(mode, q, a, top, W=W, bias=b, norm0=n0, norm1=n1)
    bwd = lambda: capi.simcross_backward(mode, q, a, top, dT, dq, da, W=W, bias_term=b is not None,
                                         norm0=n0, norm1=n1, dW=dW, dbias=db)
    tf, tb = timeit(f, iters), timeit(bwd, iters)
    s = 4
    bytes_f = s * (N * (W1 + W2) * D + N * M * W1 * W2)
    bytes_b = s * (2 * N * (W1 + W2) * D + 2 * N * M * W1 * W2)
    rec = {"op": "SimCross", "mode": mode, "N": N, "W1": W1, "W2": W2, "D": D, "M": M,
           "fwd_us": tf, "bwd_us": tb, "pairs_per_s_fwd_bwd": N / ((tf + tb) * 1e-6),
           "fwd_GBps_alg": bytes_f / tf / 1e3, "bwd_GBps_alg": bytes_b / tb / 1e3}
    if mode == 2:
        ff = 2.0 * N * M * W1 * D * (D + W2)
        fb = 8.0 * N * M * max(W1, W2) * D * (D + max(W1, W2))
        rec.update({"fwd_TFLOPs": ff / tf / 1e6, "bwd_TFLOPs_est": fb / tb / 1e6})
    print(json.dumps(rec), flush=True)


def simmatrix(N, K1, K2, iters=20):
    q, a = rnd(N, K1), rnd(N, K2)
    W = torch.rand(K1, K2, device=dev, generator=g) * 0.16 - 0.08
    top, scr = torch.empty(N, 1, device=dev), torch.empty(N, K2, device=dev)
    dT = rnd(N, 1, scale=1.0)
    dq, da, dW = torch.empty_like(q), torch.empty_like(a), torch.zeros_like(W)
    tf = timeit(lambda: capi.simmatrix_forward(q, a, W, top, scr), iters)
    tb = timeit(lambda: capi.simmatrix_backward(q, a, W, dT, dq, da, dW, qw=scr), iters)   # the Layer's sequence
    tb_re = timeit(lambda: capi.simmatrix_backward(q, a, W, dT, dq, da, dW), iters)
    ff = 2.0 * N * K1 * K2 + 2.0 * N * K2
    fb = 6.0 * N * K1 * K2
    print(json.dumps({"op": "SimMatrix", "N": N, "K1": K1, "K2": K2, "fwd_us": tf, "bwd_us": tb, "bwd_recomputing_us": tb_re,
                      "fwd_TFLOPs": ff / tf / 1e6, "bwd_TFLOPs": fb / tb / 1e6,
                      "frac_mfma_fwd_bwd": (ff + fb) / ((tf + tb) * 1e-6) / 157.3e12,
                      "pairs_per_s_fwd_bwd": N / ((tf + tb) * 1e-6)}), flush=True)


def simcross_f16(N, D, iters=20):
    """BASELINE cfg 5 per-GPU shard: fp16 storage, fused fwd+bwd, one launch."""
    q = rnd(N, 1, D).half()
    a = rnd(N, 1, D).half()
    dT = rnd(N, 1, 1, 1, scale=1.0)
    top = torch.empty(N, 1, 1, 1, device=dev)
    dq, da = torch.empty_like(q), torch.empty_like(a)
    t = timeit(lambda: capi.simcross_euclid_forward_backward_f16(q, a, dT, top, dq, da), iters)
    b_unfused = 2 * (3 * N * 2 * D) + 4 * 3 * N          # SURVEY 8d cfg 5 formula (s = 2)
    b_fused = 2 * (2 * N * 2 * D) + 4 * 2 * N
    print(json.dumps({"op": "SimCross Euclid fp16 storage fused fwd+bwd", "N": N, "D": D, "us": t,
                      "pairs_per_s": N / (t * 1e-6), "GBps_alg_unfused": b_unfused / t / 1e3,
                      "GBps_fused_bytes": b_fused / t / 1e3,
                      "frac_hbm_unfused_bytes": b_unfused / (t * 1e-6) / 8e12}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "cross", "cfg3", "v4"]
    if "g1" in which:
        simcross(1, 4096, 1, 1, 300)
        simcross(0, 4096, 1, 1, 300)
        simcross(1, 65536, 1, 1, 1024, iters=10)      # cfg 5 shape (fp32 here)
    if "cfg5" in which or "g1" in which:
        simcross_f16(8192, 1024)                      # cfg 5: 65536 pairs over 8 GPUs
        simcross_f16(65536, 1024, iters=10)
    if "cross" in which:
        simcross(1, 50, 40, 40, 50)                    # reference default geometry
        simcross(1, 32, 40, 40, 300)                   # cfg 1 geometry
        simcross(1, 1517, 40, 40, 50, iters=5)         # cfg 4: TREC-QA test split
        simcross(0, 50, 40, 40, 50)
    if "cfg3" in which:
        simmatrix(16384, 300, 300)
        simcross(2, 16384, 1, 1, 300, 1, bias=False, iters=10)
    if "v4" in which:
        simcross(2, 50, 40, 40, 50, 4, iters=10)       # network_v4 (do_trec_qa_clean.py:468)
        simcross(2, 32, 40, 40, 300, 4, iters=5)       # cfg 1
