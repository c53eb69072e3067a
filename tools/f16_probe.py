"""tools/f16_probe.py -- dev-only: cfg 5's shard (8192 x 1024 fp16 storage) fused fwd+bwd, hipGraph-replayed,
cache-warm and HBM-cold; MMS_F16_CHAIN=lanes selects the per-lane chain for A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 1024
if os.environ.get("MMS_F16_TREE") == "1":
    capi.set_f16_distance_mode("tree")
g = torch.Generator(device="cuda").manual_seed(1)
for ring in (1, 12):
    q = (torch.randn(ring, N, 1, D, device="cuda", generator=g) * 0.4).half()
    a = (torch.randn(ring, N, 1, D, device="cuda", generator=g) * 0.4).half()
    dT = torch.randn(ring, N, 1, 1, 1, device="cuda", generator=g)
    top = torch.empty(ring, N, 1, 1, 1, device="cuda")
    dq, da = torch.empty_like(q), torch.empty_like(a)
    step = lambda i: capi.simcross_euclid_forward_backward_f16(q[i], a[i], dT[i], top[i], dq[i], da[i])
    for i in range(ring): step(i)
    torch.cuda.synchronize()
    cap = torch.cuda.Stream(); cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=cap):
            for i in range(12): step(i % ring)
    torch.cuda.current_stream().wait_stream(cap)
    for _ in range(3): gph.replay()
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): gph.replay()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 48)
    ts.sort()
    print("f16 %d x %d chain=%s %s: median %.2f us  min %.2f" % (N, D, "tree-sum" if os.environ.get("MMS_F16_TREE") == "1" else os.environ.get("MMS_F16_CHAIN", "lane-walk (default)"),
          "warm" if ring == 1 else "cold", ts[len(ts) // 2], ts[0]))
