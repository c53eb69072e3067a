"""tools/cross_scale_probe.py -- dev-only: SimCross Euclid 40x40 forward time against the number of pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
D = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator(device="cuda").manual_seed(1)
for N in (64, 256, 512, 1024, 1517, 2048, 3072, 4096, 8192):
    q = torch.randn(N, 40, D, device="cuda", generator=g) * 0.4
    a = torch.randn(N, 40, D, device="cuda", generator=g) * 0.4
    top = torch.empty(N, 1, 40, 40, device="cuda")
    fn = lambda: capi.simcross_forward(1, q, a, top)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(gph, stream=st):
            for _ in range(20): fn()
    torch.cuda.current_stream().wait_stream(st)
    gph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gph.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"N={N:5d} D={D}  fwd {e0.elapsed_time(e1) * 10:7.2f} us", flush=True)
