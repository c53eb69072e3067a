"""tools/f64_probe.py -- dev-only: timings of the double-precision entry points (csrc/f64_paths.hip) at the
BASELINE shapes, eager launches timed with events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: (torch.randn(*s, device="cuda", generator=g, dtype=torch.float64) * 0.4)
def timeit(name, fn, reps=20):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-58s %10.2f us" % (name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)
for mode in (1, 0):
    N, D = 4096, 300
    q, a = rnd(N, 1, D), rnd(N, 1, D)
    dT = rnd(N, 1, 1, 1)
    top = torch.empty(N, 1, 1, 1, device="cuda", dtype=torch.float64)
    n0 = torch.empty(N, 1, device="cuda", dtype=torch.float64); n1 = torch.empty_like(n0)
    dq, da = torch.empty_like(q), torch.empty_like(a)
    kw = dict(norm0=n0, norm1=n1) if mode == 0 else {}
    timeit("simcross f64 mode %d fwd  4096x1x1x300" % mode, lambda: capi.simcross_forward_f64(mode, q, a, top, **kw))
    timeit("simcross f64 mode %d bwd  4096x1x1x300" % mode, lambda: capi.simcross_backward_f64(mode, q, a, top, dT, dq, da, **kw))
N, W, D = 50, 40, 50
q, a = rnd(N, W, D), rnd(N, W, D)
top = torch.empty(N, 1, W, W, device="cuda", dtype=torch.float64); dT = rnd(N, 1, W, W)
dq, da = torch.empty_like(q), torch.empty_like(a)
timeit("simcross f64 mode 1 fwd  50x40x40x50", lambda: capi.simcross_forward_f64(1, q, a, top))
timeit("simcross f64 mode 1 bwd  50x40x40x50", lambda: capi.simcross_backward_f64(1, q, a, top, dT, dq, da))
M = 4
Wm = rnd(M, D, D); bias = rnd(M, W, W)
top = torch.empty(N, M, W, W, device="cuda", dtype=torch.float64); dT = rnd(N, M, W, W)
dW, db = torch.empty_like(Wm), torch.zeros_like(bias)
timeit("simcross f64 mode 2 fwd  50x40x40x50 M=4", lambda: capi.simcross_forward_f64(2, q, a, top, W=Wm, bias=bias))
timeit("simcross f64 mode 2 bwd  50x40x40x50 M=4", lambda: capi.simcross_backward_f64(2, q, a, top, dT, dq, da, W=Wm, bias_term=True, dW=dW, dbias=db))
N, K = 16384, 300
q, a, Wk = rnd(N, K), rnd(N, K), rnd(K, K)
top = torch.empty(N, 1, device="cuda", dtype=torch.float64); qw = torch.empty_like(q)
dT = rnd(N, 1); dq, da, dW = torch.empty_like(q), torch.empty_like(a), torch.zeros_like(Wk)
timeit("simmatrix f64 fwd 16384x300x300", lambda: capi.simmatrix_forward_f64(q, a, Wk, top, qw), reps=5)
timeit("simmatrix f64 bwd 16384x300x300", lambda: capi.simmatrix_backward_f64(q, a, Wk, dT, dq, da, dW), reps=5)
