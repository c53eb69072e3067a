"""tools/layer_host_probe.py -- dev-only: wall-clock per Forward+Backward through the Caffe Layer mirror
(libmms_caffe.so via ctypes) at cfg 2's shape, against the same two C-ABI calls made directly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mms_answer_selection_amd import capi, layers as L
N, D = 4096, 300
r = np.random.default_rng(1)
q = L.Blob((N, 1, D)); a = L.Blob((N, 1, D)); top = L.Blob()
q.data[...] = r.standard_normal((N, 1, D)).astype(np.float32) * 0.4
a.data[...] = r.standard_normal((N, 1, D)).astype(np.float32) * 0.4
lay = L.SimCross(dist_mode=1)
lay.SetUp([q, a], [top])
lay.Forward([q, a], [top])
top.diff[...] = 1.0
lay.Backward([top], [True, True], [q, a])
torch.cuda.synchronize()
K = 3000
t0 = time.perf_counter()
for _ in range(K):
    lay.Forward([q, a], [top])
    lay.Backward([top], [True, True], [q, a])
torch.cuda.synchronize()
t1 = time.perf_counter()
print("Layer mirror: %.2f us per Forward+Backward (host-paced, eager)" % ((t1 - t0) / K * 1e6))
qd = torch.randn(N, 1, D, device="cuda") * 0.4; ad = torch.randn(N, 1, D, device="cuda") * 0.4
dT = torch.randn(N, 1, 1, 1, device="cuda"); tp = torch.empty(N, 1, 1, 1, device="cuda")
dq = torch.empty_like(qd); da = torch.empty_like(ad)
for _ in range(10):
    capi.simcross_forward(1, qd, ad, tp); capi.simcross_backward(1, qd, ad, tp, dT, dq, da)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    capi.simcross_forward(1, qd, ad, tp); capi.simcross_backward(1, qd, ad, tp, dT, dq, da)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("C ABI via ctypes + torch pointers: %.2f us per forward+backward (host-paced, eager)" % ((t1 - t0) / K * 1e6))
