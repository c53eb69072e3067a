"""dev-only: cfg 2 step with the operands crossing PCIe each step (pinned host memory): H2D of q, a, dT,
the fused launch, D2H of top, dq, da -- the rate a host-buffer boundary would see."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mms_answer_selection_amd import capi
N, D = 4096, 300
hq = torch.randn(N, 1, D).pin_memory(); ha = torch.randn(N, 1, D).pin_memory(); hT = torch.randn(N, 1, 1, 1).pin_memory()
hdq = torch.empty(N, 1, D).pin_memory(); hda = torch.empty(N, 1, D).pin_memory(); htop = torch.empty(N, 1, 1, 1).pin_memory()
q = torch.empty(N, 1, D, device="cuda"); a = torch.empty_like(q); dT = torch.empty(N, 1, 1, 1, device="cuda")
top = torch.empty_like(dT); dq = torch.empty_like(q); da = torch.empty_like(q)
def step():
    q.copy_(hq, non_blocking=True); a.copy_(ha, non_blocking=True); dT.copy_(hT, non_blocking=True)
    capi.simcross_forward_backward(1, q, a, dT, top, dq, da)
    hdq.copy_(dq, non_blocking=True); hda.copy_(da, non_blocking=True); htop.copy_(top, non_blocking=True)
for _ in range(5): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): step()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 10
print("PCIe-inclusive: %.1f us per step, %.3g pairs/s (%.1f GB/s over the link)" % (us, N / (us * 1e-6), 19.7e6 / (us * 1e-6) / 1e9))
