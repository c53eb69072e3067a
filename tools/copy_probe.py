"""tools/copy_probe.py -- dev-only: what a plain device copy reaches at the footprints of cfg 5 (one GPU's
shard and the whole job on one GPU), for the roofline discussion of the fp16-storage kernel."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N in (8192, 65536, 262144):
    D = 1024
    src = torch.randn(2, N, D, device="cuda").half()
    dst = torch.empty_like(src)
    us = t(lambda: dst.copy_(src))
    mb = src.numel() * 2 * 2 / 1e6
    print(f"copy of q,a -> dq,da footprint N={N}: {us:8.1f} us, {mb:8.1f} MB moved (r+w) = {mb / us * 1e-3 * 1e3 / 1e3:.2f} TB/s", flush=True)
    a = src[0].float(); 
    us2 = t(lambda: torch.add(src[0], src[1], out=dst[0]))
    mb2 = src[0].numel() * 2 * 3 / 1e6
    print(f"   z = x + y (2 reads, 1 write) N={N}: {us2:8.1f} us = {mb2 / us2:.2f} TB/s", flush=True)
