"""Achievable HBM write / read / copy bandwidth with plain torch kernels (context for the stash-bound kernels)."""
import torch
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for mb in (512, 1400, 4096):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    w = t(lambda: x.fill_(1.0)); r = t(lambda: x.sum()); c = t(lambda: y.copy_(x))
    print(f"{mb} MiB: fill {n*4/w*1e-9:.2f} TB/s  sum(read) {n*4/r*1e-9:.2f} TB/s  copy {2*n*4/c*1e-9:.2f} TB/s (r+w)", flush=True)
