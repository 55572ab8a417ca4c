"""Probe: the vanilla training step as TWO half-batches with the weight-gradient pass of the first half (HBM-bound) running
beside the forward + dgrad chains of the second half (MFMA-bound) on disjoint CU sets (library options chain_grid /
wgrad_grid cap the persistent kernels' workgroup counts; the two run on different HIP streams).
    python tools/halfbatch_probe.py            -> sequential step vs the pipelined schedule for several CU splits"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import project_nerf_amd  # noqa: E402,F401
from project_nerf_amd import _lib, ops  # noqa: E402
from project_nerf_amd.engine import default_init  # noqa: E402

R, S = 4096, 64
dev = "cuda"
params = default_init(0).cuda()
packed = ops.mlp_pack(params)
o = torch.randn(R, 3, device=dev)
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
target = torch.rand(R, 3, device=dev)
bg = torch.ones(3, device=dev)
z = ops.sample_rays(o, d, 2.0, 6.0, S, u=torch.rand(R, S, device=dev))
lib = _lib.load()
P = lambda t: t.data_ptr()


class Half:
    def __init__(self, lo, hi):
        self.o, self.d, self.t, self.z = o[lo:hi].contiguous(), d[lo:hi].contiguous(), target[lo:hi].contiguous(), z[lo:hi].contiguous()
        self.R = hi - lo
        self.n = self.R * S
        self.stash = torch.empty(ops.mlp_stash_bytes(self.n), dtype=torch.uint8, device=dev)
        self.ws = torch.empty(ops.mlp_bwd_workspace_bytes(self.n), dtype=torch.uint8, device=dev)
        self.grads = torch.empty(ops.MLP_PARAM_COUNT, device=dev)
        self.scal = torch.zeros(2, device=dev)

    def chains(self):
        rgb, sigma = ops.mlp_fwd(packed, self.o, self.d, self.z, self.stash)
        d_rgb, d_sigma, _ = ops.composite_mse_bwd(rgb.view(self.R, S, 3), sigma.view(self.R, S), self.z, self.d, bg, self.t, self.scal[0:1],
                                                  amax_accum=self.scal[1:2], loss_weight=1.0 / (3 * R))
        _lib.check(lib.nerf_mlp_bwd_dgrad_ex(P(packed), P(self.stash), P(rgb), P(sigma), P(d_rgb), P(d_sigma), self.n, P(self.ws),
                                             P(self.scal[1:2]), torch.cuda.current_stream().cuda_stream), "dgrad")
        self.keep = (rgb, sigma, d_rgb, d_sigma)

    def wgrad(self):
        _lib.check(lib.nerf_mlp_bwd_wgrad(P(self.stash), P(self.ws), self.n, P(self.grads), torch.cuda.current_stream().cuda_stream), "wgrad")


full, a, b = Half(0, R), Half(0, R // 2), Half(R // 2, R)
side = torch.cuda.Stream()


def sequential():
    full.chains()
    full.wgrad()


def pipelined(g_chain, g_wgrad):
    main = torch.cuda.current_stream()
    a.chains()                                   # all CUs
    ev_a = torch.cuda.Event()
    ev_a.record(main)
    _lib.set_option("wgrad_grid", g_wgrad)
    with torch.cuda.stream(side):
        side.wait_event(ev_a)
        a.wgrad()                                # on g_wgrad CUs ...
    _lib.set_option("chain_grid", g_chain)
    b.chains()                                   # ... beside the second half's chains on g_chain CUs
    _lib.set_option("chain_grid", 0)
    _lib.set_option("wgrad_grid", 0)
    main.wait_stream(side)
    b.wgrad()                                    # all CUs
    a.grads.add_(b.grads)


def ms(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


print(f"sequential (fwd + loss + dgrad + wgrad of {R} rays): {ms(sequential):.4f} ms")
print(f"two halves, no overlap (all kernels on all CUs, one stream): {ms(lambda: (a.chains(), a.wgrad(), b.chains(), b.wgrad())):.4f} ms")
sequential()
ref = full.grads.clone()
for g_chain, g_wgrad in ((256, 256), (192, 64), (176, 80), (160, 96), (144, 112), (128, 128), (208, 48)):
    t = ms(lambda: pipelined(g_chain, g_wgrad))
    pipelined(g_chain, g_wgrad)
    torch.cuda.synchronize()
    err = float((a.grads - ref).norm() / ref.norm())
    print(f"pipelined, chains on {g_chain} CUs beside wgrad on {g_wgrad}: {t:.4f} ms   (gradient vs sequential: rel {err:.2e})")
