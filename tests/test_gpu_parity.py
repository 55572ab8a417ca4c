"""HIP-vs-oracle parity, through the C ABI, on a real MI355X (pytest -m gpu).

Bit-exact for sample depths / voxel indices / masks; stated fp tolerance elsewhere.
"""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test on a box without a HIP device")
    import project_nerf_amd  # noqa: F401
    from project_nerf_amd import ops as _ops
    _ops._lib.load()
    return _ops


def dev(a):
    if isinstance(a, np.ndarray):
        a = T(a)
    return a.cuda()


def synth_rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True) * 4.0311
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 1.6
    d = tgt - o
    return o, d / d.norm(dim=-1, keepdim=True)


# ------------------------------------------------------------------ a1 / a2
@pytest.mark.parametrize("S", [2, 3, 64, 65, 96, 128, 192, 256])
def test_depths_bit_exact_vs_oracle(ops, S):
    R = 37
    o, d = synth_rays(R, 1)
    z = ops.sample_rays(dev(o), dev(d), 2.0, 6.0, S)
    assert torch.equal(z.cpu(), O.stratified_depths(2.0, 6.0, S, R, False))
    u = torch.rand(R, S, generator=torch.Generator().manual_seed(S))
    z = ops.sample_rays(dev(o), dev(d), 2.0, 6.0, S, u=dev(u))
    assert torch.equal(z.cpu(), O.stratified_depths(2.0, 6.0, S, R, True, u=u))


@pytest.mark.parametrize("S", [64, 128])
def test_depths_bit_exact_vs_reference_golden(ops, S):
    g = golden(f"g2_sampling_S{S}")
    o, d = synth_rays(5, 2)
    z0 = ops.sample_rays(dev(o), dev(d), 2.0, 6.0, S)
    z1 = ops.sample_rays(dev(o), dev(d), 2.0, 6.0, S, u=dev(g["u"]))
    assert np.array_equal(z0.cpu().numpy(), g["z_plain"])
    assert np.array_equal(z1.cpu().numpy(), g["z_jitter"])


def test_ray_points(ops):
    R, S = 129, 64
    o, d = synth_rays(R, 3)
    d = d * 1.7
    u = torch.rand(R, S, generator=torch.Generator().manual_seed(9))
    z, pts, dirs = ops.sample_rays(dev(o), dev(d), 2.0, 6.0, S, u=dev(u), want_points=True)
    zc = O.stratified_depths(2.0, 6.0, S, R, True, u=u)
    p_ref, d_ref = O.ray_points(o, d, zc)
    assert torch.equal(pts.cpu(), p_ref)           # mul then add, no FMA: bit-exact
    np.testing.assert_allclose(dirs.cpu().numpy(), d_ref.numpy(), rtol=2e-7, atol=0)


def test_empty_batches(ops):
    e3 = torch.empty(0, 3).cuda()
    assert ops.sample_rays(e3, e3, 2.0, 6.0, 64).shape == (0, 64)
    assert ops.fourier_encode(e3, 10).shape == (0, 63)
    grid = torch.ones(8, 8, 8, dtype=torch.bool).cuda()
    assert ops.active_mask(e3, grid, 1.5).shape == (0,)


# ------------------------------------------------------------------ a3
@pytest.mark.parametrize("res", [64, 128])
def test_voxel_index_and_mask_bit_exact(ops, res):
    g = golden(f"g3_mask_res{res}")
    mask, idx = ops.active_mask(dev(g["pts"]), dev(g["bits"]), 1.5, want_index=True)
    assert np.array_equal(idx.cpu().numpy(), g["idx"])
    assert np.array_equal(mask.cpu().numpy(), g["mask"])


def test_mask_large_random_vs_oracle(ops):
    gen = torch.Generator().manual_seed(5)
    pts = (torch.rand(200_000, 3, generator=gen) - 0.5) * 3.3
    bits = torch.rand(128, 128, 128, generator=gen) < 0.1
    mask, idx = ops.active_mask(dev(pts), dev(bits), 1.5, want_index=True)
    assert torch.equal(idx.cpu(), O.voxel_index(pts, 1.5, 128))
    assert torch.equal(mask.cpu(), O.active_mask(pts, bits, 1.5))


# ------------------------------------------------------------------ a5
@pytest.mark.parametrize("dim,L", [(1, 10), (2, 15), (3, 4), (3, 10), (1, 6)])
def test_fourier_vs_reference_golden(ops, dim, L):
    g = golden(f"g1_fourier_d{dim}_L{L}")
    y = ops.fourier_encode(dev(g["x"]), L).cpu().numpy()
    assert y.shape == g["y"].shape
    # sinf/cosf of the identically rounded fp32 argument: a few ulp of 1.0
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=4e-7)
    assert np.array_equal(y[:, :dim], g["x"])


# ------------------------------------------------------------------ a9
@pytest.mark.parametrize("S", [64, 128])
@pytest.mark.parametrize("tag", ["none", "vec", "ray"])
def test_composite_fwd_bwd_vs_reference_golden(ops, S, tag):
    g = golden(f"g5_composite_S{S}_{tag}")
    bg = None if g["bg"].size == 0 else dev(g["bg"])
    rgb, sig = dev(g["rgb"]).requires_grad_(True), dev(g["sigma"]).requires_grad_(True)
    c, dep, acc, _ = ops.composite(rgb, sig, dev(g["z"]), dev(g["rays_d"]), bg)
    np.testing.assert_allclose(c.detach().cpu().numpy(), g["out_rgb"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dep.detach().cpu().numpy(), g["out_depth"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(acc.detach().cpu().numpy(), g["out_acc"], rtol=2e-5, atol=2e-6)
    ((c * dev(g["g_rgb_map"])).sum() + (dep * dev(g["g_depth"])).sum() + (acc * dev(g["g_acc"])).sum()).backward()
    np.testing.assert_allclose(rgb.grad.cpu().numpy(), g["d_rgb"], rtol=2e-5, atol=1e-7)
    ds, ref = sig.grad.cpu().numpy(), g["d_sigma"]
    scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-12
    assert np.max(np.abs(ds - ref) / scale) < 5e-5


@pytest.mark.parametrize("S", [2, 17, 64, 100, 129, 256])
def test_composite_ragged_sample_counts(ops, S):
    gen = torch.Generator().manual_seed(S)
    R = 33
    z = torch.sort(torch.rand(R, S, generator=gen) * 4 + 2, dim=-1).values
    sig = torch.rand(R, S, generator=gen) * 3
    rgb = torch.rand(R, S, 3, generator=gen)
    extra = torch.randn(R, S, 3, generator=gen)
    _, d = synth_rays(R, S)
    c, dep, acc, ex, w = ops.composite_fwd(dev(rgb), dev(sig), dev(z), dev(d), dev(torch.ones(3)), dev(extra), True)
    rc, rd, ra, rw = O.composite(rgb, sig, z, d, torch.ones(3), return_weights=True)
    np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dep.cpu().numpy(), rd.numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(w.cpu().numpy(), rw.numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(ex.cpu().numpy(), (rw[..., None] * extra).sum(1).numpy(), rtol=1e-4, atol=1e-5)


def test_composite_far_sample_absorbs_remainder(ops):
    """sigma > 0 everywhere => alpha_last = 1 => acc == 1 (reference quirk)."""
    R, S = 16, 64
    z = O.stratified_depths(2.0, 6.0, S, R, False).contiguous()
    sig = torch.full((R, S), 1e-3)
    rgb = torch.rand(R, S, 3)
    _, d = synth_rays(R, 4)
    _, _, acc, _, _ = ops.composite_fwd(dev(rgb), dev(sig), dev(z), dev(d))
    np.testing.assert_allclose(acc.cpu().numpy(), 1.0, atol=1e-6)


@pytest.mark.parametrize("S,bg_kind", [(64, "vec"), (128, "ray"), (17, "none")])
def test_composite_mse_bwd_equals_separate_kernels_and_torch_mse(ops, S, bg_kind):
    """nerf_composite_mse_bwd == volume_render -> nn.MSELoss -> backward (reference run.py:324-337), i.e. the
    golden-pinned composite kernels chained with torch's MSE; the reported maximum is the largest
    output-layer derivative of the vanilla decoder."""
    gen = torch.Generator().manual_seed(S)
    R = 41
    z = torch.sort(torch.rand(R, S, generator=gen) * 4 + 2, dim=-1).values
    sig = torch.relu(torch.randn(R, S, generator=gen)) * 3          # exact zeros included (relu'd density)
    rgb = torch.rand(R, S, 3, generator=gen)
    target = torch.rand(R, 3, generator=gen)
    _, d = synth_rays(R, S)
    bg = {"vec": torch.ones(3), "ray": torch.rand(R, 3, generator=gen), "none": None}[bg_kind]
    bgd = None if bg is None else dev(bg)
    r1, s1 = dev(rgb).requires_grad_(True), dev(sig).requires_grad_(True)
    c, _, _, _ = ops.composite(r1, s1, dev(z), dev(d), bgd)
    loss_ref = torch.nn.functional.mse_loss(c, dev(target))
    loss_ref.backward()
    scal = torch.zeros(2, device="cuda")
    d_rgb, d_sigma, pred = ops.composite_mse_bwd(dev(rgb), dev(sig), dev(z), dev(d), bgd, dev(target), scal[0:1],
                                                 amax_accum=scal[1:2], want_pred=True)
    np.testing.assert_allclose(pred.cpu().numpy(), c.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(float(scal[0]), float(loss_ref), rtol=1e-5)
    np.testing.assert_allclose(d_rgb.cpu().numpy(), r1.grad.cpu().numpy(), rtol=1e-5, atol=1e-10)
    ref = s1.grad.cpu().numpy()
    assert np.max(np.abs(d_sigma.cpu().numpy() - ref) / (np.abs(ref).max(axis=1, keepdims=True) + 1e-20)) < 2e-5
    g_out = torch.cat([(r1.grad * r1 * (1 - r1)).detach().abs().flatten(), (s1.grad * (s1 > 0)).detach().abs().flatten()])
    np.testing.assert_allclose(float(scal[1]), float(g_out.max()), rtol=1e-4)


# ------------------------------------------------------------------ a6
def bf16_decoder(params, pts, dirs):
    """The oracle decoder with the product's numerics: bf16 weights/activations, fp32 accumulate."""
    q = lambda t: t.to(torch.bfloat16).to(torch.float32)
    x, d = q(O.fourier_encode(pts, 10)), q(O.fourier_encode(dirs, 4))
    W = {k: (q(v) if k.endswith("weight") else v) for k, v in params.items()}
    h = x
    for i in range(8):
        if i == 4:
            h = torch.cat([h, x], -1)
        h = q(torch.relu(torch.nn.functional.linear(h, W[f"pts_layers.{i}.weight"], W[f"pts_layers.{i}.bias"])))
    sigma = torch.relu(torch.nn.functional.linear(h, W["sigma_layer.weight"], W["sigma_layer.bias"]))
    feat = q(torch.nn.functional.linear(h, W["feature_layer.weight"], W["feature_layer.bias"]))
    hv = q(torch.relu(torch.nn.functional.linear(torch.cat([feat, d], -1), W["view_layer.weight"], W["view_layer.bias"])))
    rgb = torch.sigmoid(torch.nn.functional.linear(hv, W["rgb_layer.weight"], W["rgb_layer.bias"]))
    return rgb, sigma


def flat_params(params):
    return torch.cat([params[k].reshape(-1) for k, _ in O.nerf_param_shapes()])


def golden_params():
    g = golden("g4_decoder")
    return {k[2:]: T(v) for k, v in g.items() if k.startswith("w:")}, g


def test_decoder_point_mode_vs_reference_golden(ops):
    params, g = golden_params()
    packed = ops.mlp_pack(dev(flat_params(params)))
    rgb, sigma = ops.mlp_fwd(packed, dev(g["pts"]), dev(g["dirs"]), None)
    # fp32 reference vs bf16-MFMA product: stated tolerance 2e-2 abs on rgb, 3% of max on sigma
    np.testing.assert_allclose(rgb.cpu().numpy(), g["rgb"], atol=2e-2)
    np.testing.assert_allclose(sigma.cpu().numpy(), g["sigma"][:, 0], atol=0.03 * max(1.0, g["sigma"].max()))
    # against the same numerics (bf16 operands, fp32 accumulate) the match is tight: what is left is the fp32
    # summation order (MFMA k-blocks vs the host BLAS, which differs from box to box) flipping the bf16
    # rounding of single hidden activations -- 99% of the outputs within 2e-3, none beyond 6e-3
    rb, sb = bf16_decoder(params, T(g["pts"]), T(g["dirs"]))
    err = np.abs(rgb.cpu().numpy() - rb.numpy())
    assert np.quantile(err, 0.99) < 2e-3 and err.max() < 6e-3, (np.quantile(err, 0.99), err.max())
    np.testing.assert_allclose(sigma.cpu().numpy(), sb[:, 0].numpy(), atol=3e-3 * max(1.0, float(sb.max())))


FAMILIES = ["asm-stream", "asm-stream-fp8", "compiler-scheduled"]


def select_family(name):
    """chain-kernel family and training-image width (library options chain_legacy / stash_fp8, include/nerf_hip.h):
    asm-stream = the default (hand-scheduled streams, bf16 training images), asm-stream-fp8 = the opt-in 8-bit
    images, compiler-scheduled = the hipcc-scheduled kernels (bf16 images)."""
    from project_nerf_amd import _lib
    _lib.set_option("chain_legacy", 1 if name == "compiler-scheduled" else 0)
    _lib.set_option("stash_fp8", 1 if name == "asm-stream-fp8" else 0)


@pytest.fixture
def chain_family(request):
    select_family(request.param)
    yield request.param
    select_family("asm-stream")


@pytest.mark.parametrize("chain_family", ["asm-stream", "compiler-scheduled"], indirect=True)
@pytest.mark.parametrize("R,S", [(1, 64), (7, 64), (96, 64), (33, 128), (300, 2), (1101, 64)])
def test_decoder_ray_mode_ragged_tiles(ops, R, S, chain_family):
    """Tiles that are not multiples of 256 samples, scaled weights so that outputs vary; the asm-stream
    kernels (default) and the compiler-scheduled ones.  1101 x 64 = 276 tiles: more tiles than
    CUs, so some workgroups run a second pass (look-ahead DMA and ring hand-over across passes)."""
    params = O.nerf_init_params(seed=R + S)
    params = {k: (v * 2.5 if k.endswith("weight") else v) for k, v in params.items()}
    o, d = synth_rays(R, 7)
    u = torch.rand(R, S, generator=torch.Generator().manual_seed(1))
    z = O.stratified_depths(2.0, 6.0, S, R, True, u=u)
    packed = ops.mlp_pack(dev(flat_params(params)))
    rgb, sigma = ops.mlp_fwd(packed, dev(o), dev(d), dev(z.contiguous()))
    pts, dirs = O.ray_points(o, d, z)
    rb, sb = bf16_decoder(params, pts, dirs)
    assert rgb.shape == (R * S, 3) and sigma.shape == (R * S,)
    # same numerics, different summation order: a bf16 rounding flip can grow through 8 gained-up
    # layers, so the tight bound is on the 99.5th percentile and a looser one on the maximum
    e_rgb = (rgb.cpu() - rb).abs()
    e_sig = (sigma.cpu() - sb[:, 0]).abs() / max(1.0, float(sb.max()))
    # measured over the six shapes and both families (gpurun_out/r04/ragged.txt): max 7.7e-3 / 9.2e-3, 99.5th percentile
    # 2.2e-3 / 1.4e-3; against the fp32 oracle 1.46e-2 -- bounds at 1.5x
    assert float(torch.quantile(e_rgb.flatten(), 0.995)) < 3.3e-3 and float(e_rgb.max()) < 1.2e-2
    assert float(torch.quantile(e_sig, 0.995)) < 2.2e-3 and float(e_sig.max()) < 1.4e-2
    r32, s32 = O.nerf_field(params, pts, dirs)
    e32 = float((rgb.cpu() - r32).abs().max())
    print(f"[ragged tiles {R}x{S} {chain_family}] max |d rgb| vs bf16 oracle {float(e_rgb.max()):.3e} (p99.5 {float(torch.quantile(e_rgb.flatten(), 0.995)):.2e}), "
          f"rel d sigma {float(e_sig.max()):.3e} (p99.5 {float(torch.quantile(e_sig, 0.995)):.2e}), max |d rgb| vs fp32 oracle {e32:.3e}")
    assert e32 < 2.2e-2


def oracle_param_grads(params, pts, dirs, d_rgb, d_sigma):
    ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    rgb, sigma = O.nerf_field(ps, pts, dirs)
    ((rgb * d_rgb).sum() + (sigma[:, 0] * d_sigma).sum()).backward()
    return {k: v.grad for k, v in ps.items()}


class _Q(torch.autograd.Function):
    """bf16 rounding of the value (forward) and of the gradient (backward), straight-through."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(torch.float32) if fwd else x

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(torch.float32) if ctx.bwd else g), None, None


def q8(x, dtype, scale=1.0):
    """x -> 8-bit float (e4m3fn / e5m2, round to nearest even, saturating as MODE.FP16_OVFL makes the
    kernels' conversions) -> fp32, around a power-of-two divisor."""
    lim = 448.0 if dtype is torch.float8_e4m3fn else 57344.0
    return (x / scale).clamp(-lim, lim).to(dtype).to(torch.float32) * scale


def grad_image_scale(amax):
    """mlp_stash.h::grad_image_scale: the power of two that puts amax in [64, 128)."""
    import math
    if amax <= 0:
        return 2.0 ** -126
    return 2.0 ** (math.floor(math.log2(amax)) - 6)


class _Lin8(torch.autograd.Function):
    """Linear layer whose WEIGHT gradient is formed from the 8-bit training images of the asm-stream
    kernels: e4m3 of the (bf16) layer input, e5m2 of the (bf16) pre-activation gradient divided by the
    launch's gradient scale; the input gradient (dgrad chain) keeps the bf16 values."""

    @staticmethod
    def forward(ctx, x, w, b, gscale):
        ctx.save_for_backward(x, w)
        ctx.gscale = gscale          # a one-element list: filled in once the output derivatives are known
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyb = dy.to(torch.bfloat16).to(torch.float32)            # the chain's bf16 pre-activation gradient
        dy8 = q8(dyb, torch.float8_e5m2, ctx.gscale[0])
        x8 = q8(x, torch.float8_e4m3fn)
        return dyb @ w, dy8.t() @ x8, dy8.sum(0), None


def bf16_param_grads(params, pts, dirs, d_rgb, d_sigma, fp8_images=False):
    """Autograd through the oracle decoder with the product's rounding points: bf16 weights,
    bf16 activations forward, bf16 pre-activation gradients backward, fp32 accumulation; with
    ``fp8_images`` the weight gradients additionally see the 8-bit images (asm-stream family)."""
    if fp8_images:
        return fp8_param_grads(params, pts, dirs, d_rgb, d_sigma)
    lin = torch.nn.functional.linear
    ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    W = lambda k: _Q.apply(ps[k], True, False)
    x = _Q.apply(O.fourier_encode(pts, 10), True, False)
    d = _Q.apply(O.fourier_encode(dirs, 4), True, False)
    h = x
    for i in range(8):
        if i == 4:
            h = torch.cat([h, x], -1)
        z = _Q.apply(lin(h, W(f"pts_layers.{i}.weight"), ps[f"pts_layers.{i}.bias"]), False, True)
        h = _Q.apply(torch.relu(z), True, False)
    sigma = torch.relu(_Q.apply(lin(h, W("sigma_layer.weight"), ps["sigma_layer.bias"]), False, True))
    feat = _Q.apply(lin(h, W("feature_layer.weight"), ps["feature_layer.bias"]), True, True)
    zv = _Q.apply(lin(torch.cat([feat, d], -1), W("view_layer.weight"), ps["view_layer.bias"]), False, True)
    hv = _Q.apply(torch.relu(zv), True, False)
    rgb = torch.sigmoid(_Q.apply(lin(hv, W("rgb_layer.weight"), ps["rgb_layer.bias"]), False, True))
    ((rgb * d_rgb).sum() + (sigma[:, 0] * d_sigma).sum()).backward()
    return {k: v.grad for k, v in ps.items()}


def fp8_param_grads(params, pts, dirs, d_rgb, d_sigma):
    """As bf16_param_grads, with every weight gradient contracted from the 8-bit images (_Lin8).  The
    gradient scale comes from the largest output-layer derivative, as bwd_amax_kernel computes it."""
    qb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    W = lambda k: _Q.apply(ps[k], True, False)
    gs = [1.0]
    lin = lambda x, wk, bk: _Lin8.apply(x, W(wk), ps[bk], gs)
    x, d = qb(O.fourier_encode(pts, 10)), qb(O.fourier_encode(dirs, 4))
    ones = torch.ones(x.shape[0], 1)
    h = x
    for i in range(8):
        if i == 4:
            h = torch.cat([h, x], -1)
        h = _Q.apply(torch.relu(lin(h, f"pts_layers.{i}.weight", f"pts_layers.{i}.bias")), True, False)
    sigma = torch.relu(lin(h, "sigma_layer.weight", "sigma_layer.bias"))
    feat = _Q.apply(lin(h, "feature_layer.weight", "feature_layer.bias"), True, False)
    hv = _Q.apply(torch.relu(lin(torch.cat([feat, d], -1), "view_layer.weight", "view_layer.bias")), True, False)
    rgb = torch.sigmoid(lin(hv, "rgb_layer.weight", "rgb_layer.bias"))
    with torch.no_grad():
        g_out = torch.cat([d_rgb * rgb * (1 - rgb), (d_sigma * (sigma[:, 0] > 0))[:, None]], -1)
        gs[0] = grad_image_scale(float(g_out.abs().max()))
    ((rgb * d_rgb).sum() + (sigma[:, 0] * d_sigma).sum()).backward()
    return {k: v.grad for k, v in ps.items()}


def test_decoder_backward_in_two_parts_equals_one_launch(ops):
    """nerf_mlp_bwd_wgrad_part 1 + 2 (the data-parallel, comm-overlapped form) == nerf_mlp_bwd"""
    params = O.nerf_init_params(seed=5)
    R, S = 37, 64
    o, d = synth_rays(R, 9)
    z = O.stratified_depths(2.0, 6.0, S, R, True, u=torch.rand(R, S, generator=torch.Generator().manual_seed(3))).contiguous()
    n = R * S
    gen = torch.Generator().manual_seed(7)
    d_rgb, d_sigma = dev(torch.randn(n, 3, generator=gen)), dev(torch.randn(n, generator=gen))
    packed = ops.mlp_pack(dev(flat_params(params)))
    stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
    rgb, sigma = ops.mlp_fwd(packed, dev(o), dev(d), dev(z), stash)
    whole = ops.mlp_bwd(packed, stash, rgb, sigma, d_rgb, d_sigma)
    parts = torch.full_like(whole, float("nan"))
    ws = torch.empty(ops.mlp_bwd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    seen = []
    ops.mlp_bwd_overlapped(packed, stash, rgb, sigma, d_rgb, d_sigma, parts, ws, lambda v: seen.append(v.numel()))
    assert sum(seen) == whole.numel() and len(seen) == 2
    assert torch.isfinite(parts).all()
    assert float((parts - whole).norm() / whole.norm()) < 1e-5      # float-atomic summation order only


def test_decoder_training_families_agree_over_multiple_passes(ops, monkeypatch):
    """1101 x 64 samples = 276 tiles > 256 CUs: the second pass of a workgroup (stash offsets, mask
    words, look-ahead DMA) in all three families of training kernels.  The two bf16-image families differ only in
    summation order; the 8-bit images add their quantisation."""
    params = O.nerf_init_params(seed=11)
    R, S = 1101, 64
    o, d = synth_rays(R, 3)
    z = O.stratified_depths(2.0, 6.0, S, R, True, u=torch.rand(R, S, generator=torch.Generator().manual_seed(8))).contiguous()
    n = R * S
    gen = torch.Generator().manual_seed(6)
    d_rgb, d_sigma = dev(torch.randn(n, 3, generator=gen)), dev(torch.randn(n, generator=gen))
    packed = ops.mlp_pack(dev(flat_params(params)))
    out = {}
    try:
        for fam in FAMILIES:
            select_family(fam)
            stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
            rgb, sigma = ops.mlp_fwd(packed, dev(o), dev(d), dev(z), stash)
            out[fam] = (rgb.cpu(), sigma.cpu(), ops.mlp_bwd(packed, stash, rgb, sigma, d_rgb, d_sigma).cpu())
    finally:
        select_family("asm-stream")
    a = out["compiler-scheduled"]
    for fam, bound in (("asm-stream", 2e-2), ("asm-stream-fp8", 0.12)):
        b = out[fam]
        assert float((a[0] - b[0]).abs().max()) < 2e-2 and float((a[1] - b[1]).abs().max()) < 2e-2 * max(1.0, float(a[1].max()))
        off = 0
        for name, shape in O.nerf_param_shapes():
            cnt = int(np.prod(shape))
            ga, gb = a[2][off:off + cnt], b[2][off:off + cnt]
            off += cnt
            # bf16 images in both: the same operands, another summation order (and the ReLU flips of the forward's);
            # 8-bit against bf16 training images on random upstream gradients: 0.12
            assert float((ga - gb).norm() / (ga.norm() + 1e-12)) < bound, (fam, name)


# per-tensor bound against the fp32 oracle's gradients, stated from measurement (profiles/r03_measurements.md: bf16
# images rel <= 0.107 / cos >= 0.9943, 8-bit images rel <= 0.130 / cos >= 0.9916 on these inputs)
FP32_GRAD_BOUND = {"asm-stream": (0.13, 0.992), "compiler-scheduled": (0.13, 0.992), "asm-stream-fp8": (0.2, 0.98)}


@pytest.mark.parametrize("chain_family", FAMILIES, indirect=True)
@pytest.mark.parametrize("R,S", [(2, 64), (40, 64), (9, 128)])
def test_decoder_backward_vs_oracle_autograd(ops, R, S, chain_family):
    """dgrad chain + wgrad vs autograd of the oracle, for every family of chain kernels.
    * against the oracle evaluated with the SAME rounding points (bf16 operands, fp32 accumulate; for the
      opt-in 8-bit training images also their e4m3 / e5m2 rounding): per-tensor relative L2 error <= 2e-2;
    * against the pure fp32 oracle: bf16 rounding and the ReLU masks it flips accumulate over the
      10 chained layers -- bound per family in FP32_GRAD_BOUND (bf16 images: relative L2 <= 0.13, cosine >= 0.992;
      the looser 0.2 / 0.98 only for the 8-bit opt-in)."""
    params = O.nerf_init_params(seed=3)
    o, d = synth_rays(R, 17)
    u = torch.rand(R, S, generator=torch.Generator().manual_seed(2))
    z = O.stratified_depths(2.0, 6.0, S, R, True, u=u).contiguous()
    n = R * S
    gen = torch.Generator().manual_seed(5)
    d_rgb = torch.randn(n, 3, generator=gen)
    d_sigma = torch.randn(n, generator=gen)
    flat = dev(flat_params(params))
    packed = ops.mlp_pack(flat)
    stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
    rgb, sigma = ops.mlp_fwd(packed, dev(o), dev(d), dev(z), stash)
    grads = ops.mlp_bwd(packed, stash, rgb, sigma, dev(d_rgb), dev(d_sigma)).cpu()
    pts, dirs = O.ray_points(o, d, z)
    ref32 = oracle_param_grads(params, pts, dirs, d_rgb, d_sigma)
    ref16 = bf16_param_grads(params, pts, dirs, d_rgb, d_sigma, fp8_images=chain_family == "asm-stream-fp8")
    max_rel, min_cos = FP32_GRAD_BOUND[chain_family]
    worst = [0.0, 1.0]
    off = 0
    for name, shape in O.nerf_param_shapes():
        cnt = int(np.prod(shape))
        g = grads[off:off + cnt].reshape(shape)
        off += cnt
        rel16 = float((g - ref16[name]).norm() / (ref16[name].norm() + 1e-12))
        rel32 = float((g - ref32[name]).norm() / (ref32[name].norm() + 1e-12))
        cos32 = float((g * ref32[name]).sum() / (g.norm() * ref32[name].norm() + 1e-20))
        assert rel16 < 2e-2, (name, "bf16-matched", rel16)
        assert rel32 < max_rel and cos32 > min_cos, (name, "fp32", rel32, cos32)
        worst = [max(worst[0], rel32), min(worst[1], cos32)]
    assert off == grads.numel()
    print(f"[grad parity vs fp32 oracle] {chain_family} R={R} S={S}: worst rel {worst[0]:.4f} cos {worst[1]:.5f}")


@pytest.mark.parametrize("chain_family", FAMILIES, indirect=True)
def test_decoder_autograd_function_end_to_end(ops, chain_family):
    """decoder -> composite -> MSE through torch.autograd, against the reference's own gradients (g6), with the
    per-family bound of FP32_GRAD_BOUND (bf16 images: the tight one)."""
    params, _ = golden_params()
    g = golden("g6_render")
    flat = dev(flat_params(params)).requires_grad_(True)
    packed = ops.mlp_pack(flat.detach())
    o, d = dev(g["rays_o"]), dev(g["rays_d"])
    z = ops.sample_rays(o, d, 2.0, 6.0, 64, u=dev(g["u"]))
    rgb, sigma = ops.decoder(flat, packed, o, d, z)
    c, _, _, _ = ops.composite(rgb.view(96, 64, 3), sigma.view(96, 64), z, d, torch.ones(3).cuda())
    np.testing.assert_allclose(c.detach().cpu().numpy(), g["rgb_jitter"], atol=2e-2)
    loss = torch.nn.functional.mse_loss(c, dev(g["target"]))
    assert abs(loss.item() - float(g["loss"])) < 2e-3
    loss.backward()
    grads = flat.grad.cpu()
    off, worst = 0, (0.0, 1.0)
    for name, shape in O.nerf_param_shapes():
        cnt = int(np.prod(shape))
        ref = T(g["dw:" + name])
        g_ = grads[off:off + cnt].reshape(shape)
        rel = float((g_ - ref).norm() / (ref.norm() + 1e-12))
        cos = float((g_ * ref).sum() / (g_.norm() * ref.norm() + 1e-20))
        max_rel, min_cos = FP32_GRAD_BOUND[chain_family]
        assert rel < max_rel and cos > min_cos, (name, rel, cos)     # bf16 chain vs the reference's fp32 autograd
        worst = (max(worst[0], rel), min(worst[1], cos))
        off += cnt
    print(f"[grad parity vs reference g6] {chain_family}: worst rel {worst[0]:.4f} cos {worst[1]:.5f}")


# ------------------------------------------------------------------ a14
def test_adam_and_adamw_vs_reference_golden(ops):
    g = golden("g10_optim")
    for name, lr0, wd in (("adam", 5e-4, 0.0), ("adamw", 1e-2, 1e-5)):
        for k in range(3):
            p = dev(g[f"init_p{k}"]).reshape(-1).clone()
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            for step in range(5):
                lr = lr0 if name == "adam" else O.cosine_lr(1e-2, 1e-4, step, 2000)
                ops.adam_step(p, dev(g[f"grads_p{k}"][step]).reshape(-1), m, v, step + 1, lr, weight_decay=wd)
            np.testing.assert_allclose(p.cpu().numpy(), g[f"{name}_p{k}"].reshape(-1), rtol=3e-6, atol=2e-7)


# ------------------------------------------------------------------ a1-a4 fused
@pytest.mark.parametrize("perturb,R,S", [(False, 77, 64), (True, 77, 64), (True, 3, 5), (True, 33001, 128)])
def test_sample_compact_matches_separate_kernels(ops, perturb, R, S):
    """77 x 64: one ragged pass of a workgroup (4096 samples per pass); 3 x 5: less than a wave; 33001 x 128 = 4.2 M
    samples: more than the 1024 workgroups cover in one pass (grid-stride loop, ragged last pass)."""
    o, d = synth_rays(R, 23)
    gen = torch.Generator().manual_seed(3)
    bits = torch.rand(128, 128, 128, generator=gen) < 0.2
    u = torch.rand(R, S, generator=gen) if perturb else None
    ud = None if u is None else dev(u)
    z_ref, pts, dirs = ops.sample_rays(dev(o), dev(d), 2.0, 6.0, S, u=ud, want_points=True)
    mask = ops.active_mask(pts, dev(bits), 1.5)
    z, slots, pts_c, dirs_c = ops.sample_compact(dev(o), dev(d), 2.0, 6.0, S, dev(bits), 1.5, u=ud)
    assert torch.equal(z, z_ref)                                   # depths: bit-exact
    assert torch.equal(slots >= 0, mask)                           # same voxel test, bit-exact
    n_act = int(mask.sum())
    assert pts_c.shape == (n_act, 3) and n_act > 0
    act = slots[mask].long()
    assert torch.equal(torch.sort(act).values, torch.arange(n_act, device="cuda"))   # a permutation
    assert torch.equal(pts_c[act], pts[mask])                      # every active sample landed in its slot
    np.testing.assert_allclose(dirs_c[act].cpu().numpy(), dirs[mask].cpu().numpy(), rtol=2e-7)


def test_sample_compact_jitter_draws_in_the_kernel(ops):
    """nerf_sample_compact_jitter: same kernel, the uniforms come from the counter-based generator instead of a tensor.
    The depths stay inside their strata, feeding the recovered uniforms back through the tensor entry reproduces them,
    the draws are uniform, repeat for the same (seed, counter) and change with either."""
    R, S = 513, 128
    o, d = synth_rays(R, 5)
    gen = torch.Generator().manual_seed(4)
    bits = dev(torch.rand(128, 128, 128, generator=gen) < 0.2)
    o, d = dev(o), dev(d)
    run = lambda seed, counter: ops.sample_compact_async(o, d, 2.0, 6.0, S, bits, 1.5, jitter=(seed, counter)).get()
    z, slots, pts_c, dirs_c = run(7, 3)
    plain = dev(O.stratified_depths(2.0, 6.0, S, R, False).contiguous())
    mids = 0.5 * (plain[:, 1:] + plain[:, :-1])
    lo, hi = torch.cat([plain[:, :1], mids], -1), torch.cat([mids, plain[:, -1:]], -1)
    assert bool((z >= lo).all()) and bool((z <= hi).all())
    u = ((z - lo) / (hi - lo)).clamp(0, 1)
    assert abs(float(u.mean()) - 0.5) < 5e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    hist = torch.histc(u, bins=16, min=0, max=1) / u.numel()
    assert float((hist - 1 / 16).abs().max()) < 3e-3
    # same voxel test and compaction as the tensor entry on these depths
    z2, slots2, pts2, _ = ops.sample_compact(o, d, 2.0, 6.0, S, bits, 1.5, u=u)
    assert float((z2 - z).abs().max()) < 2e-6
    same = (z2 == z).view(-1)
    assert torch.equal((slots >= 0)[same], (slots2 >= 0)[same]) and float(same.float().mean()) > 0.5
    assert pts_c.shape[0] == int((slots >= 0).sum()) > 0
    z_again = run(7, 3)[0]
    assert torch.equal(z, z_again)
    assert not torch.equal(z, run(7, 4)[0]) and not torch.equal(z, run(8, 3)[0])
    # the asynchronous path is the CHAINED form (nerf_sample_compact_jitter_chain: two counters used alternately, the count published
    # into host-mapped memory by the kernel's last workgroup): against the plain entry point with its own cleared counter and a
    # device-to-host copy, over several consecutive links
    lib = ops._lib.load()
    for counter in (3, 4, 5, 6, 7):
        zc, slots_c, pts_cc, _ = run(7, counter)
        zp = torch.empty_like(zc)
        slots_p = torch.empty_like(slots_c)
        pts_p, dirs_p = torch.empty(R * S, 3, device="cuda"), torch.empty(R * S, 3, device="cuda")
        count = torch.full((1,), 12345, dtype=torch.int32, device="cuda")
        ops._lib.check(lib.nerf_sample_compact_jitter_shard(o.data_ptr(), d.data_ptr(), 7, counter, 0, R, S, 2.0, 6.0, bits.data_ptr(), 128, 1.5,
                                                            zp.data_ptr(), slots_p.data_ptr(), pts_p.data_ptr(), dirs_p.data_ptr(), count.data_ptr(),
                                                            torch.cuda.current_stream().cuda_stream), "nerf_sample_compact_jitter_shard")
        assert torch.equal(zc, zp) and torch.equal(slots_c >= 0, slots_p >= 0)
        assert int(count) == pts_cc.shape[0] == int((slots_c >= 0).sum())
        act = slots_c >= 0
        assert torch.equal(pts_cc[slots_c[act].long()], pts_p[slots_p[act].long()])      # the same sample -> the same point, whatever its slot


def test_composite_indexed_equals_zero_filled_scatter(ops):
    """reference renderer.py:328-343: scatter into zeros then volume_render == compositing through the slot map."""
    R, S = 41, 64
    gen = torch.Generator().manual_seed(8)
    z = dev(O.stratified_depths(2.0, 6.0, S, R, True, u=torch.rand(R, S, generator=gen)).contiguous())
    _, d = synth_rays(R, 9)
    mask = torch.rand(R * S, generator=gen) < 0.3
    mask[:S] = False                                               # one ray with nothing active
    n_act = int(mask.sum())
    perm = torch.randperm(n_act, generator=gen)
    slots = torch.full((R * S,), -1, dtype=torch.int32)
    slots[mask] = perm.int()
    rgb_c = torch.rand(n_act, 3, generator=gen)
    sig_c = torch.rand(n_act, generator=gen) * 5
    bg = torch.tensor([0.3, 0.6, 0.9])
    # dense reference through the oracle
    rgb_d = torch.zeros(R * S, 3); sig_d = torch.zeros(R * S)
    rc, sc = rgb_c.clone().requires_grad_(True), sig_c.clone().requires_grad_(True)
    rgb_d = rgb_d.index_put((torch.nonzero(mask)[:, 0],), rc[perm])
    sig_d = sig_d.index_put((torch.nonzero(mask)[:, 0],), sc[perm])
    c_ref, dep_ref, acc_ref = O.composite(rgb_d.view(R, S, 3), sig_d.view(R, S), z.cpu(), d, bg)
    g1, g2, g3 = torch.rand(R, 3, generator=gen), torch.rand(R, generator=gen) * .1, torch.rand(R, generator=gen) * .1
    ((c_ref * g1).sum() + (dep_ref * g2).sum() + (acc_ref * g3).sum()).backward()
    rg, sg = dev(rgb_c).requires_grad_(True), dev(sig_c).requires_grad_(True)
    c, dep, acc = ops.composite_indexed(rg, sg, dev(slots), z, dev(d), dev(bg))
    np.testing.assert_allclose(c.detach().cpu().numpy(), c_ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(acc.detach().cpu().numpy(), acc_ref.detach().numpy(), rtol=2e-5, atol=2e-6)
    ((c * dev(g1)).sum() + (dep * dev(g2)).sum() + (acc * dev(g3)).sum()).backward()
    np.testing.assert_allclose(rg.grad.cpu().numpy(), rc.grad.numpy(), rtol=3e-5, atol=1e-7)
    ref = sc.grad.numpy()
    assert np.max(np.abs(sg.grad.cpu().numpy() - ref)) < 5e-5 * max(1.0, np.abs(ref).max())


# ------------------------------------------------------------------ hierarchical sampling (extension)
@pytest.mark.parametrize("S,NF,rand", [(64, 128, False), (64, 128, True), (32, 64, True), (128, 128, False)])
def test_sample_pdf_vs_oracle(ops, S, NF, rand):
    R = 50
    gen = torch.Generator().manual_seed(S + NF)
    z = O.stratified_depths(2.0, 6.0, S, R, True, u=torch.rand(R, S, generator=gen)).contiguous()
    w = torch.rand(R, S, generator=gen) ** 4                       # peaky weights
    w[:3] = 0.0                                                    # empty rays: uniform pdf from the 1e-5 floor
    u = torch.rand(R, NF, generator=gen) if rand else None
    out = ops.sample_pdf(dev(z), dev(w), NF, None if u is None else dev(u)).cpu()
    ref = O.sample_pdf(z, w, NF, u)
    assert out.shape == (R, S + NF)
    assert bool((out[:, 1:] >= out[:, :-1]).all())                 # sorted
    assert float(out.min()) >= float(z.min()) - 1e-6 and float(out.max()) <= float(z.max()) + 1e-6
    err = (out - ref).abs()
    # the cdf is a parallel prefix sum here and a sequential cumsum in the oracle: a draw that lands
    # within 1e-7 of a cdf entry may pick the neighbouring bin, so bound the bulk tightly and the
    # rare flips loosely
    assert float(torch.quantile(err.flatten(), 0.999)) < 1e-4, float(torch.quantile(err.flatten(), 0.999))
    assert float(err.max()) < 0.1


def test_inference_with_64_samples_per_wave_equals_the_32_sample_stream_bit_for_bit(ops):
    """Option infer64 (default on: four waves per workgroup, 64 samples per wave, the activations in AGPRs -- every A fragment
    feeds four MFMAs) runs the same MFMAs on the same operands in the same order per sample as the eight-wave stream (infer64 = 0):
    rgb and sigma bit for bit, on whole tiles, a ragged batch and the already-encoded entry."""
    params = O.nerf_init_params(seed=6)
    packed = ops.mlp_pack(dev(flat_params(params)))
    gen = torch.Generator().manual_seed(3)
    for R, S in ((2048, 128), (1000, 64), (3, 2)):
        o, d = synth_rays(R, 7)
        z = dev(O.stratified_depths(2.0, 6.0, S, R, True, u=torch.rand(R, S, generator=gen)).contiguous())
        o, d = dev(o), dev(d)
        outs = []
        for mode in (1, 0):
            ops._lib.set_option("infer64", mode)
            try:
                outs.append(ops.mlp_fwd(packed, o, d, z))
            finally:
                ops._lib.set_option("infer64", 1)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (R, S)
        assert bool(torch.isfinite(outs[0][0]).all()) and float(outs[0][0].std()) > 0


@pytest.mark.parametrize("option", [None, "infer_shape32", "infer64=0"])
def test_chain_kernels_bit_identical_under_full_chip_load(ops, option):
    """Races inside the generated streams show as run-to-run differences once every CU holds several waves
    that compete for the matrix pipe (round 2: a fragment read placed between the two MFMAs of a 16x16x32
    pair overwrote the second one's operand on some boxes).  2048 rays x 128 samples = 1024 wave tiles; the
    forward (inference + training stash) and the dgrad outputs must repeat bit for bit."""
    name, value = (option.split("=") + ["1"])[:2] if option else (None, None)
    if option:
        previous = ops._lib.get_option(name)
        ops._lib.set_option(name, int(value))
    try:
        params = O.nerf_init_params(seed=5)
        R, S = 2048, 128
        o, d = synth_rays(R, 9)
        z = dev(O.stratified_depths(2.0, 6.0, S, R, False).contiguous())
        o, d = dev(o), dev(d)
        packed = ops.mlp_pack(dev(flat_params(params)))
        n = R * S
        gen = torch.Generator().manual_seed(1)
        d_rgb, d_sigma = dev(torch.randn(n, 3, generator=gen)), dev(torch.randn(n, generator=gen))
        ref = None
        for rep in range(6):
            rgb_i, sig_i = ops.mlp_fwd(packed, o, d, z)
            stash = torch.zeros(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")   # zeros: padding compares equal
            rgb_t, sig_t = ops.mlp_fwd(packed, o, d, z, stash)
            work = torch.zeros(ops.mlp_bwd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
            ops.mlp_bwd(packed, stash, rgb_t, sig_t, d_rgb, d_sigma, workspace=work)   # dgrad images live in `work`
            cur = (rgb_i, sig_i, rgb_t, sig_t, stash, work)
            if ref is None:
                ref = tuple(t.clone() for t in cur)
                # inference and training kernels agree on the outputs up to the MFMA shape's summation order
                assert float((rgb_i - rgb_t).abs().max()) < 2e-2
            else:
                for i, (a, b) in enumerate(zip(ref, cur)):
                    assert torch.equal(a, b), (rep, i, int((a != b).sum()), (a != b).nonzero()[:4].flatten().tolist())
    finally:
        if option:
            ops._lib.set_option(name, previous)


def test_wgrad_partial_tiles_equal_the_atomic_flush_and_repeat_bit_for_bit(ops):
    """From 65,536 samples on, the split-K weight-gradient kernel writes one partial tile per (workgroup, layer)
    and a second kernel sums them in a fixed order (option wgrad_atomic = 1: float atomics as before).  Same
    gradients up to fp32 summation order; the slab form repeats bit for bit; the two-range form used by the
    data-parallel step (parts 1 and 2) fills the same vector."""
    params = O.nerf_init_params(seed=21)
    R, S = 2048, 64
    n = R * S
    o, d = synth_rays(R, 4)
    z = dev(O.stratified_depths(2.0, 6.0, S, R, False).contiguous())
    o, d = dev(o), dev(d)
    packed = ops.mlp_pack(dev(flat_params(params)))
    gen = torch.Generator().manual_seed(2)
    d_rgb, d_sigma = dev(torch.randn(n, 3, generator=gen)), dev(torch.randn(n, generator=gen))
    stash = torch.empty(ops.mlp_stash_bytes(n), dtype=torch.uint8, device="cuda")
    rgb, sigma = ops.mlp_fwd(packed, o, d, z, stash)
    out = {}
    for mode in (1, 0):
        ops._lib.set_option("wgrad_atomic", mode)
        try:
            nbytes = ops.mlp_bwd_workspace_bytes(n)
            out[mode] = [ops.mlp_bwd(packed, stash, rgb, sigma, d_rgb, d_sigma).clone() for _ in range(2)]
            out[(mode, "bytes")] = nbytes
        finally:
            ops._lib.set_option("wgrad_atomic", 0)
    assert out[(0, "bytes")] > out[(1, "bytes")] + 64 * 1024 * 1024          # the slab is part of the workspace
    a, s0, s1 = out[1][0], out[0][0], out[0][1]
    assert torch.equal(s0, s1)
    off = 0
    for name, shape in O.nerf_param_shapes():
        cnt = int(np.prod(shape))
        ga, gs = a[off:off + cnt], s0[off:off + cnt]
        off += cnt
        assert float((ga - gs).norm() / (ga.norm() + 1e-20)) < 2e-6, name
    # two-range form
    ws = torch.empty(ops.mlp_bwd_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    grads = torch.full_like(s0, float("nan"))
    ops.mlp_bwd_overlapped(packed, stash, rgb, sigma, d_rgb, d_sigma, grads, ws, lambda view: None)
    assert bool(torch.isfinite(grads).all())                               # every parameter was written
    assert float((grads - s0).norm() / s0.norm()) < 2e-6                   # other spans, hence another summation order
