"""Parity of the fused kernels (ray march with early termination, sample compaction, compositing
forward/backward) against the oracle's op-by-op composition of the same reference lines."""
import pytest
import torch

from oracle import kernels as K
from oracle import ref_render as R
from tests import util

pytestmark = pytest.mark.gpu


def _field_and_rays(L, F, log2_T, S, step, n_rays, density_bias, seed):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    field = R.Hash3DAnchored(L, F, log2_T, None, g, feat_init="trained")
    with torch.no_grad():
        field.mlp.bias[0] = density_bias
    o = torch.randn(n_rays, 3, generator=g) * 0.25
    d = torch.randn(n_rays, 3, generator=g)
    noise = torch.rand(n_rays, S, generator=g) - 0.5 + 1.0
    return field, o, d, noise


def _oracle_first_pass(field, o, d, noise, S, step):
    """renderer.cpp:58-90 on the CPU: per-ray kept counts and the compacted samples."""
    with torch.no_grad():
        pts, dirs, dt, t, bounds = R.get_samples(o, d, noise, S, step)
        feat = field.query(pts)
        sec = torch.exp(feat[:, 0] - 3.0) * dt
        acc = K.seg_scan_fwd(sec, bounds, 0)
        mask = torch.exp(-acc) > 1e-4
        num = mask.reshape(-1, S).sum(1).to(torch.int32)
        sel = torch.where(mask)[0]
    return num, pts[sel], dirs[sel], dt[sel], t[sel], mask.reshape(-1, S)


@pytest.mark.parametrize("L,F,log2_T,S,step,bias0,train", [
    (16, 2, 19, 1024, 1.0 / 256, 8.0, True),    # reference-exact sampler, terminating regime
    (16, 2, 19, 1024, 1.0 / 256, 0.0, True),    # dense regime: nothing terminates
    (16, 2, 19, 128, 4.0 / 128, 6.0, True),     # config C2 sampling
    (4, 2, 19, 64, 4.0 / 64, 6.0, False),       # config C1, VALIDATE (no noise)
    (8, 4, 12, 100, 0.04, 7.0, True),           # S not a multiple of 64, F=4
])
def test_density_march_and_compact(capi, dev, L, F, log2_T, S, step, bias0, train):
    n_rays = 150
    field, o, d, noise = _field_and_rays(L, F, log2_T, S, step, n_rays, bias0, seed=L + S)
    if not train:
        noise = None
    num, pts_k, dirs_k, dt_k, t_k, mask = _oracle_first_pass(field, o, d, noise, S, step)
    # prefix property the fused march relies on
    assert torch.equal(mask.to(torch.int32).cumsum(1)[:, -1].to(torch.int32), num)
    assert (mask[:, 1:] <= mask[:, :-1]).all()

    table16 = K.cast_f16(field.feat_pool.detach().reshape(-1)).to(dev)
    w0 = field.mlp.weight.detach()[0].contiguous().to(dev)
    b0 = field.mlp.bias.detach()[0:1].contiguous().to(dev)
    d_o, d_d = o.to(dev), d.to(dev)
    d_noise = noise.to(dev) if train else None
    kept = torch.zeros(n_rays, dtype=torch.int32, device=dev)
    capi.call("density_march", d_o, d_d, d_noise, table16, field.prim_pool.to(dev),
              field.bias_pool.detach().to(dev), field.mul.to(dev), w0, b0, kept, n_rays, S, step,
              L, F, field.T, field.level_stride, 1e-4, 3.0)
    got = kept.cpu()
    # the one-ray-per-wavefront march (strides of 64), the four-rays-per-wavefront one (strides of
    # 16) and the default eight-rays-per-wavefront one (strides of 8) perform the same additions in
    # the same order: identical counts, always
    for route in (1, 2):
        kept_r = torch.zeros(n_rays, dtype=torch.int32, device=dev)
        with capi.option("MARCH", route):
            capi.call("density_march", d_o, d_d, d_noise, table16, field.prim_pool.to(dev),
                      field.bias_pool.detach().to(dev), field.mul.to(dev), w0, b0, kept_r, n_rays, S,
                      step, L, F, field.T, field.level_stride, 1e-4, 3.0)
        assert torch.equal(kept_r.cpu(), got), route
    # T is compared against a threshold: a ray whose T sits within rounding of 1e-4 may keep one
    # sample more or less.  Everything else must agree exactly.
    diff = (got - num).abs()
    assert diff.max().item() <= 1, diff.max()
    assert (diff != 0).float().mean().item() <= 0.02
    if bias0 >= 6.0:
        assert (num < S).any()  # the terminating regime does terminate

    # bounds + compaction driven by the ORACLE's counts so the ragged arrays line up element-wise
    bounds = torch.zeros(n_rays, 2, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    capi.call("bounds_from_counts", num.to(dev), bounds, total, n_rays)
    cum = torch.cumsum(num, 0).to(torch.int32)
    assert torch.equal(bounds.cpu(), torch.stack([cum - num, cum], 1))
    n_kept = int(total.item())
    assert n_kept == int(num.sum())
    c_pts, c_dirs = torch.empty(n_kept, 3, device=dev), torch.empty(n_kept, 3, device=dev)
    c_dt, c_t = torch.empty(n_kept, device=dev), torch.empty(n_kept, device=dev)
    capi.call("sample_compact", d_o, d_d, d_noise, bounds, c_pts, c_dirs, c_dt, c_t, n_rays, S,
              step)
    torch.testing.assert_close(c_pts.cpu(), pts_k, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(c_dirs.cpu(), dirs_k, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(c_t.cpu(), t_k, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(c_dt.cpu(), dt_k, rtol=1e-3, atol=2e-6)


def test_bounds_from_counts_large(capi, dev):
    g = torch.Generator().manual_seed(0)
    # (1 .. 1024 and above 2^17 rays: one workgroup; between: one workgroup per 1024 rays)
    for n in (1, 63, 1024, 1025, 4096, 4097, 65536, 70001, 131072, 131073):
        cnt = torch.randint(0, 1025, (n,), generator=g).to(torch.int32)
        bounds = torch.zeros(n, 2, dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        capi.call("bounds_from_counts", cnt.to(dev), bounds, total, n)
        cum = torch.cumsum(cnt, 0).to(torch.int32)
        assert torch.equal(bounds.cpu(), torch.stack([cum - cnt, cum], 1))
        assert int(total.item()) == int(cum[-1])


def _composite_oracle(logit, rgb, dt, t, idx, bg):
    """renderer.cpp:93,107-118 with the oracle's autograd Functions."""
    density = R.TruncExp.apply(logit - 3.0)
    sec = density * dt
    alphas = 1.0 - torch.exp(-sec)
    acc = R.flex_accumulate_sum(sec, idx, False)
    trans = torch.exp(-acc)
    weights = trans * alphas
    last_trans = torch.exp(-R.flex_sum(sec, idx))
    colors = R.flex_sum(weights.unsqueeze(-1) * rgb, idx) + last_trans.unsqueeze(-1) * bg
    depths = R.flex_sum(weights * (t + 1e-2), idx) / (1.0 - last_trans + 1e-4)
    return colors, depths, weights, last_trans


@pytest.mark.parametrize("n_rays,max_len,with_dw", [(200, 150, True), (64, 1024, False),
                                                    (33, 30, True)])
def test_composite_fwd_bwd(capi, dev, n_rays, max_len, with_dw):
    idx, n = util.ragged_bounds(n_rays, max_len, seed=n_rays)
    g = torch.Generator().manual_seed(n_rays)
    feat = torch.randn(n, 16, generator=g) * 1.5 + 1.0   # column 0 = density logit (ld 16)
    feat[::7, 0] = 9.5                                    # exercise the clamp of TruncExp::backward
    logit = feat[:, 0].clone().requires_grad_(True)
    rgb = torch.rand(n, 3, generator=g).requires_grad_(True)
    dt = torch.rand(n, generator=g) * 0.01
    t = torch.rand(n, generator=g) * 4.0
    bg = torch.rand(n_rays, 3, generator=g)
    colors, depths, weights, last_trans = _composite_oracle(logit, rgb, dt, t, idx, bg)
    d_c = torch.randn(n_rays, 3, generator=g)
    d_d = torch.randn(n_rays, generator=g) * 0.1
    d_w = torch.randn(n, generator=g) * 0.1 if with_dw else torch.zeros(n)
    (colors * d_c).sum().add((depths * d_d).sum()).add((weights * d_w).sum()).backward()

    dv = lambda x: x.detach().to(dev).contiguous()
    o_c, o_d = torch.empty(n_rays, 3, device=dev), torch.empty(n_rays, device=dev)
    o_w, o_lt = torch.zeros(n, device=dev), torch.empty(n_rays, device=dev)
    d_feat = dv(feat)
    capi.call("composite_fwd", d_feat, 16, dv(rgb), dv(dt), dv(t), dv(idx), dv(bg), o_c, o_d, o_w,
              o_lt, n_rays, 3.0, 1e-2)
    torch.testing.assert_close(o_w.cpu(), weights.detach(), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(o_lt.cpu(), last_trans.detach(), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(o_c.cpu(), colors.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(o_d.cpu(), depths.detach(), rtol=1e-4, atol=1e-5)

    g_logit, g_rgb = torch.zeros(n, device=dev), torch.zeros(n, 3, device=dev)
    capi.call("composite_bwd", d_feat, 16, dv(rgb), dv(dt), dv(t), dv(idx), dv(bg), o_w, o_lt,
              dv(d_c), dv(d_d), dv(d_w) if with_dw else None, g_logit, g_rgb, n_rays, 3.0, 1e-2)
    # d_rgb = w * dC and alpha = 1 - exp(-s) carries an absolute error of ~ulp(1) = 6e-8 on both sides
    torch.testing.assert_close(g_rgb.cpu(), rgb.grad, rtol=1e-4, atol=1e-6)
    ref = logit.grad
    torch.testing.assert_close(g_logit.cpu(), ref, rtol=1e-3, atol=1e-4 * ref.abs().max().item())


@pytest.fixture(scope="module")
def host():
    import importlib
    return importlib.import_module("f2-nerf_amd").load_host()


@pytest.mark.parametrize("n_rays,weight", [(1, 0.0), (777, 1e-2), (70001, 0.3)])
def test_train_loss_matches_aten_formula(host, dev, n_rays, weight):
    """f2n::train_loss (two launches) against the reference's ATen spelling of the loss
    (train_manager.cpp:78-96) with torch autograd for the gradients."""
    g = torch.Generator().manual_seed(n_rays)
    colors = torch.rand(n_rays, 3, generator=g).to(dev).requires_grad_(True)
    gt = torch.rand(n_rays, 3, generator=g).to(dev)
    var = (torch.rand(n_rays, generator=g) * 2).to(dev).requires_grad_(True)
    stats = host.train_loss(colors, gt, var, weight)
    c2 = colors.detach().clone().requires_grad_(True)
    v2 = var.detach().clone().requires_grad_(True)
    err = c2 - gt
    color_loss = torch.sqrt(err.square() + 1e-4).mean()
    var_loss = (v2 + 1e-2).sqrt().mean()
    ref = color_loss + var_loss * weight
    torch.testing.assert_close(stats[0], ref, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(stats[1], color_loss.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(stats[2], var_loss.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(stats[3], err.detach().square().sum(), rtol=1e-5, atol=1e-7)
    (stats[0] * 3.0).backward()          # an upstream factor must reach both gradients
    (ref * 3.0).backward()
    torch.testing.assert_close(colors.grad, c2.grad, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(var.grad, v2.grad, rtol=1e-5, atol=1e-9)
    again = host.train_loss(colors.detach(), gt, var.detach(), weight)
    assert torch.equal(again, stats.detach())   # deterministic sums


def test_density_margin_flag(capi, dev):
    """f2n_density_margin: flag set iff some ray's optical depth reaches the limit (or is NaN); the
    exact early-stop scan keeps every sample whenever the flag is clear (that is what the Renderer
    relies on when it accepts the all-samples shading pass without running the scan)."""
    g = torch.Generator().manual_seed(11)
    n_rays, S, thresh = 300, 192, 1e-4
    limit = -torch.log(torch.tensor(thresh)).item() - 0.5
    dt = (torch.rand(n_rays, S, generator=g) * 0.04).reshape(-1)
    for scale, poison in ((0.0, None), (1.0, None), (3.0, None), (0.0, 77)):
        logit = (torch.randn(n_rays, S, generator=g) * 1.5 + scale).reshape(-1)
        if poison is not None:
            logit[poison * S + 5] = float("nan")
        depth = (torch.exp(logit.double() - 3.0) * dt.double()).reshape(n_rays, S).sum(1)
        want = bool((~(depth < limit)).any())
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        capi.call("density_margin", logit.to(dev), dt.to(dev), flag, n_rays, S, 3.0, float(limit))
        assert bool(flag.item()) == want, (scale, poison)
        if not want:
            # sufficiency: exclusive prefix sums never reach -ln(thresh) either
            sec = torch.exp(logit - 3.0) * dt
            bounds = torch.arange(0, n_rays + 1, dtype=torch.int32) * S
            idx = torch.stack([bounds[:-1], bounds[1:]], 1).contiguous()
            acc = K.seg_scan_fwd(sec, idx, 0)
            assert bool((torch.exp(-acc) > thresh).all())
