"""The reference's OWN Renderer on the MI355X: src/renderer.cpp, src/hash_3d_anchored.cpp,
src/sh_shader.cpp, src/points_sampler.cpp, src/rays.cpp, src/CustomOps/*.cpp compiled unmodified from
the reference checkout (oracle/build_ref.py) and linked with oracle/ref_cuda_side.cpp, which supplies
the symbols of the five .cu files through the C ABI of libf2nerf_hip.so.

What this pins:
  * the drop-in boundary, with the reference's real callers: its Renderer runs on this library;
  * the oracle's restatement of everything the reference does in ATen -- contraction, two-pass early
    stop with where/index compaction, cat / MLP arrangement, TruncExp, compositing expression, the
    loss lines -- against real reference code (oracle/ref_render.py vs. the reference's Renderer);
  * this repository's Renderer (fused and op-by-op) against the reference's.
It does NOT pin the kernels: rows A1/A2/A6/A7/A9/A10 are this repository's code on both sides.

The reference runs in a process of its own (oracle/ref_host_runner.py): it registers the same
TORCH_LIBRARY namespace as this repository's host library."""
import importlib
import math
import os
import subprocess
import sys

import pytest
import torch

from oracle import build_ref
from oracle import ref_render as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("f2-nerf_amd").load_host()


def _run_reference(tmp_path, payload):
    if not os.path.exists(build_ref.OUT_HOST):
        pytest.fail("oracle/_ref/_f2nerf_ref_host.so missing: run __graft_entry__.build() where "
                    "/root/reference exists; the built file travels with the snapshot")
    pin, pout = str(tmp_path / "in.pt"), str(tmp_path / "out.pt")
    torch.save(payload, pin)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_host_runner.py"), pin, pout],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return torch.load(pout, weights_only=True)


def _close(a, b, rtol, atol_frac=1e-5):
    scale = float(b.abs().max()) if b.numel() else 1.0
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol_frac * scale + 1e-12)


def _close_samples(a, b):
    """Per-sample weights of two parties whose sample POSITIONS differ.  The reference forms d^ and t
    with linalg_norm and a cumsum of 1024 f32 steps on its device (a parallel scan; the oracle runs
    the same ops on the CPU, the HIP sampler adds along a wave), so positions differ by a few
    1e-7..1e-6 (bounded where this is used), and because dt is the norm of the DIFFERENCE of
    neighbouring positions (quirk Q7) its rounding is re-drawn at the 1e-4..1e-3 level -- more in the
    tail of a ray.  That the sampler's op placement is the whole cause is shown in the test below:
    with the oracle's sampler ops run where the reference runs them, per-sample weights agree to
    1e-4 + one ulp of alpha = 1 - exp(-sigma dt) (2^-24, the resolution of the reference's own
    formula).  Integrated quantities (colours, depths, loss) are held to 1e-4 either way; individual
    weights at differing positions to 1e-4 in the median and 2e-3 in relative L2."""
    rel = (a - b).abs() / b.abs().clamp_min(1e-12)
    assert float(rel.median()) < 1e-4, float(rel.median())
    assert float((a - b).norm() / b.norm()) < 2e-3


def _params_of(oracle):
    return {
        "scene_field.feat_pool": oracle.scene_field.feat_pool.detach().clone(),
        "scene_field.prim_pool": oracle.scene_field.prim_pool.detach().clone(),
        "scene_field.bias_pool": oracle.scene_field.bias_pool.detach().clone(),
        "scene_field.mlp.weight": oracle.scene_field.mlp.weight.detach().clone(),
        "scene_field.mlp.bias": oracle.scene_field.mlp.bias.detach().clone(),
        "shader.mlp.0.weight": oracle.shader.mlp[0].weight.detach().clone(),
        "shader.mlp.0.bias": oracle.shader.mlp[0].bias.detach().clone(),
        "shader.mlp.2.weight": oracle.shader.mlp[2].weight.detach().clone(),
        "shader.mlp.2.bias": oracle.shader.mlp[2].bias.detach().clone(),
        "app_emb": oracle.app_emb.detach().clone(),
    }


def _oracle_grads(oracle):
    return {
        "scene_field.feat_pool": oracle.scene_field.feat_pool.grad,
        "scene_field.mlp.weight": oracle.scene_field.mlp.weight.grad,
        "scene_field.mlp.bias": oracle.scene_field.mlp.bias.grad,
        "shader.mlp.0.weight": oracle.shader.mlp[0].weight.grad,
        "shader.mlp.0.bias": oracle.shader.mlp[0].bias.grad,
        "shader.mlp.2.weight": oracle.shader.mlp[2].weight.grad,
        "shader.mlp.2.bias": oracle.shader.mlp[2].bias.grad,
        "app_emb": oracle.app_emb.grad,
    }


@pytest.mark.parametrize("bias0", [7.0, 0.0], ids=["terminating", "dense"])
def test_reference_renderer_train_matches_oracle_and_this_renderer(host, dev, tmp_path, bias0):
    """TRAIN render + loss + backward at the reference's compile-time configuration (L=16, F=2,
    T=2^19, 1024 samples of 1/256): reference Renderer == CPU oracle == this Renderer."""
    E, n_rays, S, seed, vw = 5, 40, 1024, 4242, 1e-2
    g = torch.Generator().manual_seed(17)
    torch.manual_seed(17)
    oracle = R.Renderer(E, L=16, F=2, log2_T=19, S=S, step=1.0 / 256, gen=g, feat_init="trained")
    with torch.no_grad():
        oracle.scene_field.mlp.bias[0] = bias0
    o = torch.randn(n_rays, 3, generator=g) * 0.25
    d = torch.randn(n_rays, 3, generator=g)
    gt = torch.rand(n_rays, 3, generator=g)
    emb = torch.randint(0, E, (n_rays,), generator=g).to(torch.int32)
    ref = _run_reference(tmp_path, dict(params=_params_of(oracle), rays_o=o, rays_d=d, emb_idx=emb,
                                        gt=gt, seed=seed, train=True, var_weight=vw, image=None))
    # the two torch::rand draws of the reference, reproduced on the same device generator
    torch.manual_seed(seed)
    noise = ((torch.rand(n_rays * S, device=dev) - 0.5) + 1.0).view(n_rays, S)   # points_sampler.cpp:35
    bg = torch.rand(n_rays, 3, device=dev)                                        # renderer.cpp:43

    # --- the CPU oracle against the reference's real host code
    loss, res, mse, psnr = R.train_loss(oracle, o, d, emb, gt, noise.cpu(), bg.cpu(), vw)
    loss.backward()
    assert torch.equal(ref["idx_start_end"], res.idx_start_end)
    _close(ref["colors"], res.colors.detach(), 1e-4)
    _close(ref["depths"], res.depths.detach(), 1e-4)
    _close_samples(ref["weights"], res.weights.detach())
    # ... and with IDENTICAL positions: the oracle's sampler ops (src/points_sampler.cpp:24-48:
    # linalg_norm, cumsum, diff / norm) run on the device the reference runs them on, everything
    # downstream still on the CPU -- per-sample weights at the north star's 1e-4 plus ONE ulp of
    # the reference's own alpha = 1 - exp(-sigma dt) (src/renderer.cpp:110): exp(-x) lies in
    # [0.5, 1), so alpha is a multiple of 2^-24 whoever computes it, and a CPU exp and a device exp
    # that differ in the last bit move it by 2^-24 -- 3e-4 of a dense-regime weight of 2e-4
    # (measured: largest difference exactly 2^-24, on 12 % of the samples)
    oracle.sampler_device = dev
    with torch.no_grad():
        res_same = oracle.render(o, d, emb, R.TRAIN, noise.cpu(), bg.cpu())
    oracle.sampler_device = None
    assert torch.equal(ref["idx_start_end"], res_same.idx_start_end)
    torch.testing.assert_close(ref["weights"], res_same.weights, rtol=1e-4, atol=2.0 ** -24 * 1.01)
    _close(ref["colors"], res_same.colors, 1e-4)
    # (at differing positions the same comparison fails by far more than an ulp of alpha)
    worst = float((ref["weights"] - res.weights.detach()).abs().max())
    if bias0 == 0.0:
        assert worst > 4 * 2.0 ** -24, worst
    # how far apart the positions of the three parties are: t of the reference (device cumsum),
    # of the oracle (sequential) and of the HIP sampler (wave scan)
    t_ref = (torch.cumsum(noise, 1) * (1.0 / 256)).cpu()
    t_seq = torch.cumsum(noise.cpu(), 1) * (1.0 / 256)
    t_hip = host.PtsSampler(S, 1.0 / 256).get_samples(o.to(dev), d.to(dev), "train", noise)[3].cpu().view(n_rays, S)
    for t_other in (t_seq, t_hip):
        rel = ((t_other - t_ref).abs() / t_ref).max().item()
        assert 0.0 < rel <= 2e-6, rel          # they DO differ, by a few f32 ulps of t
    assert abs(ref["loss"] - float(loss)) <= 1e-5 * abs(float(loss))
    assert abs(ref["mse"] - mse) <= 1e-5 * mse
    for k, want in _oracle_grads(oracle).items():
        got = ref["grads"][k]
        assert got is not None, k
        if k.endswith("feat_pool"):
            # f16-quantised contributions of samples whose positions differ by the
            # cumsum order (see _close_samples): a few elements in 16.8 M sit at 1.2e-3 of the largest
            _close(got, want, 2e-3, 3e-3)
            assert ((got - want).norm() / want.norm()) < 1e-3
        else:
            _close(got, want, 1e-3, 1e-3)
            assert ((got - want).norm() / want.norm()) < 5e-4

    # --- this repository's Renderer against the reference's
    hr = host.Renderer(E, n_levels=16, n_channels=2, log2_table=19, max_samples=S, step=1.0 / 256)
    hp = hr.named_parameters()
    with torch.no_grad():
        for k, v in _params_of(oracle).items():
            hp[k].copy_(v.to(dev))
    to = lambda x: x.to(dev)
    for fused in (True, False):
        hr.set_fused(fused)
        hr.set_fused_shade(fused)
        hr.zero_grad()
        colors, depths, weights, idx = hr.render(to(o), to(d), to(emb), "train", noise, bg)
        assert torch.equal(idx.cpu(), ref["idx_start_end"])
        _close(colors.detach().cpu(), ref["colors"], 1e-4)
        _close(depths.detach().cpu(), ref["depths"], 1e-4)
        _close_samples(weights.detach().cpu(), ref["weights"])
        hr.zero_grad()
        h_loss, h_sq, n_val, n_samp = hr.train_step(to(o), to(d), to(emb), to(gt), vw, noise, bg, True)
        assert abs(float(h_loss) - ref["loss"]) <= 1e-5 * abs(ref["loss"])
        assert abs(float(h_sq) / n_val - ref["mse"]) <= 1e-5 * ref["mse"]
        grads = hr.grads()
        for k, want in ref["grads"].items():
            if want is None or k.endswith("prim_pool") or k.endswith("bias_pool"):
                continue
            got = grads[k].cpu()
            if k.endswith("feat_pool"):
                _close(got, want, 2e-3, 3e-3)
                assert ((got - want).norm() / want.norm()) < 1e-3
            else:
                _close(got, want, 1e-3, 1e-3)
                assert ((got - want).norm() / want.norm()) < 5e-4


def test_reference_renderer_validate_and_image(host, dev, tmp_path):
    """VALIDATE render and render_image (pixel grid -> get_rays_from_pose -> chunks) of the
    reference against the oracle and this Renderer."""
    E, n_rays, S = 3, 24, 1024
    g = torch.Generator().manual_seed(23)
    torch.manual_seed(23)
    oracle = R.Renderer(E, L=16, F=2, log2_T=19, S=S, step=1.0 / 256, gen=g, feat_init="trained")
    with torch.no_grad():
        oracle.scene_field.mlp.bias[0] = 6.0
    o = torch.randn(n_rays, 3, generator=g) * 0.25
    d = torch.randn(n_rays, 3, generator=g)
    pose = torch.tensor([[1., 0, 0, 0.1], [0, 1, 0, -0.05], [0, 0, 1, 0.6]])
    K = torch.tensor([[20., 0, 4], [0, 20., 3], [0, 0, 1]])
    h, w = 5, 7
    ref = _run_reference(tmp_path, dict(
        params=_params_of(oracle), rays_o=o, rays_d=d, emb_idx=torch.zeros(0, dtype=torch.int32),
        gt=torch.zeros(0), seed=1, train=False, var_weight=0.0,
        image=dict(pose=pose, intrinsic=K, h=h, w=w, batch=16)))
    with torch.no_grad():
        res = oracle.render(o, d, None, R.VALIDATE)
    assert torch.equal(ref["idx_start_end"], res.idx_start_end)
    _close(ref["colors"], res.colors, 1e-4)
    _close(ref["depths"], res.depths, 1e-4)
    hr = host.Renderer(E, n_levels=16, n_channels=2, log2_table=19, max_samples=S, step=1.0 / 256)
    hp = hr.named_parameters()
    with torch.no_grad():
        for k, v in _params_of(oracle).items():
            hp[k].copy_(v.to(dev))
        colors, depths = hr.render_all_rays(o.to(dev), d.to(dev), 10)
        img, dep = hr.render_image(pose.to(dev), K.to(dev), h, w, 16)
    _close(colors.cpu(), ref["colors"], 1e-4)
    _close(depths.cpu().squeeze(-1), ref["depths"], 1e-4)
    ref_img, ref_dep = ref["image"]
    assert tuple(ref_img.shape) == (h, w, 3)
    _close(img.cpu(), ref_img, 1e-4)
    _close(dep.cpu(), ref_dep, 1e-4)


def test_checkpoint_written_by_the_reference_loads_here_and_back(host, dev, tmp_path):
    """`renderer.pt` as the reference writes it (torch::save of ITS module tree,
    src/main_functions/train_manager.cpp:132-136) loads into this repository's Renderer with
    torch::load, and a checkpoint written here loads into the reference's Renderer
    (src/localizer.cpp:37-39) -- same registered names, shapes and dtypes on disk -- and both render
    the same image."""
    E, n_rays, S = 4, 16, 1024
    g = torch.Generator().manual_seed(31)
    torch.manual_seed(31)
    oracle = R.Renderer(E, L=16, F=2, log2_T=19, S=S, step=1.0 / 256, gen=g, feat_init="trained")
    with torch.no_grad():
        oracle.scene_field.mlp.bias[0] = 6.0
    o = torch.randn(n_rays, 3, generator=g) * 0.25
    d = torch.randn(n_rays, 3, generator=g)
    none_i, none_f = torch.zeros(0, dtype=torch.int32), torch.zeros(0)
    ref_ckpt = str(tmp_path / "renderer_ref.pt")
    ref = _run_reference(tmp_path, dict(params=_params_of(oracle), rays_o=o, rays_d=d, emb_idx=none_i,
                                        gt=none_f, seed=1, train=False, var_weight=0.0, image=None,
                                        save_checkpoint=ref_ckpt))
    assert os.path.getsize(ref_ckpt) > 64 << 20          # the 64 MiB f32 table is in there
    hr = host.Renderer(E, n_levels=16, n_channels=2, log2_table=19, max_samples=S, step=1.0 / 256)
    hr.load(ref_ckpt)
    for k, v in _params_of(oracle).items():
        assert torch.equal(hr.named_parameters()[k].detach().cpu(), v), k
    with torch.no_grad():
        colors, depths = hr.render_all_rays(o.to(dev), d.to(dev), 16)
    _close(colors.cpu(), ref["colors"], 1e-4)
    # and the other direction: modify a parameter here, save, let the reference load and render
    with torch.no_grad():
        hr.named_parameters()["scene_field.mlp.bias"][0] = 5.0
        colors2, _ = hr.render_all_rays(o.to(dev), d.to(dev), 16)
    our_ckpt = str(tmp_path / "renderer_ours.pt")
    hr.save(our_ckpt)
    ref2 = _run_reference(tmp_path, dict(params=_params_of(oracle), rays_o=o, rays_d=d, emb_idx=none_i,
                                         gt=none_f, seed=1, train=False, var_weight=0.0, image=None,
                                         load_checkpoint=our_ckpt))
    _close(colors2.cpu(), ref2["colors"], 1e-4)
    assert float((ref2["colors"] - ref["colors"]).abs().max()) > 1e-4   # the edit really travelled
