"""Pins against the REAL reference code: oracle/_ref/_f2nerf_ref.so is built (in the authoring
container, by oracle/build_ref.py) from the reference's own src/points_sampler.cpp, src/rays.cpp and
src/CustomOps/CustomOps.cpp -- the host translation units that are pure ATen -- and runs here on the
MI355X through ROCm LibTorch.  It checks (a) the CPU oracle's restatement of those rows and (b) the
HIP kernels that replace them.  The reference's .cu kernels cannot be built (nvcc), so the hash grid,
SH and segment rows stay pinned only by the oracle's independent checks."""
import importlib

import pytest
import torch

from oracle import build_ref
from oracle import ref_render as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    m = build_ref.load()
    if m is None:
        pytest.skip("oracle/_ref/_f2nerf_ref.so not present (built only where /root/reference exists)")
    return m


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("f2-nerf_amd").load_host()


def _rays(n, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, 3, generator=g) * 0.3).to(dev), (torch.randn(n, 3, generator=g) * 2).to(dev)


def test_reference_sampler_validate(ref, host, dev):
    assert ref.max_sample_per_ray() == host.MAX_SAMPLE_PER_RAY == 1024
    o, d = _rays(61, dev)
    r_pts, r_dirs, r_dt, r_t, r_b = ref.get_samples(o, d, False)
    # (a) oracle restatement of points_sampler.cpp:20-64
    c_pts, c_dirs, c_dt, c_t, c_b = R.get_samples(o.cpu(), d.cpu(), None, 1024, 1.0 / 256)
    assert torch.equal(r_b.cpu(), c_b)
    torch.testing.assert_close(r_pts.cpu().reshape(-1, 3), c_pts, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(r_t.cpu(), c_t, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(r_dirs.cpu(), c_dirs, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(r_dt.cpu(), c_dt, rtol=1e-3, atol=2e-6)
    # (b) the HIP sampler kernel behind PtsSampler::get_samples
    s = host.PtsSampler()
    pts, dirs, dt, t, b = s.get_samples(o, d, "validate")
    assert torch.equal(b, r_b)
    torch.testing.assert_close(pts, r_pts.reshape(-1, 3), rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(t, r_t, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(dirs, r_dirs, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(dt, r_dt, rtol=1e-3, atol=2e-6)
    assert (r_dt.reshape(61, 1024)[:, 0] == 0).all()       # quirk Q7: dt_0 = 0


def test_reference_sampler_train_same_noise(ref, host, dev):
    """TRAIN mode: the reference draws torch::rand({n_all}) - .5 + 1 on the device
    (points_sampler.cpp:35).  Re-seeding the device generator reproduces that tensor, which is then
    handed to the HIP sampler as its explicit noise input."""
    o, d = _rays(40, dev, seed=3)
    torch.manual_seed(77)
    r_pts, r_dirs, r_dt, r_t, r_b = ref.get_samples(o, d, True)
    torch.manual_seed(77)
    noise = (torch.rand(40 * 1024, device=dev) - 0.5) + 1.0
    s = host.PtsSampler()
    pts, dirs, dt, t, b = s.get_samples(o, d, "train", noise.reshape(40, 1024))
    assert torch.equal(b, r_b)
    # cumulative sums: the reference's device cumsum and the wave scan associate differently
    torch.testing.assert_close(t, r_t, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(pts, r_pts.reshape(-1, 3), rtol=1e-5, atol=4e-6)
    torch.testing.assert_close(dt, r_dt, rtol=2e-3, atol=4e-6)
    # and the oracle given the same noise
    c_pts, _, c_dt, c_t, _ = R.get_samples(o.cpu(), d.cpu(), noise.cpu().reshape(40, 1024), 1024, 1 / 256)
    torch.testing.assert_close(r_t.cpu(), c_t, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(r_pts.cpu().reshape(-1, 3), c_pts, rtol=1e-5, atol=4e-6)


def test_reference_trunc_exp_and_rays(ref, host, dev):
    x = torch.linspace(-110, 12, 500, device=dev)
    xr = x.clone().requires_grad_(True)
    xm = x.clone().requires_grad_(True)
    yr = ref.trunc_exp(xr)
    ym = host.trunc_exp(xm)
    yr.sum().backward()
    ym.sum().backward()
    assert torch.equal(yr.detach(), ym.detach()) and torch.equal(xr.grad, xm.grad)
    xo = x.cpu().clone().requires_grad_(True)
    R.TruncExp.apply(xo).sum().backward()
    torch.testing.assert_close(xo.grad, xr.grad.cpu(), rtol=1e-6, atol=0)
    assert float(xr.grad[-1]) == pytest.approx(float(torch.exp(torch.tensor(5.0))), rel=1e-6)  # clamp
    # rays: pixel -> world, one pose for all pixels and one pose per pixel (rays.cpp:7-28)
    g = torch.Generator().manual_seed(1)
    n = 33
    pose = torch.randn(n, 3, 4, generator=g).to(dev)
    K = torch.tensor([[1111.1, 0, 400], [0, 1111.1, 400], [0, 0, 1.0]]).expand(n, 3, 3).contiguous().to(dev)
    ij = torch.randint(0, 800, (n, 2), generator=g).to(dev).to(torch.int32)
    ro, rd = ref.get_rays_from_pose(pose, K, ij)
    mo, md = host.get_rays_from_pose(pose, K, ij)
    torch.testing.assert_close(mo, ro, rtol=0, atol=0)
    torch.testing.assert_close(md, rd, rtol=1e-6, atol=1e-6)
    co, cd = R.get_rays_from_pose(pose.cpu(), K.cpu(), ij.cpu())
    torch.testing.assert_close(co, ro.cpu(), rtol=0, atol=0)
    torch.testing.assert_close(cd, rd.cpu(), rtol=1e-5, atol=1e-5)
