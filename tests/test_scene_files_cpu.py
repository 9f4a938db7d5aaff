"""On-disk formats either side of the render path, host-only code:
cams_meta.tsv as Dataset::Dataset reads it (reference src/dataset.cpp:27-75), the scene normalisation
(:77-86) and inference_params.yaml byte for byte as Dataset::save_inference_params writes it
(:106-133) / with the fields Localizer reads (src/localizer.cpp:23-36).  The reference holds no
sample of either file; the known answers below are its stream statements carried out by hand
(std::fixed => six decimals of the float32 value)."""
import importlib

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("f2-nerf_amd").load_host()


def test_cams_meta_tsv_and_normalisation(host, tmp_path):
    g = torch.Generator().manual_seed(3)
    n = 5
    poses = torch.randn(n, 3, 4, generator=g)
    intr = torch.tensor([[1111.1, 0.0, 400.0], [0.0, 1111.1, 400.0], [0.0, 0.0, 1.0]]).repeat(n, 1, 1)
    dist = torch.randn(n, 4, generator=g) * 1e-2
    bounds = torch.tensor([[0.1, 7.5]]).repeat(n, 1)
    header = "\t".join(["c%d" % i for i in range(27)])
    rows = [header]
    for i in range(n):
        vals = torch.cat([poses[i].reshape(-1), intr[i].reshape(-1), dist[i], bounds[i]])
        rows.append("\t".join(repr(float(v)) for v in vals))
    path = tmp_path / "cams_meta.tsv"
    path.write_text("\n".join(rows) + "\n")
    p, k, d, b = host.read_cams_meta(str(path))
    assert torch.equal(p, poses) and torch.equal(k, intr) and torch.equal(d, dist) and torch.equal(b, bounds)
    # src/dataset.cpp:77-86
    pn, center, radius = host.normalize_scene(p)
    cam = poses[:, :, 3]
    want_c = cam.mean(0)
    want_r = float((cam - want_c).norm(dim=1).max())
    assert torch.allclose(center, want_c) and abs(radius - want_r) < 1e-6
    assert torch.allclose(pn[:, :, 3], (cam - want_c) / want_r, atol=1e-6)
    assert torch.equal(pn[:, :, :3], poses[:, :, :3])
    assert abs(float(pn[:, :, 3].norm(dim=1).max()) - 1.0) < 1e-5
    with pytest.raises(RuntimeError):
        bad = tmp_path / "bad.tsv"
        bad.write_text(header + "\n1\t2\t3\n")
        host.read_cams_meta(str(bad))


def test_inference_params_yaml_is_the_reference_text(host, tmp_path):
    K = torch.tensor([[1111.1, 0.0, 400.0], [0.0, 1111.1, 300.5], [0.0, 0.0, 1.0]])
    center = torch.tensor([0.25, -1.5, 3.0000001])
    radius = 2.7182817
    host.save_inference_params(str(tmp_path), 50, 600, 800, K, center, radius)
    text = (tmp_path / "inference_params.yaml").read_text()
    f = lambda v: "%.6f" % float(np.float32(v))
    want = ("%YAML 1.2\n---\nn_images: 50\nheight: 600\nwidth: 800\n"
            "intrinsic: [" + f(1111.1) + ", " + f(0) + ", " + f(400) + ",\n"
            "            " + f(0) + ", " + f(1111.1) + ", " + f(300.5) + ",\n"
            "            " + f(0) + ", " + f(0) + ", " + f(1) + "]\n"
            "normalizing_center: [" + f(0.25) + ", " + f(-1.5) + ", " + f(3.0000001) + "]\n"
            "normalizing_radius: " + f(radius) + "\n")
    assert text == want
    n, h, w, K2, c2, r2 = host.load_inference_params(str(tmp_path))
    assert (n, h, w) == (50, 600, 800)
    assert torch.allclose(K2, K, atol=1e-4) and torch.allclose(c2, center, atol=1e-6)
    assert abs(r2 - radius) < 1e-6
    with pytest.raises(RuntimeError):
        host.load_inference_params(str(tmp_path / "nowhere"))
