"""Generates tests/golden/*.npz from the CPU oracle (run: python -m tests.golden.make_golden).

These are DATA (seeded inputs + the oracle's outputs), small enough to commit.  The reference holds
no fixtures for this path and cannot be built or run here, so the vectors come from
the oracle itself: they freeze its behaviour, and the GPU parity tests re-check the HIP kernels
against the same files on the GPU box where /root/reference does not exist.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import kernels as K  # noqa: E402
from oracle import ref_render as R  # noqa: E402
from tests import util  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _np(d):
    return {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def case_hash_ref_config():
    """Reference configuration L=16, F=2, T=2^19, overlapping level windows; 256 points incl. the
    corner cases (origin-adjacent, negative scaled coordinates, radius-2 boundary)."""
    fld = util.make_field(16, 2, 19, None, seed=2022)
    pts = util.ball_points(256, seed=1)
    pts[:3] = torch.tensor([[1e-3, -1e-3, 2e-3], [-1.999, -1.999, 0.0], [2.0, 0.0, 0.0]])
    out, idx = K.hash_fwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], 16, 2,
                          fld["T"], fld["stride"], want_idx=True)
    g = torch.Generator().manual_seed(3)
    grad = torch.randn(256, 32, generator=g) * 1e-3
    tg, pg = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                        fld["table"].numel(), 16, 2, fld["T"], fld["stride"], 128.0, True)
    nz = torch.nonzero(tg).reshape(-1)
    # the table itself is regenerated from the seed (2^24 floats are not fixture material)
    return _np(dict(pts=pts, primes=fld["primes"], bias=fld["bias"], mul=fld["mul"], out=out,
                    rows=idx, grad=grad, table_grad_index=nz.to(torch.int32),
                    table_grad_value=tg[nz], pts_grad=pg))


def case_segments():
    idx, n = util.ragged_bounds(24, 80, seed=6)
    g = torch.Generator().manual_seed(6)
    val = torch.rand(n, generator=g)
    w = torch.rand(n, generator=g) * 0.1
    dv = torch.randn(24, generator=g)
    return _np(dict(idx=idx, val=val, sum=K.seg_sum_fwd(val, idx),
                    scan_excl=K.seg_scan_fwd(val, idx, 0), scan_incl=K.seg_scan_fwd(val, idx, 1),
                    scan_bwd_excl=K.seg_scan_bwd(val, idx, 0), w=w, var=K.weight_var_fwd(w, idx),
                    dvar=dv, var_bwd=K.weight_var_bwd(w, idx, dv)))


def case_sh():
    g = torch.Generator().manual_seed(8)
    d = torch.randn(64, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    return _np(dict(dirs=d, sh=K.sh_encode(d, 4)))


def case_render_small():
    """Whole TRAIN render + loss + gradients on a small field (L=4, F=2, T=2^10, S=64)."""
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    ren = R.Renderer(3, L=4, F=2, log2_T=10, S=64, step=4.0 / 64, gen=g, feat_init="trained")
    with torch.no_grad():
        ren.scene_field.mlp.bias[0] = 5.0
    n = 12
    o = torch.randn(n, 3, generator=g) * 0.25
    d = torch.randn(n, 3, generator=g)
    noise = torch.rand(n, 64, generator=g) + 0.5
    bg, gt = torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g)
    emb = torch.randint(0, 3, (n,), generator=g).to(torch.int32)
    loss, res, mse, psnr = R.train_loss(ren, o, d, emb, gt, noise, bg, 1e-2)
    loss.backward()
    out = dict(rays_o=o, rays_d=d, noise=noise, bg=bg, gt=gt, emb_idx=emb, colors=res.colors,
               depths=res.depths, weights=res.weights, idx_start_end=res.idx_start_end,
               loss=loss.reshape(1), feat_pool_grad=ren.scene_field.feat_pool.grad,
               app_emb_grad=ren.app_emb.grad)
    for k, v in ren.named_parameters():
        out["param." + k] = v
    out["param.scene_field.prim_pool"] = ren.scene_field.prim_pool
    return _np(out)


def case_rays():
    """get_rays_from_pose (reference src/rays.cpp:7-28): five cameras, one camera per ray."""
    g = torch.Generator().manual_seed(21)
    E, h, w, n = 5, 37, 53, 200
    poses = torch.randn(E, 3, 4, generator=g)
    K3 = torch.tensor([[1111.1, 0, w / 2], [0, 1100.0, h / 2], [0, 0, 1.0]]).repeat(E, 1, 1)
    K3[:, 0, 0] += torch.arange(E) * 3.0
    cam = torch.randint(0, E, (n,), generator=g).to(torch.int32)
    ij = torch.stack([torch.randint(0, h, (n,), generator=g), torch.randint(0, w, (n,), generator=g)],
                     1).to(torch.int32)
    o, d = R.get_rays_from_pose(poses[cam.long()], K3[cam.long()], ij.float())
    return _np(dict(poses=poses, intrinsics=K3, cam_idx=cam, ij=ij, rays_o=o, rays_d=d))


def case_shade_network():
    """The per-sample network between encode and compositing (reference
    src/hash_3d_anchored.cpp:86, src/renderer.cpp:93-104, src/sh_shader.cpp:22-29) with torch
    autograd for every gradient: 333 samples, C = 32, embedding ids that change inside a stride."""
    g = torch.Generator().manual_seed(31)
    n, C, E, eps = 333, 32, 5, 1e-3
    enc = (torch.randn(n, C, generator=g) * 0.1).to(torch.float16).float().requires_grad_(True)
    dirs = torch.randn(n, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    img = (torch.arange(n) // 37 % E).to(torch.int32)
    P = {"w_h": torch.randn(16, C, generator=g) * 0.3, "b_h": torch.randn(16, generator=g) * 0.1,
         "w1": torch.randn(64, 32, generator=g) * 0.3, "b1": torch.randn(64, generator=g) * 0.1,
         "w2": torch.randn(3, 64, generator=g) * 0.3, "b2": torch.randn(3, generator=g) * 0.1,
         "emb": torch.randn(E, 16, generator=g) * 0.1}
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    d_logit, d_rgb = torch.randn(n, generator=g), torch.randn(n, 3, generator=g)
    h = enc @ P["w_h"].t() + P["b_h"]
    X = torch.cat([torch.ones_like(h[:, :1]), h[:, 1:]], 1) + P["emb"][img.long()]
    X = torch.cat([X, K.sh_encode(dirs, 4)], 1)
    o = torch.relu(X @ P["w1"].t() + P["b1"]) @ P["w2"].t() + P["b2"]
    rgb = (1 + 2 * eps) / (1 + torch.exp(-o)) - eps
    ((h[:, 0] * d_logit).sum() + (rgb * d_rgb).sum()).backward()
    out = dict(enc=enc, dirs=dirs, img=img, d_logit=d_logit, d_rgb=d_rgb, logit=h[:, 0], rgb=rgb,
               d_enc=enc.grad)
    for k, v in P.items():
        out["param." + k] = v
        out["grad." + k] = v.grad
    return _np(out)


CASES = {"hash_ref_config": case_hash_ref_config, "segments": case_segments, "sh": case_sh,
         "render_small": case_render_small, "rays": case_rays, "shade_network": case_shade_network}


if __name__ == "__main__":
    for name, fn in CASES.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fn())
        print("wrote", name)
