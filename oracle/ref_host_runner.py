#!/usr/bin/env python3
"""Runs the reference's own Renderer (oracle/_ref/_f2nerf_ref_host.so: its host sources unmodified +
oracle/ref_cuda_side.cpp over the C ABI) on cuda:0 in a process of its own and stores what it
computed.  TEST INFRASTRUCTURE: called by tests/test_gpu_ref_host.py.

  python oracle/ref_host_runner.py in.pt out.pt

in.pt : {params: {name: tensor}, rays_o, rays_d, emb_idx, gt, seed, train, var_weight,
         image: optional {pose [3,4], intrinsic [3,3], h, w, batch},
         save_checkpoint: optional path -- torch::save(renderer_, path) as train_manager.cpp:132-136,
         load_checkpoint: optional path -- torch::load into a fresh Renderer as localizer.cpp:37-39
                          (then `params` is ignored)}
out.pt: {colors, depths, weights, idx_start_end, loss, mse, grads: {name: tensor}, image: (c, z)}

TRAIN: the reference draws its step noise and background itself (src/points_sampler.cpp:35,
src/renderer.cpp:42-44: two torch::rand calls on the device generator); torch.manual_seed(seed)
right before render() makes them reproducible by the caller.  The loss is the expression of
src/main_functions/train_manager.cpp:78-93 on the reference's own outputs and WeightVar."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(path_in, path_out):
    from oracle import build_ref

    M = build_ref.load_host()
    assert M is not None, "oracle/_ref/_f2nerf_ref_host.so is not built"
    dev = torch.device("cuda:0")
    d = torch.load(path_in, weights_only=True)
    n_images = d["params"]["app_emb"].shape[0]
    ren = M.Renderer(n_images)
    params = ren.named_parameters()
    assert set(params) == set(d["params"]), (sorted(params), sorted(d["params"]))
    if d.get("load_checkpoint"):
        ren.load(d["load_checkpoint"])
    else:
        with torch.no_grad():
            for k, v in d["params"].items():
                assert tuple(params[k].shape) == tuple(v.shape), k
                params[k].copy_(v.to(dev))
    if d.get("save_checkpoint"):
        ren.save(d["save_checkpoint"])
    to = lambda x: x.to(dev)
    out = {}
    emb = to(d["emb_idx"]) if d["train"] else torch.empty(0, dtype=torch.int32, device=dev)
    torch.manual_seed(int(d["seed"]))
    colors, depths, weights, idx = ren.render(to(d["rays_o"]), to(d["rays_d"]), emb, bool(d["train"]))
    out.update(colors=colors.detach().cpu(), depths=depths.detach().cpu(), weights=weights.detach().cpu(),
               idx_start_end=idx.cpu())
    if d["train"]:
        gt = to(d["gt"])
        color_loss = torch.sqrt((colors - gt).square() + 1e-4).mean()       # train_manager.cpp:78
        var = M.weight_var(weights, idx)                                    # :82
        var_loss = (var + 1e-2).sqrt().mean()                               # :83
        loss = color_loss + var_loss * float(d["var_weight"])               # :93
        ren.zero_grad()
        loss.backward()                                                     # :104
        out["loss"] = float(loss)
        out["mse"] = float((colors - gt).square().mean())                   # :95
        out["grads"] = {k: (v.grad.detach().cpu() if v.grad is not None else None)
                        for k, v in ren.named_parameters().items()}
    if d.get("image") is not None:
        im = d["image"]
        with torch.no_grad():
            c, z = ren.render_image(to(im["pose"]), to(im["intrinsic"]), int(im["h"]), int(im["w"]), int(im["batch"]))
        out["image"] = (c.cpu(), z.cpu())
    torch.save(out, path_out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
