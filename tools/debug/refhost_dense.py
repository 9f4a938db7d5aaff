import importlib, os, subprocess, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_render as R
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ref_host as T
host = importlib.import_module("f2-nerf_amd").load_host()
dev = torch.device("cuda:0")
E, n_rays, S, seed, vw = 5, 40, 1024, 4242, 1e-2
g = torch.Generator().manual_seed(17); torch.manual_seed(17)
oracle = R.Renderer(E, L=16, F=2, log2_T=19, S=S, step=1.0 / 256, gen=g, feat_init="trained")
with torch.no_grad(): oracle.scene_field.mlp.bias[0] = 0.0
o = torch.randn(n_rays, 3, generator=g) * 0.25; d = torch.randn(n_rays, 3, generator=g)
gt = torch.rand(n_rays, 3, generator=g); emb = torch.randint(0, E, (n_rays,), generator=g).to(torch.int32)
import pathlib
tmp = pathlib.Path(tempfile.mkdtemp())
ref = T._run_reference(tmp, dict(params=T._params_of(oracle), rays_o=o, rays_d=d, emb_idx=emb, gt=gt, seed=seed, train=True, var_weight=vw, image=None))
torch.manual_seed(seed)
noise = ((torch.rand(n_rays * S, device=dev) - 0.5) + 1.0).view(n_rays, S); bg = torch.rand(n_rays, 3, device=dev)
loss, res, mse, psnr = R.train_loss(oracle, o, d, emb, gt, noise.cpu(), bg.cpu(), vw)
hr = host.Renderer(E, n_levels=16, n_channels=2, log2_table=19, max_samples=S, step=1.0 / 256)
hp = hr.named_parameters()
with torch.no_grad():
    for k, v in T._params_of(oracle).items(): hp[k].copy_(v.to(dev))
outs = {"ref": ref["weights"], "oracle": res.weights.detach()}
for fused in (True, False):
    hr.set_fused(fused); hr.set_fused_shade(fused)
    c, z, w, idx = hr.render(o.to(dev), d.to(dev), emb.to(dev), "train", noise, bg)
    outs["ours_fused" if fused else "ours_opbyop"] = w.detach().cpu()
names = list(outs)
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        a, b = outs[names[i]], outs[names[j]]
        dd = (a - b).abs()
        k = int(dd.argmax())
        print("%-12s vs %-12s max abs %.3e at %d (%.6e vs %.6e)  max rel(>1e-6) %.3e" % (names[i], names[j], float(dd.max()), k, float(a[k]), float(b[k]), float((dd / b.abs().clamp_min(1e-6)).max())))
# sampler pieces on both sides
pts_o = R.get_samples(o, d / d.norm(dim=1, keepdim=True) * 0 + d, noise.cpu(), S, 1.0 / 256) if hasattr(R, "get_samples") else None
