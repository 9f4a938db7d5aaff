"""f2n_shade_fwd / f2n_shade_bwd at small sample counts: what a launch costs before its first sample"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("f2-nerf_amd").capi
dev = torch.device("cuda:0")
C, E = 32, 50
g = torch.Generator(device=dev).manual_seed(0)
P = [torch.randn(16, C, device=dev) * .3, torch.randn(16, device=dev) * .1, torch.randn(64, 32, device=dev) * .3,
     torch.randn(64, device=dev) * .1, torch.randn(3, 64, device=dev) * .3, torch.randn(3, device=dev) * .1]
emb = torch.randn(E, 16, device=dev) * .1
G = [torch.zeros_like(p) for p in P] + [torch.zeros_like(emb)]
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (64, 4096, 65536, 229376, 524288, 2097152, 8388608):
    enc = torch.randn(C, n, device=dev, generator=g) * 0.1
    dirs = torch.randn(n, 3, device=dev, generator=g); dirs /= dirs.norm(dim=1, keepdim=True)
    img = torch.full((n,), 7, device=dev, dtype=torch.int32)
    logit, rgb = torch.empty(n, device=dev), torch.empty(n, 3, device=dev)
    dl, dr = torch.randn(n, device=dev), torch.randn(n, 3, device=dev)
    denc = torch.empty(C, n, device=dev)
    f = t(lambda: capi.call("shade_fwd", enc, C, dirs, img, *P, emb, logit, rgb, None, n))
    b = t(lambda: capi.call("shade_bwd", enc, C, dirs, img, *P, emb, dl, dr, denc, *G, None, n))
    print("n = %8d (%6d strides of 64): fwd %8.1f us   bwd %8.1f us" % (n, n // 64, f, b))
