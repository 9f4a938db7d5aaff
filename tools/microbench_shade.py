#!/usr/bin/env python3
"""Times f2n_shade_fwd / f2n_shade_bwd alone (C ABI) on 8.4 M synthetic samples, with and without
the hidden pre-activations handed from the forward to the backward."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("f2-nerf_amd").capi
if len(sys.argv) > 2 and sys.argv[1] == "--lib":      # an experimental build instead of lib/libf2nerf_hip.so
    capi.LIB_PATH = os.path.abspath(sys.argv[2])
    print("library:", capi.LIB_PATH)
dev = torch.device("cuda:0")
n, C, E = 65536 * 128, 32, 50
g = torch.Generator(device=dev).manual_seed(0)
enc = torch.randn(C, n, device=dev, generator=g) * 0.1
dirs = torch.randn(n, 3, device=dev, generator=g); dirs /= dirs.norm(dim=1, keepdim=True)
# one chunk of rays = one view: every sample carries the same image id (worst case for the
# embedding-gradient atomics)
img = torch.full((n,), 7, device=dev, dtype=torch.int32)
P = [torch.randn(16, C, device=dev) * .3, torch.randn(16, device=dev) * .1, torch.randn(64, 32, device=dev) * .3,
     torch.randn(64, device=dev) * .1, torch.randn(3, 64, device=dev) * .3, torch.randn(3, device=dev) * .1]
emb = torch.randn(E, 16, device=dev) * .1
logit, rgb = torch.empty(n, device=dev), torch.empty(n, 3, device=dev)
dl, dr = torch.randn(n, device=dev), torch.randn(n, 3, device=dev)
denc = torch.empty(C, n, device=dev)
G = [torch.zeros_like(p) for p in P] + [torch.zeros_like(emb)]
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
pre = torch.empty(64, n, device=dev)
print("fwd matrix-core (default) %.3f ms" % t(lambda: capi.call("shade_fwd", enc, C, dirs, img, *P, emb, logit, rgb, None, n)))
capi.set_option("SHADE_VARIANT", 2)
print("fwd matrix-core, 2 waves/SIMD %.3f ms" % t(lambda: capi.call("shade_fwd", enc, C, dirs, img, *P, emb, logit, rgb, None, n)))
capi.set_option("SHADE_VARIANT", 0)
capi.set_option("SHADE_FWD", 1)
print("fwd vector                %.3f ms" % t(lambda: capi.call("shade_fwd", enc, C, dirs, img, *P, emb, logit, rgb, None, n)))
capi.set_option("SHADE_FWD", 0)
print("fwd + save pre %.3f ms" % t(lambda: capi.call("shade_fwd", enc, C, dirs, img, *P, emb, logit, rgb, pre, n)))
bwd = lambda pre_: capi.call("shade_bwd", enc, C, dirs, img, *P, emb, dl, dr, denc, *G, pre_, n)
capi.set_option("SHADE_BWD", 0)
for v in (0, 1):
    capi.set_option("SHADE_VARIANT", v)
    print("bwd matrix-core variant %d  %.3f ms" % (v, t(lambda: bwd(None))))
capi.set_option("SHADE_VARIANT", 0)
capi.set_option("SHADE_BWD", 1)
print("bwd vector, recompute      %.3f ms" % t(lambda: bwd(None)))
print("bwd vector, saved pre      %.3f ms" % t(lambda: bwd(pre)))
