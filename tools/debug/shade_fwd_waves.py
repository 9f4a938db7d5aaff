"""f2n_shade_fwd at 2 / 3 / 4 waves per SIMD (F2N_OPT_SHADE_VARIANT 2 / 0 / 3), same inputs, outputs compared"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("f2-nerf_amd").capi
dev = torch.device("cuda:0")
n, C, E = 65536 * 128, 32, 50
g = torch.Generator(device=dev).manual_seed(0)
enc = torch.randn(C, n, device=dev, generator=g) * 0.1
dirs = torch.randn(n, 3, device=dev, generator=g); dirs /= dirs.norm(dim=1, keepdim=True)
img = torch.full((n,), 7, device=dev, dtype=torch.int32)
P = [torch.randn(16, C, device=dev) * .3, torch.randn(16, device=dev) * .1, torch.randn(64, 32, device=dev) * .3,
     torch.randn(64, device=dev) * .1, torch.randn(3, 64, device=dev) * .3, torch.randn(3, device=dev) * .1]
emb = torch.randn(E, 16, device=dev) * .1
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = None
for rep in range(2):
    for v, name in ((0, "3 waves/SIMD"), (3, "4 waves/SIMD"), (2, "2 waves/SIMD")):
        capi.set_option("SHADE_VARIANT", v)
        logit, rgb = torch.empty(n, device=dev), torch.empty(n, 3, device=dev)
        ms = t(lambda: capi.call("shade_fwd", enc, C, dirs, img, *P, emb, logit, rgb, None, n))
        if ref is None: ref = (logit.clone(), rgb.clone())
        print("%-14s %.3f ms  identical to the first: %s" % (name, ms, torch.equal(logit, ref[0]) and torch.equal(rgb, ref[1])))
capi.set_option("SHADE_VARIANT", 0)
