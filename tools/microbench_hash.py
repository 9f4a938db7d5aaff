#!/usr/bin/env python3
"""Kernel-level micro-benchmark of the hash-grid kernels through the C ABI (ctypes), timed with
torch.cuda events on the current stream (the stream the kernels are launched on).

  python tools/microbench_hash.py [--config c2|c5|c1] [--n N] [--reps R]

Reports ms and algorithmic GB/s (SURVEY.md 8d bytes) per variant; used to choose layouts/kernels.
"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--points", default="rays", choices=["rays", "ball", "view", "view_sm", "view_jit"],
                    help="rays: random rays, samples of a ray consecutive; view: 65536 consecutive pixels of "
                         "an 800-wide pinhole view (focal 1111, camera at distance 1), ray-major as the "
                         "renderer emits them; view_sm: the same points sample-major (lane = adjacent pixel)")
    ap.add_argument("--fwd-only", action="store_true")
    args = ap.parse_args()
    capi = importlib.import_module("f2-nerf_amd").capi
    dev = torch.device("cuda:0")
    cfg = {"c2": (16, 2, 19, None, 65536 * 128), "c1": (4, 2, 19, None, 65536 * 64),
           "c4": (16, 2, 19, None, 512 * 1024),
           "c5": (16, 8, 22, "disjoint", 1 << 24)}[args.config]
    L, F, log2_T, stride_mode, n = cfg
    if args.n:
        n = args.n
    T = 1 << log2_T
    stride = T if stride_mode is None else T * F
    numel = max(T * L * F, stride * (L - 1) + T * F)
    g = torch.Generator(device=dev).manual_seed(0)
    table = (torch.randn(numel, device=dev, generator=g) * 0.1)
    table16 = torch.empty(numel, dtype=torch.int16, device=dev)
    capi.call("table_to_f16", table, table16, numel)
    primes = torch.randint(1 << 28, 1 << 30, (L, 3), device=dev, generator=g).to(torch.int32) | 1
    bias = torch.rand(L, 3, device=dev, generator=g) * 1000 + 100
    mul = torch.tensor([2.0 ** (7.0 * l / max(L - 1, 1) + 3.0) for l in range(L)], device=dev)
    if args.points == "rays":
        # consecutive samples along rays (what the renderer feeds): 128 per ray, step 1/32
        # (config c4: the reference's 1024 steps of 1/256)
        S = 1024 if args.config == "c4" else 128
        R = n // S
        o = torch.randn(R, 1, 3, device=dev, generator=g) * 0.3
        d = torch.randn(R, 1, 3, device=dev, generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        t = (torch.arange(1, S + 1, device=dev).float() * (4.0 / S)).reshape(1, S, 1)
        p = (o + d * t).reshape(-1, 3)
        nrm = p.norm(dim=1, keepdim=True)
        pts = torch.where(nrm <= 1, p, (2 - 1 / nrm) * p / nrm).contiguous()
        n = pts.shape[0]
    elif args.points in ("view", "view_sm", "view_jit"):
        S, R, W, f = 128, n // 128, 800, 1111.1
        pix = torch.arange(R, device=dev)
        i, j = (pix // W).float(), (pix % W).float()
        # camera at (1, 0, 0) looking at the origin (-x), pixel (i, j) -> direction
        d = torch.stack([-torch.ones_like(i), (j - 400) / f, -(i - 400) / f], 1)
        d = d / d.norm(dim=1, keepdim=True)
        o = torch.tensor([1.0, 0.0, 0.0], device=dev).expand(R, 3)
        t = ((torch.arange(S, device=dev).float() + 0.5) * (4.0 / S)).reshape(1, S, 1)
        if args.points == "view_jit":                          # TRAIN: every step scaled by U[0.5, 1.5)
            t = ((torch.rand(R, S, device=dev, generator=g) + 0.5).cumsum(1) * (4.0 / S)).reshape(R, S, 1)
        p = (o[:, None] + d[:, None] * t)                      # [R, S, 3]
        if args.points == "view_sm":
            p = p.transpose(0, 1)                              # [S, R, 3]
        p = p.reshape(-1, 3)
        nrm = p.norm(dim=1, keepdim=True)
        pts = torch.where(nrm <= 1, p, (2 - 1 / nrm) * p / nrm).contiguous()
        n = pts.shape[0]
    else:
        dd = torch.randn(n, 3, device=dev, generator=g)
        pts = (dd / dd.norm(dim=1, keepdim=True) * torch.rand(n, 1, device=dev, generator=g) ** (1 / 3) * 2).contiguous()
    C = L * F
    bytes_fwd = 12 + 16 * C + 4 * C
    print("config %s: n=%d L=%d F=%d T=2^%d table %.0f MiB f16, points=%s" %
          (args.config, n, L, F, log2_T, numel * 2 / 2 ** 20, args.points))
    out_rm = torch.empty(n, C, device=dev)
    out_cm = torch.empty(C, n, device=dev)
    for name, out, ldp, ldc in (("fwd row-major [n,C]", out_rm, C, 1), ("fwd chan-major [C,n]", out_cm, 1, n)):
        med, best = timeit(lambda: capi.call("hash_fwd", pts, table16, primes, bias, mul, out, ldp, ldc,
                                             None, n, L, F, T, stride), args.reps)
        print("  %-28s %8.3f ms (best %8.3f)  %7.1f GB/s algorithmic" % (name, med, best, n * bytes_fwd / med / 1e6))
    if args.points in ("rays", "view", "view_jit") and n % 128 == 0:
        S_rt = 1024 if args.config == "c4" else 128
        for walk, name in ((0, "auto"), (1, "across, one sample index"), (3, "across, depth order"), (2, "along a ray")):
            capi.set_option("RAYTILE_WALK", walk)
            med, best = timeit(lambda: capi.call("hash_fwd_raytile", pts, table16, primes, bias, mul, out_cm,
                                                 n // S_rt, S_rt, L, F, T, stride), args.reps)
            print("  %-28s %8.3f ms (best %8.3f)  %7.1f GB/s algorithmic" %
                  ("fwd ray-tile, " + name, med, best, n * bytes_fwd / med / 1e6))
        capi.set_option("RAYTILE_WALK", 0)
    if args.fwd_only:
        return
    grad_rm = torch.randn(n, C, device=dev, generator=g) * 1e-3
    grad_cm = grad_rm.t().contiguous()
    tg = torch.zeros(numel, device=dev)
    for mode in ("atomic", "sliced"):
        capi.set_option("HASH_BWD", {"atomic": 1, "sliced": 2}[mode])
        for name, gr, ldp, ldc in (("row-major", grad_rm, C, 1), ("chan-major", grad_cm, 1, n)):
            try:
                med, best = timeit(lambda: capi.call("hash_bwd", pts, table16, primes, bias, mul, gr, ldp, ldc,
                                                     tg, None, n, L, F, T, stride, 128.0), max(2, args.reps // 2))
                print("  bwd %-7s grads %-10s %8.3f ms (best %8.3f)  %7.1f GB/s algorithmic" %
                      (mode, name, med, best, n * bytes_fwd / med / 1e6))
            except Exception as e:
                print("  bwd %s %s: %s" % (mode, name, e))
    capi.set_option("HASH_BWD", 0)
    need = capi.lib().cdll.f2n_hash_bwd_workspace_bytes(n, L, F, T)
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        for name, gr, ldp, ldc in (("row-major", grad_rm, C, 1), ("chan-major", grad_cm, 1, n)):
            med, best = timeit(lambda: capi.call("hash_bwd_binned", pts, primes, bias, mul, gr, ldp, ldc, tg,
                                                 n, L, F, T, stride, 128.0, ws, need), max(2, args.reps // 2))
            print("  bwd binned  grads %-10s %8.3f ms (best %8.3f)  %7.1f GB/s algorithmic  (ws %.1f GiB)" %
                  (name, med, best, n * bytes_fwd / med / 1e6, need / 2 ** 30))


if __name__ == "__main__":
    main()
