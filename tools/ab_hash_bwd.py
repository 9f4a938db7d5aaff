#!/usr/bin/env python3
"""A/B timing of f2n_hash_bwd_binned: the current library (per-tile combine on / off) against a
reference build of an earlier round (tools/ab/libf2nerf_hip_r01.so, if present), same inputs, same
process, interleaved (tools only -- not part of the product or the tests).

  python tools/ab_hash_bwd.py [--config c2|c4|c5|c5s] [--points train|view|rays|ball] [--reps 5]
"""
import argparse
import ctypes
import importlib
import os
import sys

import torch

import bin_stats

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
    return ts[len(ts) // 2]


def make_points(kind, n_rays, S, dev, g):
    """[n_rays * S, 3] contracted sample points, ray-major."""
    if kind == "ball":
        n = n_rays * S
        dd = torch.randn(n, 3, device=dev, generator=g)
        return (dd / dd.norm(dim=1, keepdim=True) * torch.rand(n, 1, device=dev, generator=g) ** (1 / 3) * 2).contiguous()
    step = 4.0 / S if S != 1024 else 1.0 / 256
    if kind in ("view", "train"):
        W, f = 800, 1111.1
        pix = torch.arange(n_rays, device=dev)
        i, j = (pix // W).float(), (pix % W).float()
        d = torch.stack([-torch.ones_like(i), (j - 400) / f, -(i - 400) / f], 1)
        d = d / d.norm(dim=1, keepdim=True)
        o = torch.tensor([1.0, 0.0, 0.0], device=dev).expand(n_rays, 3)
    else:
        o = torch.randn(n_rays, 3, device=dev, generator=g) * 0.3
        d = torch.randn(n_rays, 3, device=dev, generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
    noise = torch.ones(n_rays, S, device=dev)
    if kind == "train":
        noise = torch.rand(n_rays, S, device=dev, generator=g) + 0.5
    t = (noise.cumsum(1) * step).reshape(n_rays, S, 1)
    p = (o[:, None] + d[:, None] * t).reshape(-1, 3)
    nrm = p.norm(dim=1, keepdim=True)
    return torch.where(nrm <= 1, p, (2 - 1 / nrm) * p / nrm).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--points", default="train")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ws-gib", type=float, default=0.0, help="workspace size (0 = recommended)")
    ap.add_argument("--zero-rays", type=float, default=0.0, help="fraction of rays whose gradient is zero")
    ap.add_argument("--grad-scale", type=float, default=1e-3, help="std of the synthetic gradient")
    ap.add_argument("--levels", type=int, default=0, help="override L")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="kernel route (f2n_set_option) for the 'current' library, e.g. BIN_SPLIT=1")
    ap.add_argument("--only", default="", help="time just this library (a --other name, or 'current')")
    ap.add_argument("--other", action="append", default=[], metavar="NAME",
                    help="also time tools/ab/libf2nerf_hip_NAME.so (an experimental build)")
    args = ap.parse_args()
    capi = importlib.import_module("f2-nerf_amd").capi
    for kv in args.option:
        k, v = kv.split("=")
        capi.set_option(k, int(v))
        print("option %s = %s" % (k, v))
    dev = torch.device("cuda:0")
    # (L, F, log2_T, disjoint stride, n_rays, S)
    cfg = {"c2": (16, 2, 19, False, 65536, 128), "c4": (16, 2, 19, False, 512, 1024),
           "c3": (16, 2, 19, False, 65536, 192),
           "c5": (16, 8, 22, True, 1 << 17, 128), "c5s": (16, 8, 22, True, 1 << 15, 128)}[args.config]
    L, F, log2_T, disjoint, n_rays, S = cfg
    if args.levels:
        L = args.levels
    T = 1 << log2_T
    stride = T * F if disjoint else T
    numel = max(T * L * F, stride * (L - 1) + T * F)
    g = torch.Generator(device=dev).manual_seed(0)
    primes = (torch.randint(1 << 28, 1 << 30, (L, 3), device=dev, generator=g) | 1).to(torch.int32)
    bias = torch.rand(L, 3, device=dev, generator=g) * 1000 + 100
    mul = torch.tensor([2.0 ** (7.0 * l / max(L - 1, 1) + 3.0) for l in range(L)], device=dev)
    pts = make_points(args.points, n_rays, S, dev, g)
    n = pts.shape[0]
    C = L * F
    grad = torch.randn(C, n, device=dev, generator=g) * args.grad_scale
    if args.zero_rays > 0:
        dead = torch.rand(n_rays, device=dev, generator=g) < args.zero_rays
        grad.view(C, n_rays, S)[:, dead, :] = 0.0
    tg = torch.zeros(numel, device=dev)
    bytes_alg = 12 + 20 * C
    libs = [("current", capi.lib().cdll)]
    old_path = os.path.join(ROOT, "tools", "ab", "libf2nerf_hip_r01.so")
    if os.path.exists(old_path):
        old = ctypes.CDLL(old_path)
        libs.append(("round 1", old))
    for name in args.other:
        libs.append((name, ctypes.CDLL(os.path.join(ROOT, "tools", "ab", "libf2nerf_hip_%s.so" % name))))
    if args.only:
        libs = [(nm, lb) for nm, lb in libs if nm == args.only]
    print("config %s: n=%d (%d rays x %d) L=%d F=%d T=2^%d, points=%s" % (args.config, n, n_rays, S, L, F, log2_T, args.points))
    stream = torch.cuda.current_stream().cuda_stream
    for name, lib in libs:
        lib.f2n_hash_bwd_workspace_bytes.restype = ctypes.c_int64
        lib.f2n_hash_bwd_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
        need = lib.f2n_hash_bwd_workspace_bytes(n, L, F, T)
        if need <= 0:
            print("  %-22s binned backward not applicable" % name)
            continue
        if args.ws_gib > 0 and name == "current":
            need = int(args.ws_gib * 2 ** 30) // 256 * 256
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        fn = lib.f2n_hash_bwd_binned
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int64,
                                               ctypes.c_float, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]

        def run():
            st = fn(pts.data_ptr(), primes.data_ptr(), bias.data_ptr(), mul.data_ptr(), grad.data_ptr(), 1, n,
                    tg.data_ptr(), n, L, F, T, stride, 128.0, ws.data_ptr(), need, stream)
            assert st == 0, st
        variants = [("", None)]
        if name == "current":
            variants = [("combine on", 0), ("combine off", 1)]
        for label, opt in variants:
            if opt is not None:
                capi.set_option("BWD_COMBINE", opt)
            ms = timeit(run, args.reps)
            print("  %-10s %-12s %8.3f ms  %7.1f GB/s algorithmic  (ws %.1f GiB)" %
                  (name, label, ms, n * bytes_alg / ms / 1e6, need / 2 ** 30))
        if name == "current":
            capi.set_option("BWD_COMBINE", 0)
            stats = bin_stats.enable(lib, dev)
            over = torch.zeros(1, dtype=torch.int64, device=dev)
            lib.f2n_hash_bwd_set_overflow_counter.argtypes = [ctypes.c_void_p]
            lib.f2n_hash_bwd_set_overflow_counter(over.data_ptr())
            run()
            bin_stats.report(lib, stats, L)
            lib.f2n_hash_bwd_set_overflow_counter(None)
            print("    records applied with float atomics (past queue, run AND arena): %d of %d"
                  % (int(over.item()), n * L * 8))
        del ws


if __name__ == "__main__":
    main()
