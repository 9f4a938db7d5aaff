#!/usr/bin/env python3
"""bench.py -- rendered rays/sec (forward + backward) of the F2-NeRF hot path on MI355X.

One "step" = one full 800x800 synthetic view (640 000 rays) pushed through
Renderer::render(TRAIN) + loss + backward in 65 536-ray chunks (BASELINE.json configs[1]:
128 samples/ray, L=16 hash levels, F=2, T=2^19, one GPU).  The optimiser step is excluded, as in
BASELINE.md.  Rays, step noise inputs (drawn on device per chunk, as the reference does), ground
truth colours and all parameters are resident in HBM before the timed region starts.  A view's rays
are handed over row by row, as the reference's render_all_rays walks a view (--pixel-tiles B: BxB
pixel tiles instead, the order Renderer::render_image uses itself; same rays, same chunk sizes; the
JSON line says which in config.pixel_order, and carries the 8x8-tile figure of the same build as the
separate object "pixel_tiles_8x8" -- never as value).

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (child processes through
torch.distributed.run, before this process touches the GPU); under an external launcher
(RANK/LOCAL_RANK/WORLD_SIZE set) it is one of the ranks and WORLD_SIZE must equal --gpus.  Rays shard
embarrassingly -- every rank renders its own view per step (weak scaling); the only collective is one
RCCL all-reduce per step of {sum of squared error, value count} for the global PSNR.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- the dominant hand-written kernel's algorithmic bytes / its hipEvent-timed duration
  hash_fwd_roofline -- the same for the hash-grid gather (the kernel the north star's 40 % is about)
  kernels.hash_bwd_nonzero -- the table-gradient kernels timed again on the same points with a
                  gradient whose contributions are all non-zero (the headline's untrained network
                  leaves 90 % of them below the f16 underflow); outside value
  collective   -- backend / world size / latency of the path's one all-reduce (RCCL, also at N = 1)
  c5           -- BASELINE config C5 (hash kernels alone, T = 2^22, F = 8) run once after the
                  headline: roofline (backward), roofline_fwd, cpu_baseline; outside value
  cpu_baseline -- the CPU oracle (a port of the reference arranged as the reference is) timed on
                  this box's host cores on a bounded sample of the same workload (N=1 only).
--workload c3 / c4 / c5 select the other BASELINE configs (presets of the flags below).
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 vector = f32 MFMA peak
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--levels", type=int, default=16)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--log2-table", type=int, default=19)
    ap.add_argument("--chunk", type=int, default=65536)
    ap.add_argument("--regime", choices=["dense", "terminating"], default="dense")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=4096)
    ap.add_argument("--n-images", type=int, default=50)
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; nccl = RCCL (default). 'gloo' + --share-gpu lets "
                         "several ranks rehearse the multi-rank path on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal only)")
    ap.add_argument("--workload", choices=["render", "c2", "c3", "c4", "c5"], default="render",
                    help="render / c2: the headline (BASELINE config C2 shape by default; C1 through "
                         "--levels/--samples).  c3: BASELINE config C3 as BASELINE.md states it -- 120 "
                         "poses on a 3-turn expanding spiral (r 0.2 -> 1.0, z -0.3 -> 0.3, look-ahead "
                         "orientation), fx = fy = 1400, 1920x1080, 192 samples/ray.  c4: BASELINE config "
                         "C4's per-GPU shape -- 512 random rays per step, 1024 samples of 1/256 (the "
                         "reference's own batch).  c5: BASELINE config C5, the "
                         "hash-grid kernels alone (forward + backward) on 2^24 uniform points of the "
                         "radius-2 ball, T = 2^22, L = 16, F = 8, 1 GiB f16 table -- the HBM stress; "
                         "reports points/s with a roofline object per kernel")
    ap.add_argument("--focal", type=float, default=1111.1, help="fx = fy of the synthetic cameras")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline: skip the separately timed extra legs of the default line "
                         "(8x8 pixel tiles, non-zero-gradient table backward, config C5)")
    ap.add_argument("--no-grad-in-place", action="store_true",
                    help="A/B: hand every chunk's table gradient to autograd (a 64 MiB fill and add per "
                         "chunk) instead of adding into feat_pool.grad directly")
    ap.add_argument("--no-collective-at-1", action="store_true",
                    help="at --gpus 1 do not create the one-rank RCCL group (the path's all-reduce then "
                         "is skipped, as in rounds 1-2)")
    ap.add_argument("--c5-points", type=int, default=1 << 24)
    ap.add_argument("--c5-workspace-gib", type=float, default=0.0,
                    help="scratch for the binned backward (0 = the library's recommendation)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="kernel route for A/B measurements (f2n_set_option), e.g. BWD_COMBINE=1; "
                         "recorded in the JSON line")
    ap.add_argument("--pixel-tiles", type=int, default=0, metavar="B",
                    help="order in which a view's pixels are handed to the renderer: 0 = row by row (default: "
                         "the reference's render_all_rays order, comparable with round 1 and BASELINE) or BxB "
                         "pixel tiles (the Renderer's own traversal in render_image).  Same rays and chunk "
                         "sizes either way; falls back to rows when B does not divide the image")
    ap.add_argument("--debug-bin-stats", action="store_true",
                    help="print the binned backward's per-level counters of the timed steps on stderr")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / collective check without a GPU: every rank joins the "
                         "process group (use --backend gloo), shards a synthetic error vector, runs the "
                         "path's one all-reduce and rank 0 prints a JSON line with n_gpus -- no kernels, "
                         "no rays/s (tests/test_bench_launcher_cpu.py)")
    ap.add_argument("--rays", type=int, default=0,
                    help="rays per step per GPU instead of a full view (e.g. 512 = BASELINE config C4 "
                         "with --samples 1024); pixels drawn at random like dataset.cpp:153-155")
    ap.add_argument("--render-images", type=int, default=0,
                    help="after the headline measurement, time K inference renders of a full view "
                         "(Renderer::render_image, VALIDATE mode, forward only; the reference's own "
                         "self-reported figure is seconds per rendered image, src/main_functions/"
                         "test.cpp:35-57) and report them as \"render_image\" -- not part of value")
    ap.add_argument("--graph-iters", type=int, default=-1,
                    help="after the headline measurement, capture ONE training batch of the reference's "
                         "shape (device-side ray draw + TRAIN render + loss + backward, no host read: "
                         "Renderer deferred_check) as a hipGraph and time K replays next to K eager "
                         "batches; reported as \"graphed_batch\" -- not part of value.  Default: 200 for "
                         "--workload c4, otherwise off")
    ap.add_argument("--train-iters", type=int, default=0,
                    help="after the headline measurement, time K complete data-parallel TRAINING "
                         "iterations of the reference's own shape (512 random rays per GPU drawn on the "
                         "device, 1024 samples, fwd + bwd + gradient all-reduce + fused Adam; SURVEY 8f "
                         "ranks 1, 2, 4) and report them as \"train_iteration\" -- not part of value")
    args = ap.parse_args()
    explicit = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
    preset = {"c3": {"height": 1080, "width": 1920, "samples": 192, "n_images": 120, "focal": 1400.0},
              "c4": {"rays": 512, "samples": 1024}}.get(args.workload, {})
    for k, v in preset.items():          # a preset fills in what the command line did not say
        if "--" + k.replace("_", "-") not in explicit:
            setattr(args, k, v)
    return args


def spiral_poses(n, turns=3.0, r0=0.2, r1=1.0, z0=-0.3, z1=0.3):
    """BASELINE.md config C3 / SURVEY 8(d) "free-trajectory": n poses on an expanding spiral, each
    camera looking ahead along the path (-z forward, y up as far as the path allows), then
    normalised the way the reference normalises a dataset (src/dataset.cpp:77-86)."""
    u = torch.linspace(0.0, 1.0, n, dtype=torch.float64)

    def path(v):
        th, r = 2.0 * math.pi * turns * v, r0 + (r1 - r0) * v
        return torch.stack([r * torch.cos(th), r * torch.sin(th), z0 + (z1 - z0) * v], 1)

    pos = path(u)
    fwd = path(u + 1e-4) - pos
    fwd = fwd / fwd.norm(dim=1, keepdim=True)
    zc = -fwd
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64).expand(n, 3)
    xc = torch.cross(up, zc, dim=1)
    xc = xc / xc.norm(dim=1, keepdim=True)
    yc = torch.cross(zc, xc, dim=1)
    center = pos.mean(0)
    radius = (pos - center).norm(dim=1).max()
    pos_n = (pos - center) / radius
    return torch.cat([torch.stack([xc, yc, zc], 2), pos_n.unsqueeze(2)], 2).float()  # [n,3,4]


def camera_poses(args):
    return spiral_poses(args.n_images) if args.workload == "c3" else fox_like_poses(args.n_images)


def fox_like_poses(n, seed=2022):
    """n camera poses on a 200-degree arc of radius 1 around the origin, heights U[-0.2, 0.2],
    looking at the origin (OpenGL convention: -z forward, y up), then normalised the way the
    reference normalises a dataset (src/dataset.cpp:77-86: centre = mean position, scale = max
    distance from it)."""
    g = torch.Generator().manual_seed(seed)
    ang = torch.linspace(-100.0, 100.0, n) * math.pi / 180.0
    pos = torch.stack([torch.cos(ang), torch.sin(ang), torch.rand(n, generator=g) * 0.4 - 0.2], 1)
    zc = pos / pos.norm(dim=1, keepdim=True)          # camera z axis points away from the target
    up = torch.tensor([0.0, 0.0, 1.0]).expand(n, 3)
    xc = torch.cross(up, zc, dim=1)
    xc = xc / xc.norm(dim=1, keepdim=True)
    yc = torch.cross(zc, xc, dim=1)
    center = pos.mean(0)
    radius = (pos - center).norm(dim=1).max()
    pos_n = (pos - center) / radius
    return torch.cat([torch.stack([xc, yc, zc], 2), pos_n.unsqueeze(2)], 2)  # [n,3,4]


def view_rays(H, pose, intr, h, w):
    """All pixels of one view, row-major: one f2n_gen_rays launch (no pixel-grid tensor)."""
    o, d = H.get_view_rays(pose, intr, h, w)
    return o, d


def algorithmic_bytes(kernel, L, F, S):
    """bytes one unit must move if every byte is touched once."""
    if kernel == "hash_fwd":   # per sample: xyz in, 8 corners x F f16 per level, L*F f32 out
        return 12 + 16 * L * F + 4 * L * F
    if kernel == "hash_bwd":   # per sample: xyz + L*F f32 grads in, 8 corners x F f32 RMW counted once
        return 12 + 4 * L * F + 16 * L * F
    if kernel == "density_march":  # per ray (dense regime): o,d in, S x (noise + 8 corners x F f16 x L), count out
        return 24 + S * (4 + 16 * L * F) + 4
    if kernel == "density_scan":   # per ray: S x (L*F f32 features + dt) in, count out
        return S * (4 * L * F + 4) + 4
    raise KeyError(kernel)


def algorithmic_flops(kernel, L, F):
    """f32 flops one sample needs in the per-sample network kernels (2 per multiply-add, no padding):
    head C x 16, hidden 32 x 64, output 64 x 3; the backward recomputes the forward and adds the
    data gradients (64 x 3, 64 x 16, 16 x C) and the weight gradients (3 x 64, 64 x 32, 16 x C)."""
    C = L * F
    fwd = 16 * C + 32 * 64 + 64 * 3
    if kernel == "shade_fwd":
        return 2 * fwd
    if kernel == "shade_bwd":
        return 2 * (fwd + (64 * 3 + 64 * 16 + 16 * C) + (3 * 64 + 64 * 32 + 16 * C))
    raise KeyError(kernel)


# kernels behind each hipEvent-timed operator, for the PMC traffic lookup
OP_KERNELS = {
    "hash_fwd": ["hash_fwd_kernel", "hash_fwd_raytile_kernel"],
    "shade_fwd": ["shade_fwd_kernel"],
    "shade_bwd": ["shade_bwd_mfma_kernel", "shade_bwd_kernel"],
    "hash_bwd": ["hash_bwd_bin_kernel", "hash_bwd_split_kernel", "hash_bwd_reduce_kernel",
                 "hash_bwd_reduce_runs_kernel"],
    "density_march": ["density_march_kernel"],
    "density_scan": ["density_scan_kernel"],
}


def kernels_sha16():
    """Fingerprint of the kernel sources (the same function stamps a PMC summary when it is made):
    a traffic figure is only quoted for the kernels it was measured on."""
    import hashlib
    kdir = os.path.join(ROOT, "f2-nerf_amd", "csrc", "kernels")
    h = hashlib.sha256()
    for f in sorted(os.listdir(kdir)):
        if f.endswith((".hip", ".hiph")):
            h.update(f.encode())
            h.update(open(os.path.join(kdir, f), "rb").read())
    return h.hexdigest()[:16]


def workload_key(args):
    """Name of the workload for the PMC lookup: "c2" for the default shape (BASELINE config C2, dense
    regime), otherwise a key no committed summary carries -- traffic measured on one shape is never
    quoted for another."""
    default = (args.height == 800 and args.width == 800 and args.samples == 128 and args.levels == 16 and
               args.channels == 2 and args.log2_table == 19 and args.chunk == 65536 and args.rays == 0 and
               args.regime == "dense" and args.pixel_tiles == 0 and args.workload in ("render", "c2"))
    if default:
        return "c2"
    return "%dx%d_S%d_L%d_F%d_T%d_chunk%d_rays%d_%s_tiles%d" % (
        args.height, args.width, args.samples, args.levels, args.channels, args.log2_table, args.chunk,
        args.rays, args.regime, args.pixel_tiles)


def pmc_traffic(op, workload_key="c2"):
    """(HBM-side bytes per launch of `op`, provenance) from the newest committed rocprofv3 --pmc
    summary for this workload (profiles/r*_pmc_traffic*.json, made by tools/pmc_summary.py from
    separate FETCH_SIZE and WRITE_SIZE passes of this same command; FETCH_SIZE doubled as the gfx950
    guide prescribes).  PMC counters cannot be read inside the timed run, so this is a lookup; it is
    refused (None, with the reason) when the summary was taken on other kernel sources."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")))
    files = [f for f in files if (json.load(open(f)).get("_meta", {}).get("workload", "c2") == workload_key)]
    if not files:
        return None, {"file": None, "reason": "no PMC summary committed for this workload"}
    src = {"file": os.path.relpath(files[-1], ROOT)}
    try:
        d = json.load(open(files[-1]))
        meta = d.get("_meta", {})
        src.update({k: meta.get(k) for k in ("kernels_sha16", "git_sha", "command") if meta.get(k)})
        if meta.get("kernels_sha16") != kernels_sha16():
            src["reason"] = ("stale: kernel sources changed since this summary was taken "
                             "(now %s)" % kernels_sha16())
            return None, src
        tot = 0.0
        launches = meta.get("operator_launches", 0)
        for k in OP_KERNELS.get(op, []):
            if k in d:
                per_dispatch = d[k].get("read_bytes_corrected", 0.0) + d[k].get("write_bytes", 0.0)
                # an operator launch may be several dispatches of a kernel (rounds of the C5 backward)
                mult = d[k].get("dispatches_per_pass", launches) / launches if launches else 1.0
                tot += per_dispatch * mult
        return (tot or None), src
    except (KeyError, ValueError, OSError) as e:
        src["reason"] = "unreadable: %r" % (e,)
        return None, src


def usable_cores():
    """Host cores this process may actually run on: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def tiles_active(args, B=None):
    B = args.pixel_tiles if B is None else B
    return args.rays == 0 and B > 1 and args.height % B == 0 and args.width % B == 0


def pixel_order(h, w, B, device="cpu"):
    """Row-major pixel ids in the order a view is handed to the renderer: BxB tiles, or rows (B <= 1)."""
    ids = torch.arange(h * w, device=device)
    if B > 1:
        ids = ids.view(h // B, B, w // B, B).permute(0, 2, 1, 3).reshape(-1)
    return ids


def cpu_baseline(args, n_rays):
    """The oracle = CPU port arranged as the reference is (torch-CPU ATen ops + C/OpenMP restatement
    of the 14 CUDA kernels), same workload shape, bounded ray count, SAME RAY ORDER as the GPU arm:
    a run of n_rays consecutive rays of view 0 in the order the GPU arm walks it (starting at the
    middle of the view), or random pixels when the GPU arm draws random pixels (--rays)."""
    from oracle import kernels as K
    from oracle import ref_render as R

    cores = int(os.environ.get("F2N_CPU_THREADS", "0")) or min(usable_cores(), 64)
    torch.set_num_threads(cores)
    K.set_num_threads(cores)
    g = torch.Generator().manual_seed(2022)
    torch.manual_seed(2022)
    ren = R.Renderer(args.n_images, L=args.levels, F=args.channels, log2_T=args.log2_table,
                     S=args.samples, step=4.0 / args.samples, gen=g, feat_init="trained")
    ren.scene_field.parallel_bwd = True
    if args.regime == "terminating":
        with torch.no_grad():
            ren.scene_field.mlp.bias[0] = 8.0
    poses = camera_poses(args)
    intr = torch.tensor([[args.focal, 0, args.width / 2], [0, args.focal, args.height / 2], [0, 0, 1.0]])
    if args.rays > 0:
        pix = torch.randint(0, args.height * args.width, (n_rays,), generator=g)
        order = "random pixels"
    else:
        ids = pixel_order(args.height, args.width, args.pixel_tiles if tiles_active(args) else 0)
        lo = max(0, min(ids.numel() // 2, ids.numel() - n_rays))
        pix = ids[lo:lo + n_rays]
        n_rays = pix.numel()
        order = "consecutive rays %d.. of view 0 in the GPU arm's order (%s)" % (
            lo, "%dx%d tiles" % (args.pixel_tiles, args.pixel_tiles) if tiles_active(args) else "rows")
    ij = torch.stack([pix // args.width, pix % args.width], 1)
    o, d = R.get_rays_from_pose(poses[0:1], intr[None], ij)
    gt = torch.rand(n_rays, 3, generator=g)
    emb = torch.zeros(n_rays, dtype=torch.int32)
    # BASELINE.md section 2 / SURVEY 8(d): median of >= 5 timed runs after 2 warm-ups, same seeded
    # inputs shape as the GPU arm; bounded to ~1 minute of wall time whatever the host is
    n_warm, n_timed, times = 2, 5, []
    t_begin = time.perf_counter()
    for it in range(n_warm + n_timed):
        noise = torch.rand(n_rays, args.samples, generator=g) + 0.5
        bg = torch.rand(n_rays, 3, generator=g)
        ren.zero_grad()
        t0 = time.perf_counter()
        loss, _, _, _ = R.train_loss(ren, o, d, emb, gt, noise, bg, 0.0)
        loss.backward()
        if it >= n_warm:
            times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_begin > 60 and len(times) >= 1:
            break
    times.sort()
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return {"value": n_rays / med, "unit": "rays/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "runs_s": [round(t, 4) for t in times],
            "sample": "%d rays (%s), S=%d L=%d F=%d T=2^%d, fwd+bwd, median of %d timed "
                      "runs after %d warm-ups"
                      % (n_rays, order, args.samples, args.levels, args.channels, args.log2_table,
                         len(times), n_warm)}


def dry_run(args, rank, world):
    """The multi-rank skeleton of main() with the GPU work left out: process group, per-rank shard,
    the {sum sq err, n} all-reduce (timed per step like the real run), max-over-ranks timing, one
    JSON line on rank 0 carrying the same "collective" object as the real line."""
    pkg = importlib.import_module("f2-nerf_amd")
    dist, collective = (None, None)
    if world > 1:
        dist, collective = init_collective(args, rank, world, "cpu")
    g = torch.Generator().manual_seed(2022)
    err = torch.rand(4099, 3, generator=g, dtype=torch.float64)
    lo, hi = pkg.sharding.shard_range(err.shape[0], rank, world)
    t0 = time.perf_counter()
    us = []
    for _ in range(max(args.steps, 1)):
        t1 = time.perf_counter()
        stat = pkg.sharding.reduce_error_stats(err[lo:hi].square().sum(), err[lo:hi].numel(), dist)
        us.append((time.perf_counter() - t1) * 1e6)
    t_max = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        collective["allreduce_us"] = sorted(us)[len(us) // 2]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "backend": args.backend if world > 1 else None,
                          "collective": collective,
                          "views": [pkg.sharding.view_for(0, r, world, args.n_images) for r in range(world)],
                          "sq_err_sum": float(stat[0]), "n_values": float(stat[1]),
                          "want_sq_err_sum": float(err.square().sum()), "seconds": float(t_max)}),
              flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def run_c5(args, rank, world, dist, dev, pkg):
    """--workload c5: the C5 measurement as its own JSON line."""
    out = c5_measure(args, rank, world, dist, dev, pkg, args.steps, args.warmup,
                     cpu_points=0 if (world > 1 or args.no_cpu_baseline) else 1 << 20)
    if rank == 0:
        print(json.dumps(out), flush=True)


def c5_measure(args, rank, world, dist, dev, pkg, steps, warmup, cpu_points):
    """BASELINE config C5: hash-grid encode forward + table-gradient backward through the C ABI,
    inputs resident in HBM; one step = both kernels over the whole point set.  Returns the JSON
    object on rank 0 (None elsewhere)."""
    capi = pkg.capi
    L, F, log2_T = 16, 8, 22
    T, C, n = 1 << log2_T, 16 * 8, args.c5_points
    stride = T * F                                   # non-overlapping level windows
    numel = T * L * F
    g = torch.Generator(device=dev).manual_seed(2022 + rank)
    table = torch.randn(numel, device=dev, generator=g) * 0.1
    table16 = torch.empty(numel, dtype=torch.int16, device=dev)
    capi.call("table_to_f16", table, table16, numel)
    del table
    primes = (torch.randint(1 << 28, 1 << 30, (L, 3), device=dev, generator=g) | 1).to(torch.int32)
    bias = torch.rand(L, 3, device=dev, generator=g) * 1000 + 100
    mul = torch.tensor([2.0 ** (7.0 * l / (L - 1) + 3.0) for l in range(L)], device=dev)
    dd = torch.randn(n, 3, device=dev, generator=g)
    pts = (dd / dd.norm(dim=1, keepdim=True) * torch.rand(n, 1, device=dev, generator=g) ** (1 / 3) * 2).contiguous()
    del dd
    enc = torch.empty(C, n, device=dev)                                  # channel-major, as the host lib stores it
    grad = torch.randn(C, n, device=dev, generator=g) * 1e-3
    tg = torch.zeros(numel, device=dev)
    need = capi.lib().cdll.f2n_hash_bwd_workspace_bytes(n, L, F, T)
    if need <= 0:
        raise SystemExit("binned backward does not cover config C5")
    if args.c5_workspace_gib > 0:
        need = int(args.c5_workspace_gib * 2 ** 30) // 256 * 256
    ws = torch.empty(need, dtype=torch.uint8, device=dev)

    def fwd():
        capi.call("hash_fwd", pts, table16, primes, bias, mul, enc, 1, n, None, n, L, F, T, stride)

    def bwd():
        capi.call("hash_bwd_binned", pts, primes, bias, mul, grad, 1, n, tg, n, L, F, T, stride, 128.0, ws, need)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        fwd()
        bwd()
    barrier()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    t0 = time.perf_counter()
    for s_ in range(steps):
        evs[s_][0].record()
        fwd()
        evs[s_][1].record()
        bwd()
        evs[s_][2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    t_max = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    del ws, enc, grad, tg, table16, pts
    if rank != 0:
        return None
    ms_f = sum(e[0].elapsed_time(e[1]) for e in evs) / steps
    ms_b = sum(e[1].elapsed_time(e[2]) for e in evs) / steps
    b_f, b_b = algorithmic_bytes("hash_fwd", L, F, 0), algorithmic_bytes("hash_bwd", L, F, 0)

    def roof(op, ms, bytes_unit):
        a = n * bytes_unit / (ms * 1e-3) / 1e9
        traffic, src = pmc_traffic(op, "c5")
        return {"kernel": op, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": a / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                "algorithmic_bytes_per_launch": n * bytes_unit, "avg_launch_ms": ms}

    out = {
        "metric": "hash-grid points/sec (encode fwd + table-gradient bwd), BASELINE config C5",
        "value": n * steps * world / elapsed, "unit": "points/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "C5: %d uniform points in the radius-2 ball, T=2^22 L=16 F=8, 1 GiB f16 table, "
                               "non-overlapping level stride; f2n_hash_fwd + f2n_hash_bwd_binned" % n,
                   "workspace_GiB": need / 2 ** 30, "sharding": "independent point sets per rank"},
        "kernel_options": args.option,
        "roofline": roof("hash_bwd", ms_b, b_b),
        "roofline_fwd": roof("hash_fwd", ms_f, b_f),
        "note": "the forward gathers one random 128-byte line per 16-byte row: it runs at the measured "
                "random-line rate of the memory system (profiles/r02_gather_policy_probe.txt), which caps "
                "the algorithmic fraction at 16/128 of the line traffic",
    }
    if cpu_points > 0:
        try:
            out["cpu_baseline"] = cpu_baseline_c5(L, F, log2_T, n=cpu_points,
                                                  n_warm=2 if cpu_points >= 1 << 20 else 1,
                                                  n_timed=5 if cpu_points >= 1 << 20 else 3)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "points/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    return out


def cpu_baseline_c5(L, F, log2_T, n=1 << 20, n_warm=2, n_timed=5):
    """The oracle's hash encode forward + backward (C/OpenMP restatement of the reference's two CUDA
    kernels) on a bounded sample of config C5: same table size, n points."""
    from oracle import kernels as K

    cores = int(os.environ.get("F2N_CPU_THREADS", "0")) or min(usable_cores(), 64)
    torch.set_num_threads(cores)
    K.set_num_threads(cores)
    T = 1 << log2_T
    g = torch.Generator().manual_seed(5)
    numel = T * L * F
    table16 = K.cast_f16(torch.randn(numel, generator=g) * 0.1)
    primes = (torch.randint(1 << 28, 1 << 30, (L, 3), generator=g) | 1).to(torch.int32)
    bias = torch.rand(L, 3, generator=g) * 1000 + 100
    mul = K.level_mul(L)
    dd = torch.randn(n, 3, generator=g)
    pts = (dd / dd.norm(dim=1, keepdim=True) * torch.rand(n, 1, generator=g) ** (1 / 3) * 2).contiguous()
    grad = torch.randn(n, L * F, generator=g) * 1e-3
    times = []
    for it in range(n_warm + n_timed):
        t0 = time.perf_counter()
        K.hash_fwd(pts, table16, primes, bias, mul, L, F, T, T * F)
        K.hash_bwd(pts, table16, primes, bias, mul, grad, numel, L, F, T, T * F, 128.0, parallel=True)
        if it >= n_warm:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": n / med, "unit": "points/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "runs_s": [round(t, 4) for t in times],
            "sample": "%d points of config C5 (same table), oracle hash fwd + bwd, median of %d after %d "
                      "warm-ups" % (n, n_timed, n_warm)}


def time_train_iterations(args, pkg, H, dev, dist, world, poses, intr):
    """K complete training iterations of the reference's batch (train_manager.cpp:66-107 minus
    logging): 512 rays per rank drawn on the device, S = 1024 / step 1/256, TRAIN render + loss +
    backward, gradient average over the ranks (two collectives), fused Adam with the f16 shadow."""
    ren = H.Renderer(args.n_images, n_levels=args.levels, n_channels=args.channels,
                     log2_table=args.log2_table, max_samples=1024, step=1.0 / 256)
    with torch.no_grad():
        ren.named_parameters()["scene_field.feat_pool"].normal_(0.0, 0.1)
    opt = ren.make_fused_adam(1e-2)
    images = torch.rand(args.n_images, 64, 64, 3, device=dev)   # stand-in ground truth, 64x64 per image
    intr_small = intr.clone()
    intr_small[:2] *= 64.0 / args.width
    intr_all = intr_small[None].expand(args.n_images, 3, 3).contiguous()
    n_coll = 0

    def iteration():
        nonlocal n_coll
        o, d, gt, cam = H.sample_random_rays(poses, intr_all, 64, 64, 512, images)
        ren.zero_grad()
        ren.train_step(o, d, cam, gt, 0.0)
        n_coll = pkg.sharding.allreduce_gradients(list(ren.grads().values()), dist)
        opt.step()

    for _ in range(3):
        iteration()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.train_iters):
        iteration()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.train_iters * 1e3
    return {"ms": ms, "rays_per_iteration": 512 * world, "iterations": args.train_iters,
            "collectives_per_iteration": n_coll,
            "note": "reference batch shape (confs/train_config.yaml:4, points_sampler.hpp:15,39): "
                    "fwd + bwd + gradient all-reduce + fused Adam; not part of value"}


def init_collective(args, rank, world, dev):
    """Process group of the path's one collective.  world > 1: the launcher's ranks over --backend
    ("nccl" IS RCCL on ROCm).  world == 1: a one-rank group over the same backend, so that the
    driver's one-GPU record also shows the backend initialising and its all-reduce running on the
    MI355X (a one-rank all-reduce moves nothing between GPUs; it is the same call path).  Returns
    (torch.distributed or None, the "collective" object of the JSON line)."""
    dev = torch.device(dev)
    info = {"backend": None, "world_size": world, "allreduce_us": None,
            "device_per_rank": ("%s (%s)" % (dev, torch.cuda.get_device_name(dev)) if dev.type == "cuda"
                                else "cpu (dry run)"),
            "payload": "2 x f64 {sum of squared error, value count}, all_reduce(SUM), once per step"}
    if world == 1 and args.no_collective_at_1:
        return None, info
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        probe = torch.ones(2, device=dev, dtype=torch.float64)
        dist.all_reduce(probe)                       # communicator creation happens here, untimed
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if float(probe[0]) != float(world):
            raise RuntimeError("all_reduce(SUM) of ones over %d ranks gave %r" % (world, float(probe[0])))
    except Exception as e:
        if world > 1:
            raise
        info["error"] = "one-rank %s group not available: %r" % (args.backend, e)
        return None, info
    info["backend"] = dist.get_backend()
    info["world_size"] = dist.get_world_size()          # what the backend saw, not what was asked for
    if info["backend"] == "nccl":
        try:
            info["backend"] = "nccl (RCCL %s)" % ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            info["backend"] = "nccl (RCCL)"
    return dist, info


def time_hash_bwd_nonzero(args, pkg, H, ren, first_chunk, dev, reps=7):
    """The table-gradient kernels (f2n_hash_bwd_binned) on the sample points of one chunk of the
    headline workload -- same rays, same TRAIN jitter, same field -- with a gradient whose
    contributions are ALL non-zero (tools/ab_hash_bwd.py's generator: N(0, 1e-3^2) per channel,
    i.e. f16(128 g) w stays far above the f16 underflow), hipEvent-timed.  The headline's own
    gradient comes from a mean loss over 65536 x 3 values of an untrained network and leaves ~90 %
    of the contributions below the underflow, which the kernel skips (as zero-valued adds); a
    512-ray training batch has gradients 128x larger.  This is that case, priced separately."""
    capi = pkg.capi
    o, d = first_chunk
    S, L, F, T = args.samples, args.levels, args.channels, 1 << args.log2_table
    field = ren.scene_field
    sampler = ren.pts_sampler
    pts = sampler.get_samples(o, d, "train")[0].reshape(-1, 3).contiguous()
    n = pts.shape[0]
    x = torch.empty_like(pts)
    capi.call("contract_fwd", pts, x, n)
    C = L * F
    g = torch.Generator(device=dev).manual_seed(7)
    grad = torch.randn(C, n, device=dev, generator=g) * 1e-3
    tg = torch.zeros(field.feat_pool.numel(), device=dev)
    need = capi.lib().cdll.f2n_hash_bwd_workspace_bytes(n, L, F, T)
    if need <= 0:
        return None
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    stride = field.level_stride

    def run():
        capi.call("hash_bwd_binned", x, field.prim_pool, field.bias_pool, field.level_mul, grad, 1, n, tg,
                  n, L, F, T, stride, 128.0, ws, need)

    run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        run()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    ms = ts[len(ts) // 2]
    b = algorithmic_bytes("hash_bwd", L, F, S)
    a = n * b / (ms * 1e-3) / 1e9
    return {"launches": reps, "avg_ms": ms, "units_per_launch": n, "algorithmic_bytes_per_unit": b,
            "achieved_GBs": a, "frac": a / HBM_PEAK_GBS,
            "gradient": "N(0, 1e-3^2) per channel: every f16(128 g) w contribution non-zero",
            "note": "same sample points as one %d-ray chunk of the headline; separately timed through "
                    "the C ABI, median of %d launches; not part of value" % (o.shape[0], reps)}


def time_graphed_batches(args, H, dev, poses, intr, n_iter):
    """The reference's training batch (confs/train_config.yaml:4: 512 rays drawn at random over all
    images as src/dataset.cpp:150-171 does, here on the device; 1024 samples of 1/256; TRAIN render +
    loss + backward) as ONE hipGraph: the Renderer runs without any host read (deferred_check: the
    exact early-stop scan still runs, its verdict accumulates on the device and is read once, after
    the loop), so the ~45 launches of a batch replay as one.  Eager batches of the same Renderer are
    timed next to it."""
    S = args.samples
    ren = H.Renderer(args.n_images, n_levels=args.levels, n_channels=args.channels,
                     log2_table=args.log2_table, max_samples=S, step=1.0 / 256 if S == 1024 else 4.0 / S)
    with torch.no_grad():
        ren.named_parameters()["scene_field.feat_pool"].normal_(0.0, 0.1)
    ren.set_dense_first_pass(1)
    n_rays = args.rays if args.rays > 0 else 512
    images = torch.rand(args.n_images, 64, 64, 3, device=dev)   # stand-in ground truth, 64x64 per image
    intr_small = intr.clone()
    intr_small[:2] *= 64.0 / args.width
    intr_all = intr_small[None].expand(args.n_images, 3, 3).contiguous()

    def batch():
        o, d, gt, cam = H.sample_random_rays(poses, intr_all, 64, 64, n_rays, images)
        ren.zero_grad()
        return ren.train_step(o, d, cam, gt, 0.0)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(3):
        batch()
    ms_eager = timed(batch, n_iter)
    ren.set_deferred_check(True)
    ms_eager_deferred = timed(batch, n_iter)
    out = {"rays_per_batch": n_rays, "samples_per_ray": S, "batches": n_iter, "ms_eager": ms_eager,
           "ms_eager_no_host_read": ms_eager_deferred,
           "note": "device-side ray draw + TRAIN render + loss + backward; not part of value"}
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm-up on a side stream, as graph capture requires
            for _ in range(3):
                batch()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            res = batch()
        out["ms_graph"] = timed(g.replay, n_iter)
        out["loss_after_replay"] = float(res[0])
        out["rays_per_s_graph"] = n_rays / out["ms_graph"] * 1e3
    except Exception as e:   # an extra: never a reason to lose the line
        out["graph_error"] = repr(e)[:300]
    torch.cuda.synchronize()
    out["no_ray_terminated_early"] = bool(ren.deferred_check_ok())
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script as CHILD processes
    through torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) and return their exit code.
    Called before anything touches the GPU: this process never initialises HIP, it only waits."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # decided before any torch.cuda call: the ranks are fresh child processes
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks"
                         % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if not args.share_gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but only %d visible GPUs (use --share-gpu to rehearse)"
                         % (world, torch.cuda.device_count()))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist, collective = init_collective(args, rank, world, dev)

    pkg = importlib.import_module("f2-nerf_amd")
    H = pkg.load_host()
    for kv in args.option:
        name, value = kv.split("=")
        pkg.capi.set_option(name, int(value))
    if args.workload == "c5":
        run_c5(args, rank, world, dist if world > 1 else None, dev, pkg)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    H.manual_seed(2022)          # reference main.cpp:11; identical parameters on every rank
    torch.manual_seed(2022)
    S, L, F = args.samples, args.levels, args.channels
    step_len = 1.0 / 256 if S == 1024 else 4.0 / S   # S = 1024 is the reference's own sampler
    ren = H.Renderer(args.n_images, n_levels=L, n_channels=F, log2_table=args.log2_table,
                     max_samples=S, step=step_len)
    params = ren.named_parameters()
    if args.no_grad_in_place:
        ren.scene_field.set_accumulate_in_place(False)
    with torch.no_grad():
        # "trained-like" table: N(0, 0.1^2) exercises the f16 range and real gradients
        params["scene_field.feat_pool"].normal_(0.0, 0.1)
        if args.regime == "terminating":
            params["scene_field.mlp.bias"][0] = 8.0
    torch.manual_seed(1000 + rank)   # per-rank randomness for noise / background from here on

    poses = camera_poses(args).to(dev)
    intr = torch.tensor([[args.focal, 0, args.width / 2], [0, args.focal, args.height / 2], [0, 0, 1.0]],
                        device=dev)
    n_rays_view = args.rays if args.rays > 0 else args.height * args.width

    def make_views(first_step, n_steps, tiles):
        """Inputs resident in HBM before timing: rays + ground truth of every view this rank renders."""
        views = []
        for s in range(first_step, first_step + n_steps):
            v = pkg.sharding.view_for(s, rank, world, args.n_images)
            o, d = view_rays(H, poses[v], intr, args.height, args.width)
            if tiles_active(args, tiles):
                ids = pixel_order(args.height, args.width, tiles, dev)
                o, d = o[ids].contiguous(), d[ids].contiguous()
            if args.rays > 0:
                pick = torch.randint(0, o.shape[0], (args.rays,), device=dev)
                o, d = o[pick].contiguous(), d[pick].contiguous()
            gt = torch.rand(n_rays_view, 3, device=dev)
            emb = torch.full((n_rays_view,), v, dtype=torch.int32, device=dev)
            views.append((o, d, gt, emb))
        return views

    in_flight = []   # the previous step's all-reduce: (error statistics, work handle)

    def run_step(view, timed=False):
        o, d, gt, emb = view
        ren.zero_grad()
        sq = None
        n_val = 0
        n_samples = 0
        for lo in range(0, n_rays_view, args.chunk):
            hi = min(lo + args.chunk, n_rays_view)
            loss, sq_err, nv, ns = ren.train_step(o[lo:hi], d[lo:hi], emb[lo:hi], gt[lo:hi], 0.0)
            sq = sq_err.double() if sq is None else sq + sq_err.double()
            n_val += nv
            n_samples += ns
        # the only collective of the path: {sum sq err, count} -> global PSNR (RCCL all-reduce), issued
        # asynchronously: the render stream does not wait for it (nothing on the device consumes the
        # result; a blocking 16-byte all-reduce would add its whole latency to a 1 ms training batch).
        # At most one is in flight: the previous step's is waited for first, the last one before the
        # closing barrier of the timed region.
        while in_flight:
            _, work = in_flight.pop()
            if work is not None:
                work.wait()
        stat, work = pkg.sharding.reduce_error_stats(sq, n_val, dist, async_op=True)
        in_flight.append((stat, work))
        return stat, n_samples

    def barrier():
        while in_flight:
            _, work = in_flight.pop()
            if work is not None:
                work.wait()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_leg(views, n_warm):
        """n_warm untimed steps, then the rest timed between barrier + synchronize on both sides;
        returns (elapsed seconds = max over ranks, samples, last error statistics, kernel timings)."""
        for s in range(n_warm):
            run_step(views[s])
        barrier()
        H.kernel_timer_enable(True)
        H.kernel_timer_collect()
        t0 = time.perf_counter()
        n_samples_total = 0
        stat = None
        for s in range(n_warm, len(views)):
            stat, ns = run_step(views[s], timed=True)
            n_samples_total += ns
        barrier()
        elapsed = time.perf_counter() - t0
        H.kernel_timer_enable(False)
        timings = H.kernel_timer_collect()
        t_max = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return float(t_max.item()), n_samples_total, stat, timings

    views = make_views(0, args.warmup + args.steps, args.pixel_tiles)
    bin_counters = None
    if args.debug_bin_stats and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bin_stats
        bin_counters = bin_stats.enable(pkg.capi.lib().cdll, dev)
    elapsed, n_samples_total, stat, timings = timed_leg(views, args.warmup)
    if bin_counters is not None:
        bin_stats.report(pkg.capi.lib().cdll, bin_counters, L, out=sys.stderr)
    if dist is not None:
        # latency of the path's collective, measured after the timed region: blocking calls on the
        # same 16-byte payload, host-timed between synchronizes
        probe = torch.zeros(2, device=dev, dtype=torch.float64)
        us = []
        for _ in range(25):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            us.append((time.perf_counter() - t1) * 1e6)
        us = sorted(us[5:])
        collective["allreduce_us"] = us[len(us) // 2]
        collective["allreduce_us_how"] = ("median of 20 blocking all_reduce(SUM) calls on the 16-byte payload after "
                                          "the timed region, host-timed between synchronizes; inside the timed region "
                                          "the per-step reduce is issued asynchronously and waited for one step later")
    first_chunk = tuple(t[:args.chunk] for t in views[-1][:2])
    del views

    # ---- extras: separately timed, never part of value ---------------------------------------------
    extras = {}
    want_extras = not args.no_extras and args.workload in ("render", "c2") and args.rays == 0
    if want_extras and tiles_active(args, 8) and args.pixel_tiles != 8:
        # the same views handed over in 8x8 pixel tiles (the Renderer's own traversal in render_image)
        n_w, n_t = min(args.warmup, 2), min(args.steps, 5)
        tv = make_views(0, n_w + n_t, 8)
        t_el, _, _, t_tim = timed_leg(tv, n_w)
        del tv
        extras["pixel_tiles_8x8"] = {
            "value": n_rays_view * n_t * world / t_el, "unit": "rays/s", "steps": n_t, "warmup": n_w,
            "ms_per_step": t_el / n_t * 1e3,
            "hash_fwd_avg_ms": (t_tim["hash_fwd"][1] / max(t_tim["hash_fwd"][0], 1)) if "hash_fwd" in t_tim else None,
            "note": "same rays, same chunks, handed over in 8x8 pixel tiles instead of rows (round 2's "
                    "headline order); not part of value"}
    nonzero = None
    if want_extras and rank == 0:
        nonzero = time_hash_bwd_nonzero(args, pkg, H, ren, first_chunk, dev)

    # optimiser step: outside the headline (BASELINE.md excludes it) but reported next to it
    opt_ms = {}
    for name in ("make_fused_adam", "make_adam"):
        opt = getattr(ren, name)(1e-2)
        opt.step()                       # first step allocates the moment buffers
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            opt.step()
        torch.cuda.synchronize()
        opt_ms[name] = (time.perf_counter() - t1) / 5 * 1e3
        del opt

    render_img = None
    if args.render_images > 0:
        with torch.no_grad():
            ren.render_image(poses[0], intr, args.height, args.width, args.chunk)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(args.render_images):
                ren.render_image(poses[(k + 1) % args.n_images], intr, args.height, args.width, args.chunk)
            torch.cuda.synchronize()
        ms_img = (time.perf_counter() - t1) / args.render_images * 1e3
        render_img = {"ms_per_image": ms_img, "rays_per_s": args.height * args.width / ms_img * 1e3,
                      "images": args.render_images,
                      "note": "%dx%d, %d samples/ray, VALIDATE (no jitter), forward only, one GPU; "
                              "not part of value" % (args.height, args.width, S)}
    graphed = None
    n_graph = args.graph_iters if args.graph_iters >= 0 else (200 if args.workload == "c4" else 0)
    if n_graph > 0 and rank == 0:
        graphed = time_graphed_batches(args, H, dev, poses, intr, n_graph)
    train_iter = None
    if args.train_iters > 0:
        train_iter = time_train_iterations(args, pkg, H, dev, dist if world > 1 else None, world, poses, intr)

    psnr, mse = pkg.sharding.psnr_from_stats(stat)   # reference train_manager.cpp:96
    c5 = None
    if want_extras and world == 1:
        # BASELINE config C5 (the HBM stress) once after the headline, so that the driver's record
        # carries it: the hash kernels alone, bounded CPU leg
        del ren, params
        torch.cuda.empty_cache()
        try:
            full = c5_measure(args, rank, world, None, dev, pkg, steps=2, warmup=1,
                              cpu_points=0 if args.no_cpu_baseline else 1 << 18)
            c5 = {k: full[k] for k in ("value", "unit", "ms_per_step", "roofline", "roofline_fwd",
                                       "cpu_baseline", "gpu_over_cpu", "config") if k in full}
        except Exception as e:  # an extra: never a reason to lose the line
            c5 = {"error": repr(e)}

    if rank == 0:
        rays_total = n_rays_view * args.steps * world
        kernels = {}
        for name, (launches, total_ms, units) in timings.items():
            kernels[name] = {
                "launches": launches, "avg_ms": total_ms / max(launches, 1),
                "units_per_launch": units / max(launches, 1),
                "share_of_step": total_ms * 1e-3 / (elapsed * 1.0),
            }
            if name.startswith("shade"):   # matrix-core / vector f32 work: priced in flops
                fl = algorithmic_flops(name, L, F)
                tf = (units * fl) / (total_ms * 1e-3) / 1e12 if total_ms > 0 else None
                kernels[name].update({"algorithmic_flops_per_unit": fl, "achieved_TFLOPs": tf,
                                      "f32_peak_TFLOPs": F32_PEAK_TFLOPS,
                                      "frac_of_f32_peak": tf / F32_PEAK_TFLOPS if tf else None})
            else:
                b = algorithmic_bytes(name, L, F, S)
                kernels[name].update({
                    "algorithmic_bytes_per_unit": b,
                    "achieved_GBs": (units * b) / (total_ms * 1e-3) / 1e9 if total_ms > 0 else None})

        def roofline_of(op):
            if op not in kernels or not kernels[op].get("achieved_GBs"):
                return None
            a = kernels[op]["achieved_GBs"]
            traffic, traffic_src = pmc_traffic(op, workload_key(args))
            return {"kernel": op, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": kernels[op]["units_per_launch"] *
                    kernels[op]["algorithmic_bytes_per_unit"],
                    "avg_launch_ms": kernels[op]["avg_ms"]}

        hbm = [k for k in kernels if "achieved_GBs" in kernels[k]]
        dom = max(hbm, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"]) if hbm else None
        if nonzero is not None:
            kernels["hash_bwd_nonzero"] = nonzero
        tiles_on = tiles_active(args)
        if args.workload == "c3":
            what = ("C3: free-trajectory views %dx%d (120-pose 3-turn expanding spiral, look-ahead "
                    "orientation, fx=fy=%g), " % (args.height, args.width, args.focal))
        elif args.rays > 0:
            what = ("%s%d random rays per step per GPU, " %
                    ("C4 per-GPU shape (the reference's own batch): " if args.rays == 512 and S == 1024 else "",
                     args.rays))
        else:
            what = "ngp_fox-like synthetic views %dx%d, " % (args.height, args.width)
        out = {
            "metric": "rendered rays/sec (fwd+bwd) at 800x800",
            "value": rays_total / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": what + "%d samples/ray, L=%d F=%d T=2^%d, %d-ray chunks, TRAIN render + "
                                          "loss + backward (optimizer excluded)"
                                   % (S, L, F, args.log2_table, args.chunk),
                       "pixel_order": ("%dx%d tiles" % (args.pixel_tiles, args.pixel_tiles) if tiles_on
                                       else ("rows" if args.rays == 0 else "random")),
                       "regime": args.regime, "rays_per_step_per_gpu": n_rays_view,
                       "samples_per_ray_kept": n_samples_total / (n_rays_view * args.steps),
                       "sharding": "one view per rank per step, RCCL all-reduce of {sq_err, n} only"},
            "kernel_options": args.option,
            "psnr_vs_random_gt": psnr,
            "collective": collective,
            "optimizer_step_ms": {"fused_adam_with_f16_shadow": opt_ms["make_fused_adam"],
                                  "torch_optim_adam": opt_ms["make_adam"],
                                  "note": "per call, not part of value (BASELINE.md section 2)"},
            "roofline": roofline_of(dom) if dom else None,
            "hash_fwd_roofline": roofline_of("hash_fwd"),
            "kernels": kernels,
        }
        out.update(extras)
        if c5 is not None:
            out["c5"] = c5
        if graphed is not None:
            out["graphed_batch"] = graphed
        if train_iter is not None:
            out["train_iteration"] = train_iter
        if render_img is not None:
            out["render_image"] = render_img
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.cpu_rays)
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the line
                out["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
