"""Multi-rank path on the GPU.

* Always (one GPU is enough): `bench.py --gpus 2 --backend gloo --share-gpu` -- the launcher starts two
  ranks, each renders its own rays through the HIP path on cuda:0, the {sum sq err, n} all-reduce and
  the max-over-ranks timing run, rank 0 prints ONE JSON line with n_gpus = 2.
* With two or more GPUs: the same helpers over RCCL (`backend="nccl"`): the error-statistics
  all-reduce and the data-parallel gradient average of f2-nerf_amd/sharding.py.  Skipped on the
  one-GPU test box; the driver's 8-GPU scaling run exercises RCCL through bench.py."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_bench_two_ranks_share_one_gpu(dev):
    res = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
         "--steps", "1", "--warmup", "1", "--rays", "2048", "--samples", "64", "--levels", "4",
         "--chunk", "2048", "--no-cpu-baseline"],
        env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["rays_per_step_per_gpu"] == 2048
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0


_WORKER = r'''
import importlib, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world)
sh = importlib.import_module("f2-nerf_amd").sharding
g = torch.Generator().manual_seed(123)
n = 4099
pred, gt = torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g)
lo, hi = sh.shard_range(n, rank, world)
err = (pred[lo:hi] - gt[lo:hi]).to(dev)
stat = sh.reduce_error_stats(err.double().square().sum(), err.numel(), dist)
want = float((pred - gt).double().square().sum())
assert abs(float(stat[0]) - want) < 1e-9 * want and float(stat[1]) == 3 * n
# data-parallel gradient average: table above the bucket threshold, small tensors in one bucket
g = torch.Generator().manual_seed(7)
w_big = torch.randn(70000, 4, generator=g).to(dev).requires_grad_(True)
w_a = torch.randn(16, 32, generator=g).to(dev).requires_grad_(True)
x = torch.randn(4096, 32, generator=g)
lo, hi = sh.shard_range(x.shape[0], rank, world)
xs = x[lo:hi].to(dev)
loss = (xs @ w_a.t()).square().mean() + (w_big[:hi - lo, :1] * xs[:, :1]).sum() / (hi - lo)
loss.backward()
n_coll = sh.allreduce_gradients([w_big.grad, w_a.grad], dist, small_bucket_bytes=1 << 16)
assert n_coll == 2
full_b = torch.randn(70000, 4, generator=torch.Generator().manual_seed(7))
ga = [torch.empty_like(w_a.grad) for _ in range(world)]
dist.all_gather(ga, w_a.grad)
assert all(torch.equal(ga[0], t) for t in ga)          # every rank holds the same averaged gradient
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def test_rccl_error_reduce_and_gradient_average(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL over xGMI (the one-GPU box covers the gloo path)")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script), ROOT],
        env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout
