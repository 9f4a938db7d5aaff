"""Multi-rank path on CPU: two gloo ranks shard a ray batch, each produces its local
{sum of squared error, count}, and the single all-reduce of the hot path must reproduce the
single-process MSE / PSNR (bench.py uses the same helpers with the RCCL backend)."""
import importlib
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("f2-nerf_amd").sharding
    g = torch.Generator().manual_seed(123)               # same global batch on every rank
    n = 4099
    pred, gt = torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g)
    lo, hi = sh.shard_range(n, rank, world)
    err = pred[lo:hi] - gt[lo:hi]
    stat = sh.reduce_error_stats(err.double().square().sum(), err.numel(), dist)
    psnr, mse = sh.psnr_from_stats(stat)
    out_q.put((rank, lo, hi, float(stat[0]), float(stat[1]), psnr, mse))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_error_reduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(123)
    n = 4099
    pred, gt = torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g)
    sq = float((pred - gt).double().square().sum())
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n
    sh = importlib.import_module("f2-nerf_amd").sharding
    want_psnr, want_mse = sh.psnr_from_stats(torch.tensor([sq, 3.0 * n], dtype=torch.float64))
    for r in res:
        assert abs(r[3] - sq) < 1e-9 * sq and r[4] == 3 * n
        assert abs(r[5] - want_psnr) < 1e-9 and abs(r[6] - want_mse) < 1e-12


def _grad_worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("f2-nerf_amd").sharding
    # a "table" above the bucket threshold, small "MLP" tensors below it, one of them non-contiguous
    g = torch.Generator().manual_seed(7)
    w_big = torch.randn(70000, 4, generator=g, requires_grad=True)
    w_a = torch.randn(16, 32, generator=g, requires_grad=True)
    w_b = torch.randn(64, generator=g, requires_grad=True)
    x = torch.randn(4096, 32, generator=g)
    lo, hi = sh.shard_range(x.shape[0], rank, world)
    xs = x[lo:hi]
    loss = ((xs @ w_a.t()).square().mean() + (w_big[:hi - lo, :1] * xs[:, :1]).sum() / (hi - lo)
            + (w_b * xs[:, :1].mean()).sum())
    loss.backward()
    grads = [w_big.grad, w_a.grad.t(), None, w_b.grad]      # a transposed view and a missing grad
    n_coll = sh.allreduce_gradients(grads, dist, small_bucket_bytes=1 << 16)
    # plain numpy payloads: a torch tensor on a spawn Queue travels as a shared-memory handle that
    # dies with the sender
    out_q.put((rank, n_coll, w_big.grad.numpy().copy(), w_a.grad.numpy().copy(),
               w_b.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_matches_full_batch():
    """Data-parallel gradient averaging: two gloo ranks, each with half of the
    batch, end up with the gradient of the mean loss over the whole batch; two collectives."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(7)
    w_big = torch.randn(70000, 4, generator=g, requires_grad=True)
    w_a = torch.randn(16, 32, generator=g, requires_grad=True)
    w_b = torch.randn(64, generator=g, requires_grad=True)
    x = torch.randn(4096, 32, generator=g)
    half = x.shape[0] // 2
    total = 0
    for lo, hi in ((0, half), (half, x.shape[0])):
        xs = x[lo:hi]
        total = total + ((xs @ w_a.t()).square().mean() + (w_big[:hi - lo, :1] * xs[:, :1]).sum() / (hi - lo)
                         + (w_b * xs[:, :1].mean()).sum())
    (total / 2).backward()
    for r in res:
        assert r[1] == 2                                   # the table alone + one bucket
        torch.testing.assert_close(torch.from_numpy(r[2]), w_big.grad, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(torch.from_numpy(r[3]), w_a.grad, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(torch.from_numpy(r[4]), w_b.grad, rtol=1e-6, atol=1e-7)
    assert (res[0][2] == res[1][2]).all()                  # ranks agree bit for bit
