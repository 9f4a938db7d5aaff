"""Ray sharding across the GPUs of one node and the single collective of the hot path.

Rays are independent through the whole render, so every rank renders its own rays
with replicated parameters and no data-path collective.  The only exchange is one all-reduce (SUM)
per step of {sum of squared colour error, number of values}, from which every rank derives the
global MSE / PSNR (PSNR as in the reference: 20*log10(1/sqrt(mse)), train_manager.cpp:96).
With the "nccl" backend this is RCCL over xGMI: a 16-byte payload, latency-bound.
"""
import math

import torch


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items for `rank`; sizes differ by at most one, none empty
    unless n_items < world."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def view_for(step, rank, world, n_images):
    """Weak-scaling schedule of bench.py: at every step each rank renders a different view."""
    return (step * world + rank) % n_images


def reduce_error_stats(sq_err_sum, n_values, dist=None, group=None):
    """all-reduce(SUM) of [sum sq err, n] -> (global sum, global n) as a float64 tensor [2]."""
    dev = sq_err_sum.device if torch.is_tensor(sq_err_sum) else "cpu"
    stat = torch.stack([torch.as_tensor(sq_err_sum, dtype=torch.float64, device=dev).reshape(()),
                        torch.tensor(float(n_values), dtype=torch.float64, device=dev)])
    if dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stat, group=group)
    return stat


def psnr_from_stats(stat):
    mse = float(stat[0] / stat[1])
    return 20.0 * math.log10(1.0 / math.sqrt(mse)), mse
