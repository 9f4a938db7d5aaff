"""Ray sharding across the GPUs of one node and the single collective of the hot path.

Rays are independent through the whole render, so every rank renders its own rays
with replicated parameters and no data-path collective.  The only exchange is one all-reduce (SUM)
per step of {sum of squared colour error, number of values}, from which every rank derives the
global MSE / PSNR (PSNR as in the reference: 20*log10(1/sqrt(mse)), train_manager.cpp:96).
With the "nccl" backend this is RCCL over xGMI: a 16-byte payload, latency-bound.

Data-parallel TRAINING (beyond the north star's render metric) additionally
averages the parameter gradients: `allreduce_gradients`.
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


def reduce_error_stats(sq_err_sum, n_values, dist=None, group=None, async_op=False):
    """all-reduce(SUM) of [sum sq err, n] -> (global sum, global n) as a float64 tensor [2].
    async_op: return (tensor, work) without making the caller's stream wait for the collective; the
    tensor holds the global sums once work.wait() has returned (work is None without a group)."""
    dev = sq_err_sum.device if torch.is_tensor(sq_err_sum) else "cpu"
    # (torch.full, not torch.tensor: a host scalar copied to the device is a blocking transfer that
    # waits for everything enqueued before it -- once per step it costs a 1 ms training batch 15 %)
    stat = torch.stack([torch.as_tensor(sq_err_sum, dtype=torch.float64, device=dev).reshape(()),
                        torch.full((), float(n_values), dtype=torch.float64, device=dev)])
    # (also with a single rank: a one-rank group still runs the backend's collective, which is how
    # bench.py --gpus 1 exercises RCCL on a one-GPU box)
    work = None
    if dist is not None and dist.is_initialized():
        work = dist.all_reduce(stat, group=group, async_op=async_op)
    return (stat, work) if async_op else stat


def psnr_from_stats(stat):
    mse = float(stat[0] / stat[1])
    return 20.0 * math.log10(1.0 / math.sqrt(mse)), mse


def allreduce_gradients(grads, dist=None, group=None, small_bucket_bytes=1 << 20):
    """Average the parameter gradients over the ranks, in place (data-parallel training: every rank
    rendered its own rays with the same parameters; the loss is a mean over rays, so the global
    gradient is the mean of the local ones when the ranks hold equally many rays).

    grads: iterable of tensors (undefined / None entries are skipped).  Sized for xGMI rings, which are
    per-link bound: the hash table's gradient (64 MiB at the reference size) is reduced IN PLACE as
    one collective -- no flatten copy of the one tensor that matters -- and everything below
    `small_bucket_bytes` (the MLPs, the embedding: ~20 KiB) rides in ONE flattened bucket, so a step
    costs two collectives instead of one per parameter.  Returns the number of collectives issued."""
    grads = [g for g in grads if g is not None and torch.is_tensor(g)]
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1 or not grads:
        return 0
    world = dist.get_world_size(group)
    big = [g for g in grads if g.numel() * g.element_size() >= small_bucket_bytes]
    small = [g for g in grads if g.numel() * g.element_size() < small_bucket_bytes]
    n_coll = 0
    for g in big:
        t = g if g.is_contiguous() else g.contiguous()
        dist.all_reduce(t, group=group)
        t.div_(world)
        if t is not g:
            g.copy_(t)
        n_coll += 1
    by_type = {}
    for g in small:
        by_type.setdefault((g.dtype, g.device), []).append(g)
    for gs in by_type.values():
        flat = torch.cat([g.reshape(-1) for g in gs])
        dist.all_reduce(flat, group=group)
        flat.div_(world)
        off = 0
        for g in gs:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_coll += 1
    return n_coll
