"""Parity of every C-ABI kernel (libf2nerf_hip.so, called through ctypes) against the CPU oracle on
identical seeded inputs.  Integer / index results must be bit-exact; float tolerances are written
at each assert (north star: 1e-4 relative, hash rows bit-exact)."""
import pytest
import torch

from oracle import kernels as K
from tests import util

pytestmark = pytest.mark.gpu


def _to(dev, *ts):
    return [t.to(dev) if t is not None else None for t in ts]


# --------------------------------------------------------------------------------- hash grid ----


def _assert_hash_values(got, ref):
    """Same FMA order and the same f32-then-f16 rounding on both sides: bit-exact is expected
    (the kernel pins the f32 rounding so hipcc cannot fuse it into v_fma_mixlo_f16)."""
    assert torch.equal(got, ref), ((got != ref).sum().item(), ref.numel())


@pytest.mark.parametrize("L,F,log2_T,stride_mode", [
    (16, 2, 19, "ref"),      # the reference's compile-time configuration (overlapping levels, Q2)
    (4, 2, 19, "ref"),       # BASELINE config C1
    (16, 8, 14, "disjoint"), # config C5's shape (F=8, non-overlapping stride) at a small T
    (3, 4, 10, "disjoint"),
    (5, 1, 12, "ref"),
])
def test_hash_fwd_parity(capi, dev, L, F, log2_T, stride_mode):
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, None if stride_mode == "ref" else T * F, seed=L * 31 + F)
    n = 20000
    pts = util.ball_points(n, seed=5)
    pts[:4] = torch.tensor([[0., 0., 0.], [-2., -2., -2.], [1.9999, 0., 0.], [-1.5, 1.2, -0.3]])
    ref, ref_idx = K.hash_fwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], L, F, T,
                              fld["stride"], want_idx=True)
    d_pts, d_tab, d_pr, d_bias, d_mul = _to(dev, pts, fld["table16"], fld["primes"], fld["bias"],
                                            fld["mul"])
    out = torch.empty(n, L * F, device=dev)
    idx = torch.empty(n, L, 8, dtype=torch.int32, device=dev)
    capi.call("hash_fwd", d_pts, d_tab, d_pr, d_bias, d_mul, out, L * F, 1, idx, n, L, F, T,
              fld["stride"])
    # hash rows: integer work, bit-exact
    assert torch.equal(idx.cpu(), ref_idx)
    got = out.cpu()
    _assert_hash_values(got, ref)
    # channel-major output layout gives the same numbers
    out_t = torch.empty(L * F, n, device=dev)
    capi.call("hash_fwd", d_pts, d_tab, d_pr, d_bias, d_mul, out_t, 1, n, None, n, L, F, T,
              fld["stride"])
    assert torch.equal(out_t.t().contiguous().cpu(), got)


@pytest.mark.parametrize("L,F,T,n_rays,S", [
    (16, 2, 1 << 19, 333, 128),   # reference configuration, ragged last ray tile
    (4, 8, 1 << 12, 64, 16),      # exactly one tile
    (3, 1, 1000, 130, 48),        # T not a power of two
    (2, 4, 1 << 10, 1, 32),       # a single ray
])
def test_hash_fwd_raytile_matches_oracle(capi, dev, L, F, T, n_rays, S):
    """The ray-tile mapping (lanes = neighbouring rays at one depth) changes which thread computes a
    sample, not what is computed: bit-exact against the oracle, like f2n_hash_fwd."""
    import math
    fld = util.make_field(L, F, max(1, math.ceil(math.log2(T))), T * F, seed=L + F)
    n = n_rays * S
    pts = util.ball_points(n, seed=11)
    ref, _ = K.hash_fwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], L, F, T,
                        fld["stride"], want_idx=True)
    d = _to(dev, pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"])
    # every way a tile can be walked (chosen per tile / across at one sample index / along a ray /
    # across in depth order) computes the same values
    for walk in (0, 1, 2, 3):
        with capi.option("RAYTILE_WALK", walk):
            out = torch.full((L * F, n), 7.0, device=dev)
            capi.call("hash_fwd_raytile", *d, out, n_rays, S, L, F, T, fld["stride"])
        _assert_hash_values(out.t().contiguous().cpu(), ref)
    with pytest.raises(capi.F2NError):   # S must be a multiple of 16
        capi.call("hash_fwd_raytile", *d, out, n_rays, S - 1, L, F, T, fld["stride"])


def test_hash_fwd_non_pow2_T_and_empty(capi, dev):
    L, F, T = 4, 2, 1000
    fld = util.make_field(L, F, 10, T * F, seed=3)
    n = 5000
    pts = util.ball_points(n, seed=9)
    ref, ref_idx = K.hash_fwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], L, F, T,
                              T * F, want_idx=True)
    d = _to(dev, pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"])
    out = torch.empty(n, L * F, device=dev)
    idx = torch.empty(n, L, 8, dtype=torch.int32, device=dev)
    capi.call("hash_fwd", *d, out, L * F, 1, idx, n, L, F, T, T * F)
    assert torch.equal(idx.cpu(), ref_idx)
    _assert_hash_values(out.cpu(), ref)
    capi.call("hash_fwd", *d, out, L * F, 1, None, 0, L, F, T, T * F)  # n == 0 is a no-op
    with pytest.raises(capi.F2NError):
        capi.call("hash_fwd", *d, out, L * F, 1, None, n, L, 3, T, T * F)  # F=3 unsupported


@pytest.mark.parametrize("L,F,log2_T,stride_mode,pts_grad", [
    (16, 2, 19, "ref", False),
    (16, 2, 12, "ref", True),
    (4, 8, 10, "disjoint", True),
    (6, 4, 11, "disjoint", False),
])
def test_hash_bwd_parity(capi, dev, L, F, log2_T, stride_mode, pts_grad):
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, None if stride_mode == "ref" else T * F, seed=11 + F)
    n = 8000
    pts = util.ball_points(n, seed=6)
    g = torch.Generator().manual_seed(4)
    grad = torch.randn(n, L * F, generator=g) * 1e-3
    grad[torch.rand(n, L * F, generator=g) < 0.2] = 0.0
    numel = fld["table"].numel()
    ref_tg, ref_pg = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                                numel, L, F, T, fld["stride"], 128.0, need_pts_grad=pts_grad)
    d = _to(dev, pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad)
    tg = torch.zeros(numel, device=dev)
    pg = torch.full((n, 3), 7.0, device=dev) if pts_grad else None
    capi.call("hash_bwd", *d, L * F, 1, tg, pg, n, L, F, T, fld["stride"], 128.0)
    # each contribution is f16-rounded identically; only the f32 summation order differs
    scale = ref_tg.abs().max().item()
    assert (tg.cpu() - ref_tg).abs().max().item() <= 1e-5 * scale
    assert torch.equal(tg.cpu() != 0, ref_tg != 0) or \
        ((tg.cpu() != 0) ^ (ref_tg != 0)).float().mean().item() < 1e-6
    if pts_grad:
        s = ref_pg.abs().max().item()
        assert (pg.cpu() - ref_pg).abs().max().item() <= 1e-5 * s + 1e-12
    # accumulate semantics: a second call doubles the table gradient
    capi.call("hash_bwd", *d, L * F, 1, tg, None, n, L, F, T, fld["stride"], 128.0)
    assert (tg.cpu() - 2 * ref_tg).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("L,F,T,stride,n", [
    (16, 2, 1 << 19, None, 40000),     # reference size: 32 slices per level, overlapping levels
    (4, 2, 1 << 19, None, 70000),      # few (level, slice) pairs -> sample partitions
    (3, 4, 3000, 3000 * 4, 33000),     # non power-of-two T, one partial slice
    (2, 8, 1 << 13, (1 << 13) * 8, 33000),
])
def test_hash_bwd_sliced_path(capi, dev, L, F, T, stride, n):
    """Big batches without a point gradient take the LDS-sliced kernel; same contributions, the
    f32 sums in another order."""
    log2_T = max(1, (T - 1).bit_length())
    fld = util.make_field(L, F, log2_T, stride, seed=77)
    st = fld["stride"]
    pts = util.ball_points(n, seed=16)
    g = torch.Generator().manual_seed(14)
    grad = torch.randn(n, L * F, generator=g) * 1e-3
    numel = fld["table"].numel()
    ref_tg, _ = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                           numel, L, F, T, st, 128.0, parallel=True)
    d = _to(dev, pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad)
    tg = torch.zeros(numel, device=dev)
    capi.call("hash_bwd", *d, L * F, 1, tg, None, n, L, F, T, st, 128.0)
    scale = ref_tg.abs().max().item()
    assert (tg.cpu() - ref_tg).abs().max().item() <= 2e-5 * scale
    assert ((tg.cpu() - ref_tg).norm() / ref_tg.norm()).item() < 1e-6


@pytest.mark.parametrize("L,F,log2_T,stride_mode,n,ws_frac", [
    (16, 2, 19, "ref", 70000, 1.0),     # reference size, recommended workspace
    (16, 2, 19, "ref", 70000, 0.3),     # small workspace: region overflow -> direct atomics
    (4, 4, 17, "disjoint", 66000, 1.0),
    (3, 1, 20, "disjoint", 66000, 1.0),
    # more than 64 slices per level: two-level binning (bucket = 2^k slices, split pass, run reduce)
    (3, 8, 18, "disjoint", 70000, 1.0),     # 128 slices = 64 buckets x 2, 2 point rounds per tile
    (2, 8, 20, "disjoint", 70000, 1.0),     # 512 slices = 64 x 8
    (2, 2, 21, "disjoint", 70000, 1.0),     # F = 2 records through the split pass: 256 slices
    (2, 4, 19, "ref", 70000, 0.4),          # 128 slices, overlapping level windows, several rounds
])
def test_hash_bwd_binned_path(capi, dev, L, F, log2_T, stride_mode, n, ws_frac):
    """Binned backward (bin into workspace + LDS reduce): same contributions as the atomic kernel.
    A workspace smaller than recommended makes the passes run in several rounds over the points."""
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, None if stride_mode == "ref" else T * F, seed=5 + F)
    st = fld["stride"]
    pts = util.ball_points(n, seed=21)
    pts[: n // 2] *= 0.02          # half the points in a tiny ball: coarse levels hit few rows (skew)
    g = torch.Generator().manual_seed(15)
    grad = torch.randn(n, L * F, generator=g) * 1e-3
    grad[torch.rand(n, L * F, generator=g) < 0.1] = 0.0
    numel = fld["table"].numel()
    ref_tg, _ = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                           numel, L, F, T, st, 128.0, parallel=True)
    need = capi.lib().cdll.f2n_hash_bwd_workspace_bytes(n, L, F, T)
    assert need > 0
    nbytes = int(need * ws_frac) // 256 * 256
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d = _to(dev, pts, fld["primes"], fld["bias"], fld["mul"], grad)
    tg = torch.zeros(numel, device=dev)
    capi.call("hash_bwd_binned", *d, L * F, 1, tg, n, L, F, T, st, 128.0, ws, nbytes)
    scale = ref_tg.abs().max().item()
    assert (tg.cpu() - ref_tg).abs().max().item() <= 2e-5 * scale
    assert ((tg.cpu() - ref_tg).norm() / ref_tg.norm()).item() < 1e-5
    # channel-major gradients give the same result
    tg2 = torch.zeros(numel, device=dev)
    capi.call("hash_bwd_binned", d[0], d[1], d[2], d[3], d[4].t().contiguous(), 1, n, tg2, n, L, F,
              T, st, 128.0, ws, nbytes)
    assert (tg2.cpu() - ref_tg).abs().max().item() <= 2e-5 * scale
    assert capi.lib().cdll.f2n_hash_bwd_workspace_bytes(100, L, F, T) == 0   # small n: not applicable


def _ray_points(n_rays, S, seed):
    """Contracted samples of random rays, ray-major (what the renderer feeds the backward): consecutive
    points share coarse cells, which is what the per-tile combine of the binned backward feeds on."""
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n_rays, 1, 3, generator=g) * 0.3
    d = torch.randn(n_rays, 1, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    t = (torch.arange(1, S + 1).float() * (4.0 / S)).reshape(1, S, 1)
    p = (o + d * t).reshape(-1, 3)
    nrm = p.norm(dim=1, keepdim=True)
    return torch.where(nrm <= 1, p, (2 - 1 / nrm) * p / nrm).contiguous()


@pytest.mark.parametrize("L,F,log2_T,ws_frac,must_overflow", [
    (3, 8, 20, 1.0, True), (3, 8, 18, 0.5, False), (3, 2, 21, 1.0, True)])
def test_hash_bwd_binned_overflow_stays_exact(capi, dev, L, F, log2_T, ws_frac, must_overflow):
    """Two-level tables (more than 64 slices per level).  Half of the points sit in a ball of radius
    0.7: at the coarsest level they touch ~1500 distinct rows, so some slices of each bucket
    receive many times the mean number of records and the split pass overflows its sub-slice queues
    and per-slice runs (the skew config C5 shows at its levels 0 and 1, stronger).  Those records go
    to the overflow arena of their (level, bucket) and are summed exactly like the rest: the overflow
    counter stays 0 and two launches agree BIT FOR BIT (a float-atomic fallback would make the last
    bits order-dependent)."""
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, T * F, seed=9 + F)
    st = fld["stride"]
    n = 200000
    pts = util.ball_points(n, seed=41)
    g = torch.Generator().manual_seed(42)
    perm = torch.randperm(n, generator=g)
    pts[perm[: n // 2]] *= 0.35         # scattered over all tiles: no tile piles onto one cell
    grad = torch.randn(n, L * F, generator=g) * 1e-3           # every contribution non-zero
    numel = fld["table"].numel()
    ref_tg, _ = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                           numel, L, F, T, st, 128.0, parallel=True)
    need = capi.lib().cdll.f2n_hash_bwd_workspace_bytes(n, L, F, T)
    nbytes = int(need * ws_frac) // 256 * 256
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d = _to(dev, pts, fld["primes"], fld["bias"], fld["mul"], grad)
    overflow = torch.zeros(1, dtype=torch.int64, device=dev)
    cd = capi.lib().cdll
    assert cd.f2n_hash_bwd_set_overflow_counter(overflow.data_ptr()) == 0
    stats = torch.zeros(64, 4, dtype=torch.int32, device=dev)     # split-pass counters per level
    import ctypes
    cd.f2n_debug_bin_stats.argtypes = [ctypes.c_void_p]
    cd.f2n_debug_bin_stats.restype = None
    cd.f2n_debug_bin_stats(stats.data_ptr())
    outs = []
    try:
        for _ in range(2):
            tg = torch.zeros(numel, device=dev)
            capi.call("hash_bwd_binned", *d, L * F, 1, tg, n, L, F, T, st, 128.0, ws, nbytes)
            outs.append(tg)
        torch.cuda.synchronize()
    finally:
        cd.f2n_hash_bwd_set_overflow_counter(None)
        cd.f2n_debug_bin_stats(None)
    if must_overflow:                          # records DID run past a queue or a run of pass B ...
        assert int(stats[:, 2:4].sum()) > 0
    assert int(overflow.item()) == 0           # ... and none of them became a float atomic
    assert torch.equal(outs[0], outs[1])
    scale = ref_tg.abs().max().item()
    assert (outs[0].cpu() - ref_tg).abs().max().item() <= 2e-5 * scale
    assert cd.f2n_hash_bwd_set_overflow_counter(3) != 0            # misaligned pointer: rejected
    # another record order (the per-tile combine off), same exact sums
    with capi.option("BWD_COMBINE", 1):
        tg = torch.zeros(numel, device=dev)
        capi.call("hash_bwd_binned", *d, L * F, 1, tg, n, L, F, T, st, 128.0, ws, nbytes)
    assert torch.equal(tg, outs[0])


@pytest.mark.parametrize("L,F,log2_T,S", [(16, 2, 19, 128), (8, 4, 16, 1024), (6, 1, 19, 256)])
def test_hash_bwd_binned_combine(capi, dev, L, F, log2_T, S):
    """Ray-coherent points: coarse levels are combined per tile (sums leave as f16 pieces), fine
    levels are not; with the combine switched off the very same table gradient must come out bit
    for bit wherever no capacity overflowed (both are exact sums), and both match the oracle."""
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, None, seed=3 + F)
    n_rays = 66560 // S + 1
    pts = _ray_points(n_rays, S, seed=31)
    n = pts.shape[0]
    g = torch.Generator().manual_seed(32)
    grad = torch.randn(n, L * F, generator=g) * 1e-3 * torch.rand(n, 1, generator=g) ** 4
    grad[torch.rand(n, L * F, generator=g) < 0.05] = 0.0
    numel = fld["table"].numel()
    ref_tg, _ = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                           numel, L, F, T, fld["stride"], 128.0, parallel=True)
    need = capi.lib().cdll.f2n_hash_bwd_workspace_bytes(n, L, F, T)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    d = _to(dev, pts, fld["primes"], fld["bias"], fld["mul"], grad.t().contiguous())
    out = {}
    for combine_off in (0, 1):
        with capi.option("BWD_COMBINE", combine_off):
            tg = torch.zeros(numel, device=dev)
            capi.call("hash_bwd_binned", *d, 1, n, tg, n, L, F, T, fld["stride"], 128.0, ws, need)
            out[combine_off] = tg.cpu()
    scale = ref_tg.abs().max().item()
    for k, tg in out.items():
        assert (tg - ref_tg).abs().max().item() <= 2e-5 * scale, k
        assert ((tg - ref_tg).norm() / ref_tg.norm()).item() < 1e-5, k
    # exact sums either way: only entries touched by an overflow fallback (float atomics) may differ
    assert (out[0] != out[1]).float().mean().item() < 0.02
    assert (out[0] - out[1]).abs().max().item() <= 1e-6 * scale


@pytest.mark.parametrize("grad_std", [1e-2, 2e-6])
def test_hash_bwd_binned_hot_spot(capi, dev, grad_std):
    """Rays that stop right in front of one camera: three or four samples each, all in the same few
    cells, every gradient non-zero -- a tile's 8192 contributions land on a dozen rows per level.
    Plain binning overflows its queues there (the records leave as same-address atomics); the tile
    must notice (repeating cells / an overflowing queue), combine from then on, and stay exact: sums
    in the combined levels are bit-equal to the oracle's exact sums up to its own f32 rounding.
    Small gradients (2e-6: below the magnitude the repeating-cells test asks for, above the f16
    underflow) must be noticed BEFORE level 0 as well -- surviving contributions per distinct
    level-0 cell -- so that (next to) no record of the coarse levels ends as a float atomic: at most
    a few dozen of 4.5 million at L = 8, against a quarter of a million at level 0 alone before
    (round 3)."""
    L, F, log2_T = 16, 2, 19
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, None, seed=5)
    g = torch.Generator().manual_seed(8)
    n_rays = 20000
    o = torch.tensor([0.31, -0.22, 0.17]) + torch.zeros(n_rays, 1, 3)
    d = torch.randn(n_rays, 1, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    k = torch.randint(3, 5, (n_rays,), generator=g)                  # 3 or 4 samples per ray
    t = (torch.arange(1, 5).float() / 32.0).reshape(1, 4, 1)
    keep = (torch.arange(4).reshape(1, 4) < k.reshape(-1, 1)).reshape(-1)
    pts = (o + d * t).reshape(-1, 3)[keep].contiguous()
    n = pts.shape[0]
    assert n >= 65536
    grad = torch.randn(n, L * F, generator=g) * grad_std
    numel = fld["table"].numel()
    ref_tg, _ = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                           numel, L, F, T, fld["stride"], 128.0, parallel=True)
    need = capi.lib().cdll.f2n_hash_bwd_workspace_bytes(n, L, F, T)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    dd = _to(dev, pts, fld["primes"], fld["bias"], fld["mul"], grad.t().contiguous())
    scale = ref_tg.abs().max().item()
    cd = capi.lib().cdll
    for combine_off in (0, 1):
        overflow = torch.zeros(1, dtype=torch.int64, device=dev)
        assert cd.f2n_hash_bwd_set_overflow_counter(overflow.data_ptr()) == 0
        try:
            with capi.option("BWD_COMBINE", combine_off):
                tg = torch.zeros(numel, device=dev)
                capi.call("hash_bwd_binned", *dd, 1, n, tg, n, L, F, T, fld["stride"], 128.0, ws, need)
                after_all = int(overflow.item())
                overflow.zero_()
                # the coarse half of the levels alone: every tile must combine all of them
                tg8 = torch.zeros(numel, device=dev)
                capi.call("hash_bwd_binned", *dd, 1, n, tg8, n, 8, F, T, fld["stride"], 128.0, ws, need)
                coarse_only = int(overflow.item())
        finally:
            cd.f2n_hash_bwd_set_overflow_counter(None)
        if combine_off == 0:
            assert coarse_only <= 100 and after_all <= 200, (coarse_only, after_all)
        else:
            assert coarse_only > 100000     # (what the combine is there for)
        # (with the combine off, or before a tile has noticed, overflowing records are float atomics:
        # order-dependent in the last bits of sums of thousands of terms)
        assert (tg.cpu() - ref_tg).abs().max().item() <= 1e-4 * scale, combine_off
        assert ((tg.cpu() - ref_tg).norm() / ref_tg.norm()).item() < 1e-5, combine_off


@pytest.mark.parametrize("route", ["binned", "binned_nocombine", "sliced", "atomic"])
def test_hash_bwd_nonfinite_gradient_propagates(capi, dev, route):
    """An incoming gradient beyond the f16 range (128*g overflows to inf) must poison the rows it
    touches on every route, as the reference's f16 atomics would -- not vanish into a finite sum
    (f16_bits_to_fixed had no case for exponent 31)."""
    L, F, log2_T = 4, 2, 19
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, None, seed=9)
    pts = _ray_points(520, 128, seed=5)
    n = pts.shape[0]
    g = torch.Generator().manual_seed(6)
    grad = torch.randn(n, L * F, generator=g) * 1e-3
    bad_p, bad_c = 777, 3                      # level 1, channel 1
    grad[bad_p, bad_c] = 1.0e3                 # 128e3 > 65504 -> +inf as f16
    numel = fld["table"].numel()
    ref_tg, _ = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                           numel, L, F, T, fld["stride"], 128.0)
    assert not torch.isfinite(ref_tg).all()
    d = _to(dev, pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad)
    tg = torch.zeros(numel, device=dev)
    if route.startswith("binned"):
        need = capi.lib().cdll.f2n_hash_bwd_workspace_bytes(n, L, F, T)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        with capi.option("BWD_COMBINE", 1 if route.endswith("nocombine") else 0):
            capi.call("hash_bwd_binned", d[0], d[2], d[3], d[4], d[5], L * F, 1, tg, n, L, F, T,
                      fld["stride"], 128.0, ws, need)
    else:
        with capi.option("HASH_BWD", {"atomic": 1, "sliced": 2}[route]):
            capi.call("hash_bwd", *d, L * F, 1, tg, None, n, L, F, T, fld["stride"], 128.0)
    got = tg.cpu()
    assert torch.equal(torch.isfinite(got), torch.isfinite(ref_tg))
    fin = torch.isfinite(ref_tg)
    scale = ref_tg[fin].abs().max().item()
    assert (got[fin] - ref_tg[fin]).abs().max().item() <= 2e-5 * scale


def test_table_cast(capi, dev):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1 << 16, generator=g) * 0.1
    x[:6] = torch.tensor([65504., 65520., 1e-8, 6e-8, -0.0, 3.0e-5])
    x = torch.cat([x, torch.randn(13, generator=g)])  # tail not a multiple of 8
    out = torch.empty(x.numel(), dtype=torch.int16, device=dev)
    capi.call("table_to_f16", x.to(dev), out, x.numel())
    assert torch.equal(out.cpu(), K.cast_f16(x))


def test_contract(capi, dev):
    g = torch.Generator().manual_seed(2)
    p = torch.randn(10000, 3, generator=g) * 2.0
    p[0] = 0.0                      # quirk Q6: NaN
    p[1] = torch.tensor([1.0, 0.0, 0.0])  # |p| == 1 stays
    pr = p.clone().requires_grad_(True)
    norm = pr.norm(2, dim=1, keepdim=True)
    mask = norm <= 1.0
    x = pr * mask + ~mask * (1 + 1.0 - 1.0 / norm) * pr / norm
    out = torch.empty(p.shape, device=dev)
    capi.call("contract_fwd", p.to(dev), out, p.shape[0])
    got = out.cpu()
    assert torch.isnan(got[0]).all() and torch.isnan(x[0]).all()
    torch.testing.assert_close(got[1:], x[1:].detach(), rtol=2e-6, atol=1e-7)
    gx = torch.randn(p.shape, generator=g)
    x[1:].backward(gx[1:])
    dp = torch.empty(p.shape, device=dev)
    capi.call("contract_bwd", p.to(dev), gx.to(dev), dp, p.shape[0])
    torch.testing.assert_close(dp.cpu()[1:], pr.grad[1:], rtol=1e-4, atol=1e-6)


# --------------------------------------------------------------------------------- SH ----------


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
def test_sh_encode(capi, dev, degree):
    g = torch.Generator().manual_seed(7)
    d = torch.randn(30001, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    ref = K.sh_encode(d, degree)
    out = torch.empty(d.shape[0], degree * degree, device=dev)
    capi.call("sh_encode", d.to(dev), out, d.shape[0], degree)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("degree", [5, 6, 7, 8])
def test_sh_encode_high_degrees(capi, dev, degree):
    """Bands 4..7 (coded in the reference, src/sh_shader.cu:52-102, used by nothing in it): the
    kernel runs the f32 recurrences, the oracle the same definition in double -- tolerance 4e-6
    absolute on values of magnitude <= 3 (a few f32 roundings of an 8-term recurrence); the first 16
    columns stay bit-equal to the degree-4 result the renderer uses."""
    g = torch.Generator().manual_seed(7)
    d = torch.randn(30001, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    d[:6] = torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [-1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
    ref = K.sh_encode(d, degree)
    out = torch.empty(d.shape[0], degree * degree, device=dev)
    capi.call("sh_encode", d.to(dev), out, d.shape[0], degree)
    torch.testing.assert_close(out.cpu(), ref, rtol=0, atol=4e-6)
    out4 = torch.empty(d.shape[0], 16, device=dev)
    capi.call("sh_encode", d.to(dev), out4, d.shape[0], 4)
    assert torch.equal(out[:, :16], out4)
    assert capi.lib().cdll.f2n_sh_encode(d.to(dev).data_ptr(), out.data_ptr(), 10, 9, None) == -3


# --------------------------------------------------------------------------------- segments -----


@pytest.mark.parametrize("n_rays,max_len", [(1, 5), (37, 70), (512, 1024), (3000, 200)])
def test_segment_ops(capi, dev, n_rays, max_len):
    idx, n = util.ragged_bounds(n_rays, max_len, seed=n_rays)
    g = torch.Generator().manual_seed(3)
    val = torch.rand(max(n, 1), generator=g)[:n]
    d_idx, d_val = idx.to(dev), val.to(dev)
    # Sum
    out = torch.empty(n_rays, device=dev)
    capi.call("seg_sum_fwd", d_val, d_idx, out, n_rays)
    torch.testing.assert_close(out.cpu(), K.seg_sum_fwd(val, idx), rtol=1e-5, atol=1e-6)
    dsum = torch.randn(n_rays, generator=g)
    dval = torch.zeros(n, device=dev)
    capi.call("seg_sum_bwd", dsum.to(dev), d_idx, dval, n_rays)
    assert torch.equal(dval.cpu(), K.seg_sum_bwd(dsum, idx, n))
    # SumVec (3 channels = RGB, and 16)
    for vec in (3, 16):
        v = torch.rand(n, vec, generator=g)
        out = torch.empty(n_rays, vec, device=dev)
        capi.call("seg_sum_vec_fwd", v.to(dev), d_idx, out, n_rays, vec)
        torch.testing.assert_close(out.cpu(), K.seg_sum_fwd(v, idx), rtol=1e-5, atol=1e-6)
        ds = torch.randn(n_rays, vec, generator=g)
        dv = torch.zeros(n, vec, device=dev)
        capi.call("seg_sum_vec_bwd", ds.to(dev), d_idx, dv, n_rays, vec)
        assert torch.equal(dv.cpu(), K.seg_sum_bwd(ds, idx, n))
    # AccumulateSum, both flavours, fwd + bwd
    for inc in (0, 1):
        out = torch.zeros(n, device=dev)
        capi.call("seg_scan_fwd", d_val, d_idx, out, n_rays, inc)
        torch.testing.assert_close(out.cpu(), K.seg_scan_fwd(val, idx, inc), rtol=1e-5, atol=1e-6)
        gsum = torch.randn(n, generator=g)
        out = torch.zeros(n, device=dev)
        capi.call("seg_scan_bwd", gsum.to(dev), d_idx, out, n_rays, inc)
        # randn inputs cancel: the error scales with sum|x| over the run (<= ~1e3), not with the result
        torch.testing.assert_close(out.cpu(), K.seg_scan_bwd(gsum, idx, inc), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("n_rays,max_len", [(64, 40), (700, 300)])
def test_weight_var(capi, dev, n_rays, max_len):
    idx, n = util.ragged_bounds(n_rays, max_len, seed=21)
    g = torch.Generator().manual_seed(8)
    w = torch.rand(n, generator=g) * 0.1
    out = torch.empty(n_rays, device=dev)
    capi.call("weight_var_fwd", w.to(dev), idx.to(dev), out, n_rays)
    torch.testing.assert_close(out.cpu(), K.weight_var_fwd(w, idx), rtol=2e-4, atol=1e-6)
    dv = torch.randn(n_rays, generator=g)
    dw = torch.zeros(n, device=dev)
    capi.call("weight_var_bwd", w.to(dev), idx.to(dev), dv.to(dev), dw, n_rays)
    ref = K.weight_var_bwd(w, idx, dv)
    torch.testing.assert_close(dw.cpu(), ref, rtol=2e-4, atol=1e-4 * ref.abs().max().item())


# --------------------------------------------------------------------------------- scatter ------


def test_scatter(capi, dev):
    n_rays, E, C = 500, 50, 16
    idx, n = util.ragged_bounds(n_rays, 120, seed=2)
    g = torch.Generator().manual_seed(12)
    emb_idx = torch.randint(0, E, (n_rays,), generator=g).to(torch.int32)
    ref_all = K.scatter_idx(n, idx, emb_idx)
    all_idx = torch.zeros(n, dtype=torch.int32, device=dev)
    capi.call("scatter_idx", idx.to(dev), emb_idx.to(dev), all_idx, n_rays)
    assert torch.equal(all_idx.cpu(), ref_all)
    emb = torch.randn(E, C, generator=g)
    to_add = torch.randn(n, C, generator=g)
    out = torch.empty(n, C, device=dev)
    capi.call("scatter_add_fwd", emb.to(dev), all_idx, to_add.to(dev), out, n, C)
    assert torch.equal(out.cpu(), K.scatter_add_fwd(emb, ref_all, to_add))
    dsum = torch.randn(n, C, generator=g)
    demb = torch.full((E, C), 3.0, device=dev)  # must be overwritten, not accumulated into
    capi.call("scatter_add_bwd", all_idx, dsum.to(dev), demb, n, E, C)
    ref = K.scatter_add_bwd(ref_all, dsum, E)
    torch.testing.assert_close(demb.cpu(), ref, rtol=1e-4, atol=1e-4)


# --------------------------------------------------------------------------------- sampler ------


def _rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(n, 3, generator=g) * 0.3
    d = torch.randn(n, 3, generator=g) * 2.0
    return o, d


@pytest.mark.parametrize("S,step,train", [(1024, 1.0 / 256, True), (1024, 1.0 / 256, False),
                                          (128, 4.0 / 128, True), (100, 0.05, True)])
def test_sample_rays(capi, dev, S, step, train):
    from oracle import ref_render as R
    n = 97
    o, d = _rays(n, 1)
    g = torch.Generator().manual_seed(5)
    noise = (torch.rand(n, S, generator=g) - 0.5 + 1.0) if train else None
    pts, dirs, dt, t, bounds = R.get_samples(o, d, noise, S, step)
    N = n * S
    o_pts, o_dirs = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
    o_dt, o_t = torch.empty(N, device=dev), torch.empty(N, device=dev)
    o_b = torch.empty(n, 2, dtype=torch.int32, device=dev)
    capi.call("sample_rays", o.to(dev), d.to(dev), noise.to(dev) if train else None, o_pts,
              o_dirs, o_dt, o_t, o_b, n, S, step)
    assert torch.equal(o_b.cpu(), bounds)
    torch.testing.assert_close(o_dirs.cpu(), dirs, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(o_t.cpu(), t, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(o_pts.cpu(), pts, rtol=1e-5, atol=2e-6)
    # dt is a difference of nearby points (quirk Q7): absolute error ~ ulp(|p|), relative to dt ~1e-4
    torch.testing.assert_close(o_dt.cpu(), dt, rtol=1e-3, atol=2e-6)
    assert (o_dt.cpu().reshape(n, S)[:, 0] == 0).all()


# --------------------------------------------------------------------------- ray generation ----


def test_gen_rays_against_oracle(capi, dev):
    """f2n_gen_rays in its three addressing modes against the oracle's restatement of
    get_rays_from_pose (reference src/rays.cpp:7-28).  Origins are copies (exact); directions are a
    3-term dot product whose order the reference's matmul does not fix: 1e-6."""
    from oracle import ref_render as R
    g = torch.Generator().manual_seed(4)
    E, h, w = 5, 37, 53
    poses = torch.randn(E, 3, 4, generator=g)
    K = torch.tensor([[1111.1, 0, w / 2], [0, 1100.0, h / 2], [0, 0, 1.0]]).repeat(E, 1, 1)
    K[:, 0, 0] += torch.arange(E) * 3.0
    d_poses, d_K = poses.to(dev), K.to(dev)

    # (1) a view's pixel grid, no ij tensor, starting in the middle of the image
    first, n = 100, h * w - 100
    o = torch.empty(n, 3, device=dev)
    d = torch.empty(n, 3, device=dev)
    capi.call("gen_rays", d_poses[2:3].contiguous(), 12, d_K[2:3].contiguous(), 1, None, None, first, w, o, d, n)
    px = torch.arange(first, first + n)
    ij = torch.stack([px // w, px % w], 1).float()
    ro, rd = R.get_rays_from_pose(poses[2:3], K[2:3], ij)
    assert torch.equal(o.cpu(), ro)
    torch.testing.assert_close(d.cpu(), rd, rtol=1e-6, atol=1e-6)

    # (2) random batch: one camera per ray through cam_idx (src/dataset.cpp:150-171)
    n = 1000
    cam = torch.randint(0, E, (n,), generator=g).to(torch.int32)
    ij = torch.stack([torch.randint(0, h, (n,), generator=g), torch.randint(0, w, (n,), generator=g)], 1)
    o = torch.empty(n, 3, device=dev)
    d = torch.empty(n, 3, device=dev)
    capi.call("gen_rays", d_poses, 12, d_K, E, cam.to(dev), ij.to(torch.int32).to(dev).contiguous(), 0, 1, o, d, n)
    ro, rd = R.get_rays_from_pose(poses[cam.long()], K[cam.long()], ij.float())
    assert torch.equal(o.cpu(), ro)
    torch.testing.assert_close(d.cpu(), rd, rtol=1e-6, atol=1e-6)

    # (3) one [4,4] pose per ray, no cam_idx
    p44 = torch.cat([poses[cam.long()], torch.tensor([0., 0, 0, 1]).expand(n, 1, 4)], 1).contiguous()
    o2 = torch.empty(n, 3, device=dev)
    d2 = torch.empty(n, 3, device=dev)
    capi.call("gen_rays", p44.to(dev), 16, d_K[cam.long().to(dev)].contiguous(), n, None,
              ij.to(torch.int32).to(dev).contiguous(), 0, 1, o2, d2, n)
    assert torch.equal(o2, o) and torch.equal(d2, d)

    with pytest.raises(capi.F2NError):   # 7 cameras for 1000 rays and no cam_idx
        capi.call("gen_rays", d_poses, 12, d_K, 7, None, None, 0, w, o, d, n)
