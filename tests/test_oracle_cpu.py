"""The CPU oracle checked against INDEPENDENT formulations (plain torch ops, torch autograd,
quadrature) and against the committed golden fixtures.  The reference ships no vectors for this
path, so this is what stands behind the oracle: parity with the reference
itself stays "unpinned" (DESIGN.md)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import kernels as K
from oracle import ref_render as R
from tests import util

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_f16_emulation_matches_ieee():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(200000, generator=g) * s for s in (1e-7, 1e-4, 1.0, 300.0, 7e4)])
    x = torch.cat([x, torch.tensor([0.0, -0.0, 65504.0, 65519.9, 65520.0, 2.0 ** -24, 2.0 ** -25,
                                    2.0 ** -25 * 1.0001, float("inf"), -float("inf")])])
    assert torch.equal(K.cast_f16(x), x.to(torch.float16).view(torch.int16))


def test_level_mul_values():
    # SURVEY.md section 8 A1 lists the 16 values glibc exp2f gives for L = 16
    want = [8, 11.0553036, 15.2774668, 21.1121273, 29.1751213, 40.3174629, 55.7152443, 76.9935913,
            106.398468, 147.033371, 203.187317, 280.78717, 388.023529, 536.214539, 741.001587, 1024]
    got = K.level_mul(16)
    assert torch.allclose(got, torch.tensor(want, dtype=torch.float32), rtol=1e-7, atol=0)
    assert got[0] == 8.0 and got[-1] == 1024.0


def _torch_hash(pts, fld):
    """Independent statement of appendix A.3 with torch ops (FMA emulated through float64)."""
    L, F, T, st = fld["L"], fld["F"], fld["T"], fld["stride"]
    table = fld["table16"].view(torch.float16).to(torch.float64)
    outs, rows_all = [], []
    for l in range(L):
        mul = fld["mul"][l].double()
        pt = (pts.double() * mul + fld["bias"][l].double()).float()     # one rounding = fmaf
        fl = torch.floor(pt)
        c = fl.clamp_min(0).to(torch.int64)                              # saturate negatives (Q1)
        fr = (pt - fl)
        pr = fld["primes"][l].to(torch.int64) & 0xFFFFFFFF
        M = 0xFFFFFFFF
        acc = None
        rows = []
        for d in range(8):
            dx, dy, dz = (d >> 2) & 1, (d >> 1) & 1, d & 1
            h = (((c[:, 0] + dx) * pr[0]) & M) ^ (((c[:, 1] + dy) * pr[1]) & M) ^ \
                (((c[:, 2] + dz) * pr[2]) & M)
            row = h % T
            rows.append(row)
            wx = fr[:, 0] if dx else (1 - fr[:, 0])
            wy = fr[:, 1] if dy else (1 - fr[:, 1])
            wz = fr[:, 2] if dz else (1 - fr[:, 2])
            w = ((wx * wy) * wz).double()
            f = torch.stack([table[st * l + row * F + k] for k in range(F)], 1)
            term = w[:, None] * f                                        # exact in f64
            acc = term.float().double() if acc is None else (term + acc).float().double()
        outs.append(acc.float().to(torch.float16).float())
        rows_all.append(torch.stack(rows, 1))
    return torch.cat(outs, 1), torch.stack(rows_all, 1)


@pytest.mark.parametrize("L,F,log2_T,disjoint", [(16, 2, 19, False), (4, 2, 12, False), (3, 8, 9, True)])
def test_hash_fwd_oracle_vs_torch(L, F, log2_T, disjoint):
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, T * F if disjoint else None, seed=5)
    pts = util.ball_points(3000, seed=2)
    pts[0] = torch.tensor([-1.99, -1.99, -1.99])
    out, idx = K.hash_fwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], L, F, T,
                          fld["stride"], want_idx=True)
    ref, ref_rows = _torch_hash(pts, fld)
    assert torch.equal(idx.to(torch.int64) & 0xFFFFFFFF, ref_rows)      # hash rows: bit-exact
    # values: identical up to the (rare) double rounding of the f64-emulated FMA chain
    diff = (out - ref).abs()
    assert (diff <= util.f16_ulp(ref)).all()
    assert (diff != 0).float().mean() < 1e-3


def test_hash_negative_coordinates_saturate():
    """Quirk Q1: (unsigned)floorf(negative) is 0 on the GPU.  A point whose scaled coordinate is
    negative must hash like coordinate 0, not like 2^32 - k."""
    L, F, T = 1, 2, 1 << 10
    fld = util.make_field(L, F, 10, None, seed=1)
    fld["bias"][0] = torch.tensor([0.25, 0.25, 0.25])   # so that small negative points go < 0
    a = torch.tensor([[-0.5, -0.5, -0.5]])
    b = torch.tensor([[-0.01, -0.01, -0.01]])
    _, ia = K.hash_fwd(a, fld["table16"], fld["primes"], fld["bias"], fld["mul"], L, F, T, T, True)
    _, ib = K.hash_fwd(b, fld["table16"], fld["primes"], fld["bias"], fld["mul"], L, F, T, T, True)
    assert torch.equal(ia, ib)    # both floor to a negative cell -> both clamp to cell 0
    pr = fld["primes"][0].to(torch.int64)
    assert int(ia[0, 0, 0]) & 0xFFFFFFFF == 0                       # corner (0,0,0) -> hash 0
    assert int(ia[0, 0, 7]) & 0xFFFFFFFF == int((pr[0] ^ pr[1] ^ pr[2]) % T)


def test_hash_bwd_oracle_vs_autograd():
    """d(out)/d(table) of the trilinear blend: the oracle (f16-rounded contributions, x128 scaling)
    against torch autograd of the same blend in f64 -- equal to f16 resolution."""
    L, F, log2_T = 4, 2, 8
    T = 1 << log2_T
    fld = util.make_field(L, F, log2_T, T * F, seed=9)
    n = 4000
    pts = util.ball_points(n, seed=4)
    _, rows = _torch_hash(pts, fld)
    g = torch.Generator().manual_seed(1)
    grad = torch.randn(n, L * F, generator=g) * 1e-2
    tg, _ = K.hash_bwd(pts, fld["table16"], fld["primes"], fld["bias"], fld["mul"], grad,
                       fld["table"].numel(), L, F, T, fld["stride"], 128.0)
    ref = torch.zeros(fld["table"].numel(), dtype=torch.float64)
    for l in range(L):
        pt = (pts.double() * fld["mul"][l].double() + fld["bias"][l].double()).float()
        fr = (pt - torch.floor(pt)).double()
        for d in range(8):
            dx, dy, dz = (d >> 2) & 1, (d >> 1) & 1, d & 1
            w = (fr[:, 0] if dx else 1 - fr[:, 0]) * (fr[:, 1] if dy else 1 - fr[:, 1]) * \
                (fr[:, 2] if dz else 1 - fr[:, 2])
            for k in range(F):
                ref.index_add_(0, fld["stride"] * l + rows[:, l, d] * F + k,
                               w * grad[:, l * F + k].double())
    rel = (tg.double() - ref).norm() / ref.norm()
    assert rel < 2e-3, rel          # f16 quantisation of g and g*w: ~2^-11 relative per term
    assert (tg.double() - ref).abs().max() <= 5e-3 * ref.abs().max()


def test_sh_basis_is_orthonormal():
    """Convention-free check of the 16 SH basis functions: the Gram matrix over the sphere is I."""
    nt, npz = 64, 128
    x, wq = np.polynomial.legendre.leggauss(nt)                # cos(theta) nodes
    phi = (np.arange(npz) + 0.5) * 2 * np.pi / npz
    ct, ph = np.meshgrid(x, phi, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    dirs = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3)
    w = (wq[:, None] * np.full((1, npz), 2 * np.pi / npz)).reshape(-1)
    Y = K.sh_encode(torch.tensor(dirs, dtype=torch.float32), 4).double().numpy()
    gram = (Y * w[:, None]).T @ Y
    assert np.abs(gram - np.eye(16)).max() < 2e-5
    # band structure: l = 1 terms are -y, z, -x scaled by sqrt(3/(4 pi)) (reference sh_shader.cu:34-36)
    c1 = math.sqrt(3 / (4 * math.pi))
    d = torch.tensor([[0.6, -0.48, 0.64]])
    y = K.sh_encode(d, 2)[0]
    assert torch.allclose(y[1:4], torch.tensor([0.48 * c1, 0.64 * c1, -0.6 * c1]), atol=1e-6)


def test_sh_bands_4_to_7_match_the_closed_forms_the_reference_documents():
    """Degrees 5..8: the oracle evaluates the defining recurrences in double; the reference's
    comments (src/sh_shader.cu:52-102) give each entry in closed form.  A dozen of them -- every
    band's m = -l, 0, +l and some in between -- pin ordering, signs and normalisation; the Gram
    matrix over the sphere pins the rest up to rotation within a band."""
    g = torch.Generator().manual_seed(5)
    d = torch.randn(257, 3, generator=g, dtype=torch.float64)
    d = d / d.norm(dim=1, keepdim=True)
    Y = K.sh_encode(d.float(), 8).double()
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    x2, y2, z2 = x * x, y * y, z * z
    x4, y4, z4, x6, y6, z6 = x2 * x2, y2 * y2, z2 * z2, x2 ** 3, y2 ** 3, z2 ** 3
    sp = math.sqrt(math.pi)
    want = {
        16: 3 * math.sqrt(35) * x * y * (x2 - y2) / (4 * sp),
        20: 3 * (-30 * z2 + 35 * z4 + 3) / (16 * sp),
        24: 3 * math.sqrt(35) * (-6 * x2 * y2 + x4 + y4) / (16 * sp),
        25: 3 * math.sqrt(154) * y * (10 * x2 * y2 - 5 * x4 - y4) / (32 * sp),
        27: -math.sqrt(770) * y * (3 * x2 - y2) * (9 * z2 - 1) / (32 * sp),
        30: math.sqrt(11) * z * (-70 * z2 + 63 * z4 + 15) / (16 * sp),
        35: 3 * math.sqrt(154) * x * (10 * x2 * y2 - x4 - 5 * y4) / (32 * sp),
        36: math.sqrt(6006) * x * y * (-10 * x2 * y2 + 3 * x4 + 3 * y4) / (32 * sp),
        42: math.sqrt(13) * (105 * z2 - 315 * z4 + 231 * z6 - 5) / (32 * sp),
        45: -math.sqrt(2730) * x * z * (x2 - 3 * y2) * (11 * z2 - 3) / (32 * sp),
        48: math.sqrt(6006) * (15 * x2 * y4 - 15 * x4 * y2 + x6 - y6) / (64 * sp),
        49: 3 * math.sqrt(715) * y * (-21 * x2 * y4 + 35 * x4 * y2 - 7 * x6 + y6) / (64 * sp),
        56: math.sqrt(15) * z * (315 * z2 - 693 * z4 + 429 * z6 - 35) / (32 * sp),
        58: math.sqrt(70) * z * (x2 - y2) * (143 * z2 * (3 * z2 - 1) - 187 * z2 + 45) / (64 * sp),
        63: 3 * math.sqrt(715) * x * (-35 * x2 * y4 + 21 * x4 * y2 - x6 + 7 * y6) / (64 * sp),
    }
    for idx, w in want.items():
        assert (Y[:, idx] - w).abs().max() < 3e-6, idx
    # degrees 5..8 are prefixes of one another, and degree 4 is the prefix the renderer uses
    for deg in (4, 5, 6, 7):
        assert torch.equal(K.sh_encode(d.float(), deg), K.sh_encode(d.float(), 8)[:, :deg * deg])
    nt, npz = 64, 128
    xs, wq = np.polynomial.legendre.leggauss(nt)
    phi = (np.arange(npz) + 0.5) * 2 * np.pi / npz
    ct, ph = np.meshgrid(xs, phi, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    dirs = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3)
    w = (wq[:, None] * np.full((1, npz), 2 * np.pi / npz)).reshape(-1)
    Yq = K.sh_encode(torch.tensor(dirs, dtype=torch.float32), 8).double().numpy()
    gram = (Yq * w[:, None]).T @ Yq
    assert np.abs(gram - np.eye(64)).max() < 5e-5


def _seg_loop(val, idx, fn):
    return [fn(val[s:e]) for s, e in idx.tolist()]


def test_segment_ops_vs_torch_and_autograd():
    idx, n = util.ragged_bounds(57, 90, seed=3)
    g = torch.Generator().manual_seed(3)
    val = torch.rand(n, generator=g, dtype=torch.float32)
    ref = torch.stack([v.double().sum() for v in _seg_loop(val, idx, lambda v: v)]).float()
    torch.testing.assert_close(K.seg_sum_fwd(val, idx), ref, rtol=1e-5, atol=1e-6)
    v3 = torch.rand(n, 3, generator=g)
    ref3 = torch.stack([v.double().sum(0) for v in _seg_loop(v3, idx, lambda v: v)]).float()
    torch.testing.assert_close(K.seg_sum_fwd(v3, idx), ref3, rtol=1e-5, atol=1e-6)
    for inc in (0, 1):
        out = K.seg_scan_fwd(val, idx, inc)
        for s, e in idx.tolist():
            cs = torch.cumsum(val[s:e].double(), 0)
            want = cs if inc else cs - val[s:e].double()
            torch.testing.assert_close(out[s:e].double(), want, rtol=1e-5, atol=1e-6)
    # autograd: FlexSum / FlexAccumulateSum backward == derivative of the torch formulation
    v = val.clone().requires_grad_(True)
    w = torch.randn(n, generator=g)
    (R.flex_accumulate_sum(v, idx, False) * w).sum().backward()
    v2 = val.clone().requires_grad_(True)
    tot = 0
    for s, e in idx.tolist():
        cs = torch.cumsum(v2[s:e], 0) - v2[s:e]
        tot = tot + (cs * w[s:e]).sum()
    tot.backward()
    torch.testing.assert_close(v.grad, v2.grad, rtol=1e-4, atol=1e-5)
    v = val.clone().requires_grad_(True)
    wr = torch.randn(idx.shape[0], generator=g)
    (R.flex_sum(v, idx) * wr).sum().backward()
    want = torch.zeros(n)
    for r, (s, e) in enumerate(idx.tolist()):
        want[s:e] = wr[r]
    assert torch.equal(v.grad, want)


def test_weight_var_vs_definition():
    idx, n = util.ragged_bounds(40, 60, seed=8, empty_frac=0.2)
    g = torch.Generator().manual_seed(5)
    w = torch.rand(n, generator=g) * 0.2
    out = K.weight_var_fwd(w, idx)
    wd = w.double().requires_grad_(True)
    vals = []
    for s, e in idx.tolist():
        if e <= s:
            vals.append(torch.zeros((), dtype=torch.float64))
            continue
        x = torch.arange(e - s, dtype=torch.float64) / 16.0
        ws = wd[s:e]
        m = (ws * x).sum() / (1e-6 + ws.sum())
        vals.append((ws * (x - m) ** 2).sum())
    ref = torch.stack(vals)
    torch.testing.assert_close(out.double(), ref.detach(), rtol=1e-4, atol=1e-7)
    dv = torch.randn(idx.shape[0], generator=g)
    (ref * dv.double()).sum().backward()
    got = K.weight_var_bwd(w, idx, dv)
    # quirk Q9: the coded backward differs from the true derivative by a term ~1e-6 * mean
    torch.testing.assert_close(got.double(), wd.grad, rtol=2e-3, atol=2e-4)


def test_scatter_ops_vs_torch():
    idx, n = util.ragged_bounds(30, 40, seed=2)
    g = torch.Generator().manual_seed(2)
    E, C = 9, 16
    emb_idx = torch.randint(0, E, (30,), generator=g).to(torch.int32)
    all_idx = K.scatter_idx(n, idx, emb_idx)
    want = torch.repeat_interleave(emb_idx, (idx[:, 1] - idx[:, 0]).long())
    assert torch.equal(all_idx, want)
    emb, to_add = torch.randn(E, C, generator=g), torch.randn(n, C, generator=g)
    assert torch.equal(K.scatter_add_fwd(emb, all_idx, to_add), to_add + emb[all_idx.long()])
    dsum = torch.randn(n, C, generator=g)
    ref = torch.zeros(E, C).index_add_(0, all_idx.long(), dsum)
    torch.testing.assert_close(K.scatter_add_bwd(all_idx, dsum, E), ref, rtol=1e-5, atol=1e-5)


def test_sampler_quirks():
    """Q7: dt is the norm of point differences with dt_0 = 0, not noise * step."""
    g = torch.Generator().manual_seed(4)
    o, d = torch.randn(5, 3, generator=g), torch.randn(5, 3, generator=g) * 3
    noise = torch.rand(5, 64, generator=g) + 0.5
    pts, dirs, dt, t, b = R.get_samples(o, d, noise, 64, 1 / 16)
    assert torch.allclose(dirs.norm(dim=1), torch.ones(5 * 64), atol=1e-6)
    assert (dt.reshape(5, 64)[:, 0] == 0).all()
    torch.testing.assert_close(dt.reshape(5, 64)[:, 1:], noise[:, 1:] / 16, rtol=1e-3, atol=1e-6)
    assert torch.equal(b, torch.tensor([[i * 64, (i + 1) * 64] for i in range(5)], dtype=torch.int32))
    # contraction quirk Q6: the origin maps to NaN through the reference's mask expression
    f = R.Hash3DAnchored(2, 2, 8)
    assert torch.isnan(f.query(torch.zeros(1, 3))).all()


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz")))
def test_golden_fixtures(name):
    """Fixtures written by tests/golden/make_golden.py from this oracle (regression pin: integer
    results bit-exact, floats to 1e-6).  They pin the oracle against drift, not against the
    reference, which has no vectors of its own."""
    from tests.golden import make_golden

    z = np.load(os.path.join(GOLDEN, name))
    got = make_golden.CASES[name[:-4]]()
    assert set(got) == set(z.files)
    for k in z.files:
        a, b = got[k], z[k]
        if np.issubdtype(b.dtype, np.integer):
            assert np.array_equal(a, b), (name, k)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7, err_msg="%s:%s" % (name, k))
