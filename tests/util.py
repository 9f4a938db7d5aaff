"""Shared input builders for the parity tests (seeded; same tensors go to the oracle and the GPU)."""
import torch

from oracle import kernels as K


from oracle.ref_render import _is_prime as is_prime  # noqa: E402


def make_field(L=16, F=2, log2_T=19, level_stride=None, seed=0, init="trained"):
    """Hash-grid parameters as Hash3DAnchored's ctor draws them (reference
    src/hash_3d_anchored.cpp:19-58); T = rows per level, stride in elements (default T, quirk Q2)."""
    g = torch.Generator().manual_seed(seed)
    T = 1 << log2_T
    stride = T if level_stride is None else level_stride
    numel = max(T * L * F, stride * (L - 1) + T * F)
    if init == "reference":
        table = (torch.rand(numel, generator=g) * 0.2 - 1.0) * 1e-4
    else:
        table = torch.randn(numel, generator=g) * 0.1
    primes = []
    while len(primes) < 3 * L:
        v = int(torch.randint(1 << 28, 1 << 30, (1,), generator=g))
        if is_prime(v):
            primes.append(v)
    primes = torch.tensor(primes, dtype=torch.int32).reshape(L, 3)
    bias = torch.rand(L, 3, generator=g) * 1000.0 + 100.0
    return dict(L=L, F=F, T=T, stride=stride, table=table, table16=K.cast_f16(table),
                primes=primes, bias=bias, mul=K.level_mul(L))


def ball_points(n, seed=0, radius=2.0):
    """Points inside the radius-2 ball (the image of the scene contraction)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    r = torch.rand(n, 1, generator=g) ** (1.0 / 3.0) * radius
    return (d * r).contiguous()


def ragged_bounds(n_rays, max_len, seed=0, empty_frac=0.1):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(0, max_len + 1, (n_rays,), generator=g)
    lens[torch.rand(n_rays, generator=g) < empty_frac] = 0
    end = torch.cumsum(lens, 0)
    start = end - lens
    return torch.stack([start, end], 1).to(torch.int32).contiguous(), int(end[-1])


def f16_ulp(x):
    """Size of one f16 ulp at magnitude |x| (f32 tensor in, f32 out)."""
    ax = x.abs().clamp_min(2.0 ** -14)
    e = torch.floor(torch.log2(ax))
    return torch.pow(2.0, e - 10)
