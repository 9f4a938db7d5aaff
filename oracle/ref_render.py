"""CPU restatement of the reference's L2 operator layer, arranged as the reference is:
torch-CPU ops wherever the reference calls ATen, the C kernels of f2n_oracle.c wherever it launches
a CUDA kernel.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); PARITY UNPINNED.

Citations are reference file:line.  Random tensors the reference draws on the device (sampling
noise, background colour) are inputs here so that the GPU path and this oracle see the same values.
"""
import math

import torch

from . import kernels as K

TRAIN, VALIDATE = 0, 1


# ------------------------------------------------------------------------------------------------
# autograd Functions (src/CustomOps/*.cu, src/hash_3d_anchored.cu:150-218)
# ------------------------------------------------------------------------------------------------


class Hash3DAnchoredFunction(torch.autograd.Function):
    """src/hash_3d_anchored.cu:150-218"""

    @staticmethod
    def forward(ctx, points, feat_pool, field):
        ctx.field = field
        ctx.save_for_backward(points, feat_pool)
        table16 = K.cast_f16(feat_pool.detach().reshape(-1))  # feat_pool.to(kFloat16), :169
        return K.hash_fwd(points.detach(), table16, field.prim_pool, field.bias_pool.detach(),
                          field.mul, field.L, field.F, field.T, field.level_stride)

    @staticmethod
    def backward(ctx, grad_out):
        points, feat_pool = ctx.saved_tensors
        f = ctx.field
        table16 = K.cast_f16(feat_pool.detach().reshape(-1))  # :198
        tg, pg = K.hash_bwd(points.detach(), table16, f.prim_pool, f.bias_pool.detach(), f.mul,
                            grad_out.contiguous(), feat_pool.numel(), f.L, f.F, f.T,
                            f.level_stride, 128.0, need_pts_grad=ctx.needs_input_grad[0],
                            parallel=f.parallel_bwd)
        return pg, tg.reshape(feat_pool.shape), None


class FlexSum(torch.autograd.Function):
    """src/CustomOps/FlexOps.cu:98-153"""

    @staticmethod
    def forward(ctx, val, idx):
        ctx.save_for_backward(idx)
        ctx.n_all = val.shape[0]
        return K.seg_sum_fwd(val.detach(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return K.seg_sum_bwd(g.contiguous(), idx, ctx.n_all), None


class FlexAccumulateSum(torch.autograd.Function):
    """src/CustomOps/FlexOps.cu:155-199"""

    @staticmethod
    def forward(ctx, val, idx, include_this):
        ctx.save_for_backward(idx)
        ctx.include_this = include_this
        return K.seg_scan_fwd(val.detach(), idx, include_this)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return K.seg_scan_bwd(g.contiguous(), idx, ctx.include_this), None, None


class WeightVarLoss(torch.autograd.Function):
    """src/CustomOps/CustomOps.cu:71-112"""

    @staticmethod
    def forward(ctx, w, idx):
        ctx.save_for_backward(w, idx)
        return K.weight_var_fwd(w.detach(), idx)

    @staticmethod
    def backward(ctx, g):
        w, idx = ctx.saved_tensors
        return K.weight_var_bwd(w.detach(), idx, g.contiguous()), None


class ScatterAddFunc(torch.autograd.Function):
    """src/CustomOps/Scatter.cu:45-102"""

    @staticmethod
    def forward(ctx, emb, idx, to_add):
        ctx.save_for_backward(idx)
        ctx.n_emb = emb.shape[0]
        return K.scatter_add_fwd(emb.detach(), idx, to_add.detach())

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        return K.scatter_add_bwd(idx, g, ctx.n_emb), None, g.clone()


class TruncExp(torch.autograd.Function):
    """src/CustomOps/CustomOps.cpp:10-20"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-100.0, 5.0))


def flex_sum(val, idx):
    return FlexSum.apply(val.contiguous(), idx.contiguous())


def flex_accumulate_sum(val, idx, include_this):
    return FlexAccumulateSum.apply(val.contiguous(), idx.contiguous(), include_this)


def weight_var(w, idx):
    return WeightVarLoss.apply(w.contiguous(), idx.contiguous())


# ------------------------------------------------------------------------------------------------
# modules
# ------------------------------------------------------------------------------------------------


_SMALL_PRIMES = None


def _is_prime(x):
    """Trial division (hash_3d_anchored.cpp:29-35) restricted to prime divisors <= sqrt(2^30)."""
    global _SMALL_PRIMES
    if _SMALL_PRIMES is None:
        sieve = bytearray([1]) * 32769
        sieve[0:2] = b"\x00\x00"
        for i in range(2, 182):
            if sieve[i]:
                sieve[i * i::i] = bytearray(len(sieve[i * i::i]))
        _SMALL_PRIMES = [i for i in range(32769) if sieve[i]]
    if x < 2:
        return False
    for q in _SMALL_PRIMES:
        if q * q > x:
            break
        if x % q == 0:
            return False
    return True


class Hash3DAnchored(torch.nn.Module):
    """src/hash_3d_anchored.cpp:19-88.  L, F, log2_T are compile-time constants in the reference
    (hash_3d_anchored.hpp:10-11, .cpp:21); their defaults here reproduce it.  level_stride defaults
    to T elements like the reference's pointer arithmetic (.cu:70, quirk Q2)."""

    def __init__(self, L=16, F=2, log2_T=19, level_stride=None, gen=None, feat_init="reference"):
        super().__init__()
        self.L, self.F = L, F
        self.pool_size = (1 << log2_T) * L
        self.T = ((self.pool_size // L) >> 4) << 4  # local_size_, .cpp:57-58
        self.level_stride = self.T if level_stride is None else level_stride
        need = self.level_stride * (L - 1) + self.T * F
        rows = max(self.pool_size, (need + F - 1) // F)
        if feat_init == "reference":
            feat = (torch.rand(rows, F, generator=gen) * 0.2 - 1.0) * 1e-4  # .cpp:24
        else:
            feat = torch.randn(rows, F, generator=gen) * 0.1  # "trained-like", SURVEY 8(d)
        self.feat_pool = torch.nn.Parameter(feat)
        primes = []
        while len(primes) < 3 * L:  # .cpp:29-48
            v = int(torch.randint(1 << 28, 1 << 30, (1,), generator=gen))
            if _is_prime(v):
                primes.append(v)
        self.register_buffer("prim_pool", torch.tensor(primes, dtype=torch.int32).reshape(L, 3))
        self.bias_pool = torch.nn.Parameter(torch.rand(L, 3, generator=gen) * 1000.0 + 100.0)
        self.mlp = torch.nn.Linear(L * F, 16)
        self.mul = K.level_mul(L)
        self.parallel_bwd = False

    def query(self, points):
        radius = 1.0
        norm = points.norm(2, dim=1, keepdim=True)  # .cpp:80
        mask = norm <= radius
        x = points * mask + ~mask * (1 + radius - radius / norm) * points / norm  # .cpp:82
        feat = Hash3DAnchoredFunction.apply(x, self.feat_pool, self)
        return self.mlp(feat)


class SHShader(torch.nn.Module):
    """src/sh_shader.cpp:11-29, encode = src/sh_shader.cu:105-115 (no gradient to dirs)"""

    def __init__(self):
        super().__init__()
        self.mlp = torch.nn.Sequential(
            torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 3))

    def query(self, feats, dirs):
        enc = K.sh_encode(dirs.detach(), 4)
        out = self.mlp(torch.cat([feats, enc], -1))
        eps = 1e-3
        return (1.0 + 2.0 * eps) / (1.0 + torch.exp(-out)) - eps


def get_samples(rays_o_raw, rays_d_raw, noise, S=1024, step=1.0 / 256, sampler_device=None):
    """PtsSampler::get_samples, src/points_sampler.cpp:20-64.  `noise` [n_rays, S] replaces the
    device torch::rand of :35 (TRAIN: U[0,1)-0.5+1); None = VALIDATE (ones, :33).

    sampler_device: where these ATen ops run.  None = on the CPU, like the rest of this oracle.  The
    reference runs the very same ops on its device, where linalg_norm and cumsum (a parallel scan)
    associate differently: d^ and t differ by an ulp or a few, the positions by a few 1e-7..1e-6 --
    and because dt is the norm of the DIFFERENCE of neighbouring positions (quirk Q7: 0.004 formed
    from two numbers near 1..4) that re-draws the rounding of dt at the 1e-4..1e-3 relative level.
    Passing the GPU reproduces the reference's samples bit for bit (same ops, same torch build);
    tests/test_gpu_ref_host.py uses it to show that the per-sample differences between the reference
    and this oracle come from the sampler's op placement and from nothing downstream."""
    out_dev = rays_o_raw.device
    dev = out_dev if sampler_device is None else torch.device(sampler_device)
    rays_o = rays_o_raw.to(dev).contiguous()
    rays_d_raw = rays_d_raw.to(dev)
    rays_d = (rays_d_raw / torch.linalg.norm(rays_d_raw, 2, -1, True)).contiguous()
    n_rays = rays_o.shape[0]
    n_all = n_rays * S
    rays_noise = torch.ones(n_rays, S, device=dev) if noise is None else noise.to(dev).reshape(n_rays, S)
    cum_noise = torch.cumsum(rays_noise, 1) * step
    sampled_t = cum_noise.reshape(n_all).contiguous()
    o = rays_o.view(n_rays, 1, 3)
    d = rays_d.view(n_rays, 1, 3)
    sampled_pts = o + d * cum_noise.unsqueeze(-1)
    dist = torch.diff(sampled_pts, 1, 1).norm(2, -1)
    dist = torch.cat([torch.zeros(n_rays, 1, device=dev), dist], 1).contiguous()
    num = torch.full((n_rays,), S, dtype=torch.int32, device=dev)
    cum = torch.cumsum(num, 0).to(torch.int32)
    bounds = torch.stack([cum - num, cum], -1).contiguous()
    dirs = d.expand(-1, S, -1).reshape(n_all, 3).contiguous()
    out = (sampled_pts.view(n_all, 3), dirs, dist.view(n_all), sampled_t, bounds)
    return tuple(x.to(out_dev) for x in out)


class RenderResult:
    def __init__(self, colors, depths, weights, idx_start_end):
        self.colors, self.depths, self.weights, self.idx_start_end = (
            colors, depths, weights, idx_start_end)


class Renderer(torch.nn.Module):
    """src/renderer.cpp:18-123"""

    def __init__(self, n_images, L=16, F=2, log2_T=19, level_stride=None, S=1024,
                 step=1.0 / 256, gen=None, feat_init="reference"):
        super().__init__()
        self.S, self.step = S, step
        self.scene_field = Hash3DAnchored(L, F, log2_T, level_stride, gen, feat_init)
        self.shader = SHShader()
        self.app_emb = torch.nn.Parameter(torch.randn(n_images, 16, generator=gen) * 0.1)
        self.sampler_device = None  # see get_samples

    @staticmethod
    def density_act(x):
        return TruncExp.apply(x - 3.0)  # renderer.cpp:53-56

    def render(self, rays_o, rays_d, emb_idx, mode, noise=None, bg_color=None):
        n_rays = rays_o.shape[0]
        pts, dirs, dt, t, bounds = get_samples(
            rays_o, rays_d, noise if mode == TRAIN else None, self.S, self.step, self.sampler_device)
        if bg_color is None:
            bg_color = torch.full((n_rays, 3), 0.5)  # VALIDATE, renderer.cpp:44
        # ---- first pass: early stop (renderer.cpp:58-90)
        scene_feat = self.scene_field.query(pts)
        density = self.density_act(scene_feat[:, 0:1])
        sec = density[:, 0] * dt
        acc = flex_accumulate_sum(sec, bounds, False)
        trans = torch.exp(-acc)
        mask = trans > 1e-4
        mask_idx = torch.where(mask)[0]
        pts2 = pts[mask_idx].contiguous()
        dirs2 = dirs[mask_idx].contiguous()
        dt2 = dt[mask_idx].contiguous()
        t2 = t[mask_idx].contiguous()
        num = mask.reshape(n_rays, self.S).sum(1)
        cum = torch.cumsum(num, 0)
        idx = torch.stack([cum - num, cum], -1).to(torch.int32).contiguous()
        # ---- second pass (renderer.cpp:92-118)
        scene_feat = self.scene_field.query(pts2)
        density = self.density_act(scene_feat[:, 0:1])
        shading_feat = torch.cat([torch.ones_like(scene_feat[:, 0:1]), scene_feat[:, 1:]], 1)
        if mode == TRAIN:
            all_emb_idx = K.scatter_idx(pts2.shape[0], idx, emb_idx)
            shading_feat = ScatterAddFunc.apply(self.app_emb, all_emb_idx, shading_feat)
        colors_s = self.shader.query(shading_feat, dirs2)
        sampled_t = (t2 + 1e-2).contiguous()
        sec = density[:, 0] * dt2
        alphas = 1.0 - torch.exp(-sec)
        acc = flex_accumulate_sum(sec, idx, False)
        trans = torch.exp(-acc)
        weights = trans * alphas
        last_trans = torch.exp(-flex_sum(sec, idx))
        colors = flex_sum(weights.unsqueeze(-1) * colors_s, idx)
        colors = colors + last_trans.unsqueeze(-1) * bg_color
        depths = flex_sum(weights * sampled_t, idx) / (1.0 - last_trans + 1e-4)
        return RenderResult(colors, depths, weights, idx)


def train_loss(renderer, rays_o, rays_d, emb_idx, gt_colors, noise, bg_color, var_loss_weight=0.0):
    """The timed harness segment of TrainManager::train, src/main_functions/train_manager.cpp:76-96."""
    res = renderer.render(rays_o, rays_d, emb_idx, TRAIN, noise, bg_color)
    color_loss = torch.sqrt((res.colors - gt_colors).square() + 1e-4).mean()
    sampled_var = weight_var(res.weights, res.idx_start_end)
    var_loss = (sampled_var + 1e-2).sqrt().mean()
    loss = color_loss + var_loss * var_loss_weight
    mse = (res.colors - gt_colors).square().mean()
    mse = mse.detach()
    psnr = 20.0 * math.log10(1.0 / math.sqrt(float(mse)))
    return loss, res, float(mse), psnr


def get_rays_from_pose(pose, intrinsic, ij):
    """src/rays.cpp:7-28.  pose [B,3,4] (or [B,4,4]), intrinsic [B,3,3], ij [N,2] (row, col)."""
    i = ij[..., 0].to(torch.float32) + 0.5
    j = ij[..., 1].to(torch.float32) + 0.5
    cx, cy = intrinsic[:, 0, 2], intrinsic[:, 1, 2]
    fx, fy = intrinsic[:, 0, 0], intrinsic[:, 1, 1]
    u = ((j - cx) / fx).unsqueeze(-1)
    v = -((i - cy) / fy).unsqueeze(-1)
    w = -torch.ones_like(u)
    dir_t = torch.cat([u, v, w], 1).unsqueeze(-1)
    ori = pose[:, 0:3, 0:3]
    pos = pose[:, 0:3, 3]
    rays_d = torch.matmul(ori, dir_t).squeeze(-1)
    rays_o = pos.expand(rays_d.shape[0], 3).contiguous()
    return rays_o, rays_d
