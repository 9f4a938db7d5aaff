"""overflow counter of the binned backward on the hot-spot geometry of tests/test_gpu_ops.py, per level count"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
capi = importlib.import_module("f2-nerf_amd").capi
dev = torch.device("cuda:0")
L, F, log2_T = 16, 2, 19
T = 1 << log2_T
fld = util.make_field(L, F, log2_T, None, seed=5)
g = torch.Generator().manual_seed(8)
n_rays = 20000
o = torch.tensor([0.31, -0.22, 0.17]) + torch.zeros(n_rays, 1, 3)
d = torch.randn(n_rays, 1, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
k = torch.randint(3, 5, (n_rays,), generator=g)
t = (torch.arange(1, 5).float() / 32.0).reshape(1, 4, 1)
keep = (torch.arange(4).reshape(1, 4) < k.reshape(-1, 1)).reshape(-1)
pts = (o + d * t).reshape(-1, 3)[keep].contiguous(); n = pts.shape[0]
cd = capi.lib().cdll
need = cd.f2n_hash_bwd_workspace_bytes(n, L, F, T)
ws = torch.empty(need, dtype=torch.uint8, device=dev)
numel = fld["table"].numel()
for gs in (1e-2, 2e-6):
    grad = (torch.randn(n, L * F, generator=g) * gs).t().contiguous().to(dev)
    dd = [x.to(dev) for x in (pts, fld["primes"], fld["bias"], fld["mul"])] + [grad]
    for comb in (0, 1):
        row = []
        for Lc in range(1, 17):
            ov = torch.zeros(1, dtype=torch.int64, device=dev)
            cd.f2n_hash_bwd_set_overflow_counter(ov.data_ptr())
            with capi.option("BWD_COMBINE", comb):
                tg = torch.zeros(numel, device=dev)
                capi.call("hash_bwd_binned", *dd, 1, n, tg, n, Lc, F, T, fld["stride"], 128.0, ws, need)
            torch.cuda.synchronize(); cd.f2n_hash_bwd_set_overflow_counter(None)
            row.append(int(ov.item()))
        print("grad %g combine_off %d: overflow with the first L levels:" % (gs, comb), row)
