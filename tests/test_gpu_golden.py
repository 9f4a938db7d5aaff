"""The HIP path against the committed golden fixtures (tests/golden/*.npz): the same files the CPU
suite pins the oracle with, so GPU results are tied to stored data and not only to an oracle run in
the same process."""
import importlib
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_hash_against_golden(capi, dev):
    z = _load("hash_ref_config")
    fld = util.make_field(16, 2, 19, None, seed=2022)   # the table is regenerated from its seed
    assert torch.equal(fld["primes"], z["primes"]) and torch.equal(fld["bias"], z["bias"])
    n = z["pts"].shape[0]
    d = [t.to(dev) for t in (z["pts"], fld["table16"], z["primes"], z["bias"], z["mul"])]
    out = torch.empty(n, 32, device=dev)
    rows = torch.empty(n, 16, 8, dtype=torch.int32, device=dev)
    capi.call("hash_fwd", *d, out, 32, 1, rows, n, 16, 2, fld["T"], fld["stride"])
    assert torch.equal(rows.cpu(), z["rows"])           # hash rows: bit-exact
    assert torch.equal(out.cpu(), z["out"])             # f16-rounded features: bit-exact
    tg = torch.zeros(fld["table"].numel(), device=dev)
    pg = torch.empty(n, 3, device=dev)
    capi.call("hash_bwd", *d, z["grad"].to(dev), 32, 1, tg, pg, n, 16, 2, fld["T"], fld["stride"],
              128.0)
    nz = torch.nonzero(tg.cpu()).reshape(-1).to(torch.int32)
    assert torch.equal(nz, z["table_grad_index"])       # exactly the same table entries are touched
    torch.testing.assert_close(tg.cpu()[nz.long()], z["table_grad_value"], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(pg.cpu(), z["pts_grad"], rtol=1e-4,
                               atol=1e-5 * float(z["pts_grad"].abs().max()))


def test_segments_and_sh_against_golden(capi, dev):
    z = _load("segments")
    idx, val, w = z["idx"].to(dev), z["val"].to(dev), z["w"].to(dev)
    R, n = idx.shape[0], val.shape[0]
    out = torch.empty(R, device=dev)
    capi.call("seg_sum_fwd", val, idx, out, R)
    torch.testing.assert_close(out.cpu(), z["sum"], rtol=1e-5, atol=1e-6)
    for key, inc, fn in (("scan_excl", 0, "seg_scan_fwd"), ("scan_incl", 1, "seg_scan_fwd"),
                         ("scan_bwd_excl", 0, "seg_scan_bwd")):
        o = torch.zeros(n, device=dev)
        capi.call(fn, val, idx, o, R, inc)
        torch.testing.assert_close(o.cpu(), z[key], rtol=1e-5, atol=1e-5)
    o = torch.empty(R, device=dev)
    capi.call("weight_var_fwd", w, idx, o, R)
    torch.testing.assert_close(o.cpu(), z["var"], rtol=2e-4, atol=1e-7)
    o = torch.zeros(n, device=dev)
    capi.call("weight_var_bwd", w, idx, z["dvar"].to(dev), o, R)
    torch.testing.assert_close(o.cpu(), z["var_bwd"], rtol=2e-4,
                               atol=1e-4 * float(z["var_bwd"].abs().max()))
    s = _load("sh")
    o = torch.empty(s["dirs"].shape[0], 16, device=dev)
    capi.call("sh_encode", s["dirs"].to(dev), o, s["dirs"].shape[0], 4)
    torch.testing.assert_close(o.cpu(), s["sh"], rtol=1e-6, atol=1e-7)


def test_render_against_golden(dev):
    H = importlib.import_module("f2-nerf_amd").load_host()
    z = _load("render_small")
    ren = H.Renderer(3, n_levels=4, n_channels=2, log2_table=10, max_samples=64, step=4.0 / 64)
    with torch.no_grad():
        for k, v in ren.named_parameters().items():
            v.copy_(z["param." + k].to(dev))
    to = lambda k: z[k].to(dev)
    for fused, fused_shade, dense in ((True, True, 0), (True, True, 1), (True, False, 0),
                                      (False, False, 0)):
        ren.set_fused(fused)
        ren.set_fused_shade(fused_shade)
        ren.set_dense_first_pass(dense)
        ren.zero_grad()
        c, d, w, idx = ren.render(to("rays_o"), to("rays_d"), to("emb_idx"), "train", to("noise"),
                                  to("bg"))
        assert torch.equal(idx.cpu(), z["idx_start_end"])
        torch.testing.assert_close(c.detach().cpu(), z["colors"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(d.detach().cpu(), z["depths"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(w.detach().cpu(), z["weights"], rtol=1e-4, atol=1e-7)
        ren.zero_grad()
        loss, _, _, _ = ren.train_step(to("rays_o"), to("rays_d"), to("emb_idx"), to("gt"), 1e-2,
                                       to("noise"), to("bg"), True)
        assert abs(float(loss) - float(z["loss"][0])) <= 1e-5 * abs(float(z["loss"][0]))
        g = ren.grads()
        ref = z["feat_pool_grad"]
        assert ((g["scene_field.feat_pool"].cpu() - ref).norm() / ref.norm()) < 2e-4
        torch.testing.assert_close(g["app_emb"].cpu(), z["app_emb_grad"], rtol=1e-3,
                                   atol=1e-4 * float(z["app_emb_grad"].abs().max()))


def test_raytile_and_rays_against_golden(capi, dev):
    """f2n_hash_fwd_raytile on the golden points read as a 16 x 16 sample grid; f2n_gen_rays on the
    stored cameras and pixels."""
    z = _load("hash_ref_config")
    fld = util.make_field(16, 2, 19, None, seed=2022)
    d = [t.to(dev) for t in (z["pts"], fld["table16"], z["primes"], z["bias"], z["mul"])]
    out = torch.empty(32, 256, device=dev)
    capi.call("hash_fwd_raytile", *d, out, 16, 16, 16, 2, fld["T"], fld["stride"])
    assert torch.equal(out.t().contiguous().cpu(), z["out"])
    r = _load("rays")
    n = r["ij"].shape[0]
    o = torch.empty(n, 3, device=dev)
    dd = torch.empty(n, 3, device=dev)
    capi.call("gen_rays", r["poses"].to(dev), 12, r["intrinsics"].to(dev), r["poses"].shape[0],
              r["cam_idx"].to(dev), r["ij"].to(dev).contiguous(), 0, 1, o, dd, n)
    assert torch.equal(o.cpu(), r["rays_o"])
    torch.testing.assert_close(dd.cpu(), r["rays_d"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("fwd_route,bwd_route", [(0, 0), (1, 1)], ids=["matrix-core", "vector"])
def test_shade_network_against_golden(capi, dev, fwd_route, bwd_route):
    """f2n_shade_fwd / f2n_shade_bwd (matrix-core and vector kernels) against the stored outputs and
    gradients of the op-by-op network."""
    z = _load("shade_network")
    capi.set_option("SHADE_FWD", fwd_route)
    capi.set_option("SHADE_BWD", bwd_route)
    n, C = z["enc"].shape
    dv = lambda t: t.to(dev).contiguous()
    enc_cm = dv(z["enc"].t())
    P = {k: dv(z["param." + k]) for k in ("w_h", "b_h", "w1", "b1", "w2", "b2", "emb")}
    logit, rgb = torch.empty(n, device=dev), torch.empty(n, 3, device=dev)
    args = (enc_cm, C, dv(z["dirs"]), dv(z["img"]), P["w_h"], P["b_h"], P["w1"], P["b1"], P["w2"], P["b2"],
            P["emb"])
    capi.call("shade_fwd", *args, logit, rgb, None, n)
    torch.testing.assert_close(logit.cpu(), z["logit"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rgb.cpu(), z["rgb"], rtol=1e-4, atol=1e-5)
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    d_enc = torch.empty(C, n, device=dev)
    capi.call("shade_bwd", *args, dv(z["d_logit"]), dv(z["d_rgb"]), d_enc, G["w_h"], G["b_h"], G["w1"],
              G["b1"], G["w2"], G["b2"], G["emb"], None, n)
    torch.testing.assert_close(d_enc.t().cpu(), z["d_enc"], rtol=1e-3,
                               atol=1e-4 * float(z["d_enc"].abs().max()))
    for k in G:
        ref = z["grad." + k]
        torch.testing.assert_close(G[k].cpu(), ref, rtol=1e-3, atol=2e-4 * float(ref.abs().max()),
                                   msg=lambda m, k=k: k + ": " + m)
