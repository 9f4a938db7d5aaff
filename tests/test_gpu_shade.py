"""Fused per-sample network kernels (f2n_shade_fwd / f2n_shade_bwd) against the op-by-op torch-CPU
composition of the same reference lines (hash_3d_anchored.cpp:86, renderer.cpp:93-104,
sh_shader.cpp:22-29) with torch autograd for every gradient."""
import pytest
import torch

from oracle import kernels as K

pytestmark = pytest.mark.gpu
EPS = 1e-3


def _reference(enc, dirs, img, P, d_logit, d_rgb):
    enc = enc.clone().requires_grad_(True)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    h = enc @ P["w_h"].t() + P["b_h"]
    logit = h[:, 0]
    X = torch.cat([torch.ones_like(h[:, :1]), h[:, 1:]], 1)
    if img is not None:
        X = X + P["emb"][img.long()]
    X = torch.cat([X, K.sh_encode(dirs, 4)], 1)
    hid = torch.relu(X @ P["w1"].t() + P["b1"])
    o = hid @ P["w2"].t() + P["b2"]
    rgb = (1 + 2 * EPS) / (1 + torch.exp(-o)) - EPS
    ((logit * d_logit).sum() + (rgb * d_rgb).sum()).backward()
    return logit.detach(), rgb.detach(), enc.grad, {k: v.grad for k, v in P.items()}


@pytest.mark.parametrize("C,n,with_emb", [(32, 5000, True), (32, 64 * 9 + 17, False), (8, 3000, True),
                                          (64, 2000, True), (16, 1, True), (16, 777, False),
                                          (32, 64 * 2100 + 5, True)])
def test_shade_fwd_bwd(capi, dev, C, n, with_emb):
    g = torch.Generator().manual_seed(C + n)
    E = 5
    enc = (torch.randn(n, C, generator=g) * 0.1).to(torch.float16).float()
    dirs = torch.randn(n, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    # image ids: runs of equal ids (samples of one ray) with a few changes inside a 64-sample stride
    img = (torch.arange(n) // 37 % E).to(torch.int32) if with_emb else None
    P = {"w_h": torch.randn(16, C, generator=g) * 0.3, "b_h": torch.randn(16, generator=g) * 0.1,
         "w1": torch.randn(64, 32, generator=g) * 0.3, "b1": torch.randn(64, generator=g) * 0.1,
         "w2": torch.randn(3, 64, generator=g) * 0.3, "b2": torch.randn(3, generator=g) * 0.1,
         "emb": torch.randn(E, 16, generator=g) * 0.1}
    d_logit = torch.randn(n, generator=g)
    d_rgb = torch.randn(n, 3, generator=g)
    r_logit, r_rgb, r_denc, r_g = _reference(enc, dirs, img, P, d_logit, d_rgb)

    dv = lambda t: t.to(dev).contiguous()
    enc_cm = dv(enc.t())
    Pd = {k: dv(v) for k, v in P.items()}
    d_img = dv(img) if with_emb else None
    emb = Pd["emb"] if with_emb else None
    logit = torch.empty(n, device=dev)
    rgb = torch.empty(n, 3, device=dev)
    pre_cm = torch.empty(64, n, device=dev)
    # forward: the matrix-core kernel (default) and the vector kernel (F2N_OPT_SHADE_FWD = 1)
    pre_first = None
    for froute in ("mfma", "vector"):
        capi.set_option("SHADE_FWD", 1 if froute == "vector" else 0)
        logit.fill_(7.0)
        rgb.fill_(7.0)
        pre_cm.fill_(7.0)
        capi.call("shade_fwd", enc_cm, C, dv(dirs), d_img, Pd["w_h"], Pd["b_h"], Pd["w1"], Pd["b1"],
                  Pd["w2"], Pd["b2"], emb, logit, rgb, pre_cm, n)
        torch.testing.assert_close(logit.cpu(), r_logit, rtol=1e-4, atol=1e-5, msg=lambda m: froute + " logit: " + m)
        torch.testing.assert_close(rgb.cpu(), r_rgb, rtol=1e-4, atol=1e-5, msg=lambda m: froute + " rgb: " + m)
        if pre_first is None:
            pre_first = pre_cm.clone()
        else:   # the hidden pre-activations the two kernels hand to a backward agree
            torch.testing.assert_close(pre_cm, pre_first, rtol=1e-4, atol=1e-5)
    capi.set_option("SHADE_FWD", 0)
    # the matrix-core forward at two and three waves per SIMD (default: four): the same instruction
    # stream per wave, so the same bits
    capi.call("shade_fwd", enc_cm, C, dv(dirs), d_img, Pd["w_h"], Pd["b_h"], Pd["w1"], Pd["b1"],
              Pd["w2"], Pd["b2"], emb, logit, rgb, pre_cm, n)
    base = (logit.clone(), rgb.clone(), pre_cm.clone())
    for variant in (2, 3):
        with capi.option("SHADE_VARIANT", variant):
            capi.call("shade_fwd", enc_cm, C, dv(dirs), d_img, Pd["w_h"], Pd["b_h"], Pd["w1"], Pd["b1"],
                      Pd["w2"], Pd["b2"], emb, logit, rgb, pre_cm, n)
        assert torch.equal(logit, base[0]) and torch.equal(rgb, base[1]) and torch.equal(pre_cm, base[2]), variant

    # three routes to the same gradients: the matrix-core kernel (default where it has a tiling:
    # C in 8/16/32/64), the vector kernel recomputing the forward, and the vector kernel fed with the
    # forward's saved pre-activations
    def run_bwd(G, d_enc, pre):
        capi.call("shade_bwd", enc_cm, C, dv(dirs), d_img, Pd["w_h"], Pd["b_h"], Pd["w1"], Pd["b1"],
                  Pd["w2"], Pd["b2"], emb, dv(d_logit), dv(d_rgb), d_enc, G["w_h"], G["b_h"], G["w1"],
                  G["b1"], G["w2"], G["b2"], G["emb"] if with_emb else None, pre, n)

    for route in ("default", "valu", "valu_saved_pre"):
        capi.set_option("SHADE_BWD", 1 if route != "default" else 0)
        d_enc = torch.full((C, n), 7.0, device=dev)      # must be overwritten
        G = {k: torch.zeros_like(v) for k, v in Pd.items()}
        run_bwd(G, d_enc, pre_cm if route == "valu_saved_pre" else None)
        tag = lambda m, k: "%s [%s]: %s" % (k, route, m)
        torch.testing.assert_close(d_enc.t().cpu(), r_denc, rtol=1e-3,
                                   atol=1e-4 * float(r_denc.abs().max()),
                                   msg=lambda m: tag(m, "d_enc"))
        for k in ("w_h", "b_h", "w1", "b1", "w2", "b2") + (("emb",) if with_emb else ()):
            ref = r_g[k]
            torch.testing.assert_close(G[k].cpu(), ref, rtol=1e-3, atol=2e-4 * float(ref.abs().max()),
                                       msg=lambda m, k=k: tag(m, k))
        # accumulate semantics of the parameter gradients
        run_bwd(G, d_enc, pre_cm if route == "valu_saved_pre" else None)
        torch.testing.assert_close(G["w1"].cpu(), 2 * r_g["w1"], rtol=1e-3,
                                   atol=4e-4 * float(r_g["w1"].abs().max()),
                                   msg=lambda m: tag(m, "w1 accumulate"))


def test_shade_mfma_wide_rows(capi, dev):
    """Above 2^30 elements of encoding (C n 4 >= 2^32 bytes) the matrix-core kernels switch to 64-bit
    row offsets (WIDE): same results as the vector kernels, which never had the limit, on 34 M samples
    -- spot-checked where 32-bit offsets would have wrapped (the last quarter's rows, the far end)."""
    C, n, E = 32, (1 << 25) + (1 << 20) + 77, 5
    assert C * n >= 1 << 30
    g = torch.Generator(device=dev).manual_seed(5)
    enc_cm = (torch.randn(C, n, device=dev, generator=g) * 0.1).to(torch.float16).float()
    dirs = torch.randn(n, 3, device=dev, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    img = (torch.arange(n, device=dev) // 4099 % E).to(torch.int32)
    gc = torch.Generator().manual_seed(6)
    P = {"w_h": torch.randn(16, C, generator=gc) * 0.3, "b_h": torch.randn(16, generator=gc) * 0.1,
         "w1": torch.randn(64, 32, generator=gc) * 0.3, "b1": torch.randn(64, generator=gc) * 0.1,
         "w2": torch.randn(3, 64, generator=gc) * 0.3, "b2": torch.randn(3, generator=gc) * 0.1,
         "emb": torch.randn(E, 16, generator=gc) * 0.1}
    Pd = {k: v.to(dev).contiguous() for k, v in P.items()}
    d_logit = torch.randn(n, device=dev, generator=g)
    d_rgb = torch.randn(n, 3, device=dev, generator=g)
    out = {}
    for route in ("mfma", "vector"):
        capi.set_option("SHADE_FWD", 1 if route == "vector" else 0)
        capi.set_option("SHADE_BWD", 1 if route == "vector" else 0)
        logit = torch.full((n,), 7.0, device=dev)
        rgb = torch.full((n, 3), 7.0, device=dev)
        capi.call("shade_fwd", enc_cm, C, dirs, img, Pd["w_h"], Pd["b_h"], Pd["w1"], Pd["b1"], Pd["w2"],
                  Pd["b2"], Pd["emb"], logit, rgb, None, n)
        d_enc = torch.full((C, n), 7.0, device=dev)
        G = {k: torch.zeros_like(v) for k, v in Pd.items()}
        capi.call("shade_bwd", enc_cm, C, dirs, img, Pd["w_h"], Pd["b_h"], Pd["w1"], Pd["b1"], Pd["w2"],
                  Pd["b2"], Pd["emb"], d_logit, d_rgb, d_enc, G["w_h"], G["b_h"], G["w1"], G["b1"],
                  G["w2"], G["b2"], G["emb"], None, n)
        out[route] = (logit, rgb, d_enc, G)
    capi.set_option("SHADE_FWD", 0)
    capi.set_option("SHADE_BWD", 0)
    a, b = out["mfma"], out["vector"]
    torch.testing.assert_close(a[0], b[0], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a[1], b[1], rtol=1e-4, atol=1e-5)
    scale = float(b[2].abs().max())
    for rows in (slice(0, 2), slice(23, 25), slice(30, 32)):          # quarters 0, 2 / 3, 3
        for cols in (slice(0, 4096), slice(n // 2, n // 2 + 4096), slice(n - 4096, n)):
            d = (a[2][rows, cols] - b[2][rows, cols]).abs()
            assert float(d.median()) <= 1e-5 * scale and int((d > 1e-3 * scale).sum()) <= 4
    # Everywhere: equal up to the samples whose hidden pre-activation lies within rounding of zero
    # (two summation orders put the ReLU on different sides: ~1e-6 of 2.2e9 pre-activations, each
    # moving its sample's 32 gradient channels); a wrapped offset would corrupt whole rows instead.
    diff = (a[2] - b[2]).abs()
    bad = int((diff > 1e-3 * scale).sum())
    assert bad <= 2e-4 * diff.numel(), (bad, diff.numel())
    assert float(diff.mean()) <= 1e-6 * scale
    for k in ("w_h", "b_h", "w1", "b1", "w2", "b2", "emb"):
        ref = b[3][k]
        torch.testing.assert_close(a[3][k], ref, rtol=2e-3, atol=1e-3 * float(ref.abs().max()))
