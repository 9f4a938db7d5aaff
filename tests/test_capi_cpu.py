"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares, argument validation answers without touching a GPU, the C++ host library mirrors
the reference's module/parameter layout, and the product path refuses to run without the GPU."""
import ctypes
import importlib
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(capi):
    decls = capi.parse_header()
    # every f2n_ name that appears in the header as a declaration must have been parsed ...
    text = open(capi.HEADER).read()
    names_in_header = set(re.findall(r"\b(f2n_[a-z0-9_]+)\s*\(", text))
    assert names_in_header == set(decls), names_in_header ^ set(decls)
    assert len(decls) >= 25
    # ... and exported by the shared library with C linkage
    cdll = ctypes.CDLL(capi.LIB_PATH)
    for name in decls:
        assert hasattr(cdll, name), name
    lib = capi.lib()
    assert lib.cdll.f2n_abi_version() == 2
    assert lib.status_string(0) == "ok" and "invalid" in lib.status_string(-1)


def test_documented_workspace_cap_is_the_real_one(capi):
    """the ABI header's "at most N GiB" for f2n_hash_bwd_workspace_bytes must be the cap
    the library applies (an integrator budgets memory from the header).  A host-side planning call:
    no kernel runs."""
    text = open(capi.HEADER).read()
    m = re.search(r"recommended workspace size \(at most (\d+) GiB\)", text)
    assert m, "the header must state the cap"
    cap = int(m.group(1)) << 30
    fn = capi.lib().cdll.f2n_hash_bwd_workspace_bytes
    huge = fn(1 << 28, 16, 8, 1 << 22)          # far more than one round's worth: capped
    assert 0.8 * cap <= huge <= cap + 4096
    small = fn(1 << 17, 16, 2, 1 << 19)
    assert 0 < small < cap // 16
    assert fn(1000, 16, 2, 1 << 19) == 0         # below 65536 points: not applicable


def test_header_cites_reference_for_every_entry_point(capi):
    text = open(capi.HEADER).read()
    for name in capi.parse_header():
        if name in ("f2n_abi_version", "f2n_status_string", "f2n_set_option", "f2n_get_option"):
            continue
        pos = text.index(name + "(")
        block = text[max(0, pos - 2500):pos]
        assert re.search(r"src/[\w/]+\.(cu|cpp|hpp):\d+", block), "no reference citation near " + name


def test_route_options_are_explicit_state_not_environment(capi, monkeypatch):
    """Kernel routes are chosen through f2n_set_option, never through getenv at launch time
    (VERDICT r1 item 8): the library does not import getenv at all, options default to 0, reject
    unknown keys / values and return the previous value."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", capi.LIB_PATH], capture_output=True,
                          text=True).stdout
    assert "getenv" not in syms
    keys = capi.option_keys()
    assert set(keys) >= {"SHADE_FWD", "SHADE_BWD", "SHADE_VARIANT", "RAYTILE", "HASH_BWD"}
    c = capi.lib().cdll
    monkeypatch.setenv("F2N_HASH_BWD", "atomic")          # the old switch must be inert
    for k in keys.values():
        assert c.f2n_get_option(k) == 0
    assert capi.set_option("HASH_BWD", 2) == 0 and c.f2n_get_option(keys["HASH_BWD"]) == 2
    with capi.option("RAYTILE", 16):
        assert c.f2n_get_option(keys["RAYTILE"]) == 16
    assert c.f2n_get_option(keys["RAYTILE"]) == 0
    assert c.f2n_set_option(keys["RAYTILE"], 17) == -1 and c.f2n_set_option(99, 0) == -1
    assert c.f2n_get_option(-1) == -1
    assert capi.set_option("HASH_BWD", 0) == 2


def test_argument_validation_without_gpu(capi):
    """Invalid calls are rejected before any HIP work, so this is safe on a CPU-only host."""
    c = capi.lib().cdll
    assert c.f2n_hash_fwd(None, None, None, None, None, None, 32, 1, None, 10, 16, 2, 1 << 19,
                          1 << 19, None) == -1
    assert c.f2n_sh_encode(None, None, 10, 4, None) == -1
    assert c.f2n_sample_rays(None, None, None, None, None, None, None, None, -1, 8, 0.1, None) == -1
    assert c.f2n_seg_sum_fwd(None, None, None, 0, None) == -1      # idx required even when empty
    assert c.f2n_scatter_idx(None, None, None, -3, None) == -1
    with pytest.raises(capi.F2NError):
        capi.call("contract_fwd", None, None, 5, stream=0)


def test_host_module_mirrors_reference_layout(pkg):
    H = pkg.load_host()
    H.manual_seed(2022)
    r = H.Renderer(4, device="cpu")            # reference defaults: L=16, F=2, pool 2^19*16
    p = r.named_parameters()
    want = {
        "app_emb": (4, 16),
        "scene_field.feat_pool": ((1 << 19) * 16, 2),
        "scene_field.prim_pool": (16, 3),
        "scene_field.bias_pool": (16, 3),
        "scene_field.mlp.weight": (16, 32),
        "scene_field.mlp.bias": (16,),
        "shader.mlp.0.weight": (64, 32),
        "shader.mlp.0.bias": (64,),
        "shader.mlp.2.weight": (3, 64),
        "shader.mlp.2.bias": (3,),
    }
    assert {k: tuple(v.shape) for k, v in p.items()} == want   # checkpoint-compatible names/shapes
    f = r.scene_field
    assert f.pool_size == (1 << 19) * 16 and f.local_size == 1 << 19 and f.level_stride == 1 << 19
    assert p["scene_field.prim_pool"].dtype == torch.int32
    pr = p["scene_field.prim_pool"]
    assert int(pr.min()) >= 1 << 28 and int(pr.max()) < 1 << 30
    bias = p["scene_field.bias_pool"]
    assert float(bias.min()) >= 100 and float(bias.max()) < 1100
    fp = p["scene_field.feat_pool"]
    assert float(fp.min()) >= -1.0001e-4 and float(fp.max()) <= -0.7999e-4     # (U*0.2-1)*1e-4
    from oracle import kernels as K
    assert torch.equal(f.level_mul, K.level_mul(16))
    assert H.MAX_SAMPLE_PER_RAY == 1024
    assert r.make_adam(1e-2).n_groups() == 4   # table | hash mlp | shader mlp | app_emb


def test_no_cpu_fallback(pkg):
    H = pkg.load_host()
    r = H.Renderer(2, n_levels=2, log2_table=8, max_samples=8, device="cpu")
    o, d = torch.zeros(3, 3), torch.ones(3, 3)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        r.render(o, d)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        H.flex_sum(torch.ones(4), torch.tensor([[0, 4]], dtype=torch.int32))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        H.SHShader("cpu").encode(d)
    # the product tree never reaches into the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "f2-nerf_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".hiph")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in src.replace("the oracle", "").replace("with the oracle", ""), fn


def test_sharding_helpers(pkg):
    sh = pkg.sharding
    for n, w in ((640000, 8), (10, 3), (7, 8), (4096, 8)):
        spans = [sh.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert sh.shard_range(4096, 3, 8) == (1536, 2048)          # config C4: 512 rays per GPU
    views = {sh.view_for(s, r, 8, 50) for s in range(3) for r in range(8)}
    assert len(views) == 24
