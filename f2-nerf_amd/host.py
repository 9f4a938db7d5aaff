"""Loader for lib/_f2nerf_host.so: the LibTorch C++ operator surface (Hash3DAnchored, PtsSampler,
SHShader, Renderer, FlexOps, CustomOps) exposed through pybind11 for bench.py and the tests."""
import importlib.util
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
HOST_PATH = os.path.join(_PKG, "lib", "_f2nerf_host.so")
_mod = None


def module():
    global _mod
    if _mod is None:
        import torch  # noqa: F401  (libtorch must be loaded before the extension)

        if not os.path.exists(HOST_PATH):
            raise RuntimeError(
                "%s is not built; run __graft_entry__.build(). There is no Python fallback for the "
                "C++ host library." % HOST_PATH)
        spec = importlib.util.spec_from_file_location("_f2nerf_host", HOST_PATH)
        _mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_mod)
    return _mod
