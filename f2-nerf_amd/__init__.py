"""f2-nerf_amd -- MI355X (gfx950) native rendering hot path of F2-NeRF.

What is here is only what the hot path needs:
  csrc/kernels  hand-written HIP kernels + the C ABI (include/f2nerf_hip.h)  -> lib/libf2nerf_hip.so
  csrc/host     the LibTorch C++ operator surface of the reference           -> lib/_f2nerf_host.so
  capi.py       ctypes view of the C ABI (tests call the kernels through it)
  host.py       loader for the pybind11 module that exposes the C++ classes to bench/tests
  sharding.py   ray sharding over ranks + the one all-reduce of the path (loss / PSNR scalar)

The directory name carries a hyphen (it is the reference's name); import it with
importlib.import_module("f2-nerf_amd").
"""
from . import _build, capi, sharding  # noqa: F401

__all__ = ["_build", "capi", "sharding", "load_host"]


def load_host():
    """Import the pybind11 module of the C++/LibTorch host library (raises if it is not built)."""
    from . import host

    return host.module()
