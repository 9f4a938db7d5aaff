"""Builds oracle/_ref/_f2nerf_ref.so from the reference's own pure-ATen host sources where they lie
under /root/reference (plus oracle/ref_shim.cpp).  Run in the authoring container only; the built
.so travels to the GPU box with the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored).
No reference source is copied; nothing the image lacks is stubbed.  See ref_shim.cpp for scope."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("F2N_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref", "_f2nerf_ref.so")
REF_SOURCES = ["src/points_sampler.cpp", "src/rays.cpp", "src/CustomOps/CustomOps.cpp"]


def build(force=False):
    if not os.path.isdir(REF):
        return None   # GPU box: use the prebuilt file if it travelled
    srcs = [os.path.join(REF, s) for s in REF_SOURCES] + [os.path.join(HERE, "ref_shim.cpp")]
    if not force and os.path.exists(OUT) and all(
            os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    inc = []
    for p in ce.include_paths() + [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include"),
                                    sysconfig.get_paths()["include"], pybind11.get_include()]:
        inc += ["-isystem", p]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_f2nerf_ref",
           "-I", os.path.join(REF, "src"), "-I", os.path.join(REF, "External", "eigen-3.4.0"), *inc,
           *srcs, "-o", OUT, "-L" + tlib, "-Wl,-rpath," + tlib,
           "-lc10", "-ltorch_cpu", "-ltorch", "-ltorch_python"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("reference host build failed:\n" + res.stdout[-3000:])
    return OUT


def load():
    """Import the module if the .so exists (None otherwise)."""
    if not os.path.exists(OUT):
        return None
    import importlib.util
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location("_f2nerf_ref", OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
