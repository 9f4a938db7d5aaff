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


# ---- the reference's whole render path: its host sources unmodified + the C-ABI binding ---------

OUT_HOST = os.path.join(HERE, "_ref", "_f2nerf_ref_host.so")
REF_HOST_SOURCES = ["src/renderer.cpp", "src/hash_3d_anchored.cpp", "src/sh_shader.cpp",
                    "src/points_sampler.cpp", "src/rays.cpp", "src/CustomOps/CustomOps.cpp",
                    "src/CustomOps/FlexOps.cpp", "src/CustomOps/Scatter.cpp"]


def build_host(force=False):
    """oracle/_ref/_f2nerf_ref_host.so = the reference's host translation units of the render path
    (REF_HOST_SOURCES, compiled where they lie, against the reference's own headers) + the symbols
    its five .cu files would define, supplied by oracle/ref_cuda_side.cpp through
    include/f2nerf_hip.h (libf2nerf_hip.so) + the pybind shim.  No stand-in for anything the image
    lacks: the .cu files themselves are simply not part of this build."""
    if not os.path.isdir(REF):
        return None
    ours = [os.path.join(HERE, "ref_cuda_side.cpp"), os.path.join(HERE, "ref_host_shim.cpp")]
    srcs = [os.path.join(REF, s) for s in REF_HOST_SOURCES] + ours
    hip_lib_dir = os.path.join(os.path.dirname(HERE), "f2-nerf_amd", "lib")
    dep = srcs + [os.path.join(os.path.dirname(HERE), "include", "f2nerf_hip.h")]
    if not force and os.path.exists(OUT_HOST) and all(
            os.path.getmtime(OUT_HOST) >= os.path.getmtime(s) for s in dep):
        return OUT_HOST
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(os.path.dirname(OUT_HOST), exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = []
    for p in ce.include_paths() + [os.path.join(rocm, "include"), sysconfig.get_paths()["include"],
                                    pybind11.get_include()]:
        inc += ["-isystem", p]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cflags = ["-O2", "-std=c++17", "-fPIC", "-w",
              "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
              "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_f2nerf_ref_host",
              "-I", os.path.join(REF, "src"), "-I", os.path.join(REF, "External", "eigen-3.4.0"),
              "-I", os.path.join(os.path.dirname(HERE), "include"), *inc]
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    with tempfile.TemporaryDirectory() as tmp:
        def compile_one(i_src):
            i, src = i_src
            obj = os.path.join(tmp, "%d_%s.o" % (i, os.path.basename(src)))
            res = subprocess.run(["g++", *cflags, "-c", src, "-o", obj], stdout=subprocess.PIPE,
                                 stderr=subprocess.STDOUT, text=True)
            if res.returncode != 0:
                raise RuntimeError("reference render-path build failed (%s):\n%s" % (src, res.stdout[-4000:]))
            return obj

        with ThreadPoolExecutor(max_workers=5) as ex:
            objs = list(ex.map(compile_one, enumerate(srcs)))
        cmd = ["g++", "-shared", "-o", OUT_HOST, *objs,
               "-L" + tlib, "-Wl,-rpath," + tlib, "-L" + hip_lib_dir,
               "-Wl,-rpath,$ORIGIN/../../f2-nerf_amd/lib", "-lf2nerf_hip",
               "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
               "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python"]
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("reference render-path link failed:\n" + res.stdout[-4000:])
    return OUT_HOST


def load_host():
    """Import _f2nerf_ref_host if it was built (None otherwise).  Never in a process that also loads
    this repository's _f2nerf_host.so: both register TORCH_LIBRARY(dec_hash3d_anchored)."""
    if not os.path.exists(OUT_HOST):
        return None
    import importlib.util
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location("_f2nerf_ref_host", OUT_HOST)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


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
    print(build_host(force="--force" in sys.argv))
