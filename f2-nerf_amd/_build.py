"""Build recipes for the native parts of the package (no cmake: plain hipcc / g++ command lines).

  libf2nerf_hip.so   -- the gfx950 kernels behind the C ABI of include/f2nerf_hip.h (hipcc)
  libf2nerf_host.so  -- the LibTorch C++ operator surface (Hash3DAnchored, PtsSampler, SHShader,
                        Renderer, FlexOps, CustomOps) + its pybind11 module (g++ against torch headers)

Everything is built IN-TREE under f2-nerf_amd/lib so the binaries travel with a repo snapshot.
hipcc cross-compiles for gfx950 without a GPU present.
"""
import hashlib
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
LIB_DIR = os.path.join(PKG_DIR, "lib")
KERNEL_DIR = os.path.join(PKG_DIR, "csrc", "kernels")
HOST_DIR = os.path.join(PKG_DIR, "csrc", "host")
INCLUDE_DIR = os.path.join(REPO_DIR, "include")

HIP_LIB = os.path.join(LIB_DIR, "libf2nerf_hip.so")
HOST_LIB = os.path.join(LIB_DIR, "_f2nerf_host.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

# -ffp-contract=off: the kernels spell out every fused multiply-add (parity with the oracle).
# -fno-slp-vectorize: SLP pairs FMAs of two weight rows into v_pk_fma_f32, whose SGPR operand pairs
#   must then be assembled with s_mov/v_readlane -- 25 % more instructions in the fused MLP kernels.
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fno-slp-vectorize", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
]


# shade.hip keeps SLP: with the weights in VGPRs (LDS-staged backward) v_pk_fma_f32 halves the FMA
# issue count of an issue-bound kernel; the SGPR-pair problem only bites the small forward kernel.
HIP_FLAGS_DROP = {"shade.hip": ("-fno-slp-vectorize",)}


def _run(cmd, **kw):
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if res.returncode != 0:
        raise RuntimeError("build command failed:\n  %s\n%s" % (" ".join(cmd), res.stdout))
    return res.stdout


def _sources(d, exts):
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


def _stamp(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _up_to_date(target, stamp):
    sf = target + ".stamp"
    return os.path.exists(target) and os.path.exists(sf) and open(sf).read() == stamp


def _write_stamp(target, stamp):
    with open(target + ".stamp", "w") as f:
        f.write(stamp)


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 for every kernels/*.hip -> lib/libf2nerf_hip.so"""
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = _sources(KERNEL_DIR, (".hip",))
    deps = srcs + _sources(KERNEL_DIR, (".hiph",)) + _sources(INCLUDE_DIR, (".h",))
    stamp = _stamp(deps, " ".join(HIP_FLAGS) + repr(sorted(HIP_FLAGS_DROP.items())))
    if not force and _up_to_date(HIP_LIB, stamp):
        return HIP_LIB
    objs = []

    def compile_one(src):
        obj = os.path.join(LIB_DIR, os.path.basename(src) + ".o")
        flags = [f for f in HIP_FLAGS if f not in HIP_FLAGS_DROP.get(os.path.basename(src), ())]
        _run([HIPCC, *flags, "-I", INCLUDE_DIR, "-I", KERNEL_DIR, "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB, *objs])
    for o in objs:
        os.remove(o)
    _write_stamp(HIP_LIB, stamp)
    if verbose:
        print("built", HIP_LIB)
    return HIP_LIB


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce

    inc = []
    for p in ce.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", os.path.join(ROCM, "include")]
    inc += ["-isystem", sysconfig.get_paths()["include"]]
    try:
        import pybind11

        inc += ["-isystem", pybind11.get_include()]
    except ImportError:
        pass
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cflags = [
        "-O2", "-std=c++17", "-fPIC", "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi,
        "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_f2nerf_host",
        "-DTORCH_API_INCLUDE_EXTENSION_H", "-Wno-deprecated-declarations",
    ]
    ldflags = [
        "-L" + tlib, "-Wl,-rpath," + tlib, "-Wl,-rpath,$ORIGIN",
        "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
        "-L" + LIB_DIR, "-lf2nerf_hip", "-L" + os.path.join(ROCM, "lib"), "-lamdhip64",
    ]
    return inc, cflags, ldflags


def build_host(force=False, verbose=False):
    """g++ against the torch headers for host/*.cpp -> lib/_f2nerf_host.so (needs libf2nerf_hip.so)"""
    build_hip(force=False, verbose=verbose)
    srcs = _sources(HOST_DIR, (".cpp",))
    if not srcs:
        return None
    deps = srcs + _sources(HOST_DIR, (".hpp",)) + _sources(INCLUDE_DIR, (".h",))
    inc, cflags, ldflags = _torch_flags()
    stamp = _stamp(deps, " ".join(cflags + ldflags))
    if not force and _up_to_date(HOST_LIB, stamp):
        return HOST_LIB
    cxx = os.environ.get("CXX", "g++")

    def compile_one(src):
        obj = os.path.join(LIB_DIR, os.path.basename(src) + ".o")
        _run([cxx, *cflags, *inc, "-I", INCLUDE_DIR, "-I", HOST_DIR, "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    _run([cxx, "-shared", "-o", HOST_LIB, *objs, *ldflags])
    for o in objs:
        os.remove(o)
    _write_stamp(HOST_LIB, stamp)
    if verbose:
        print("built", HOST_LIB)
    return HOST_LIB


def build_all(force=False, verbose=False):
    build_hip(force=force, verbose=verbose)
    build_host(force=force, verbose=verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
