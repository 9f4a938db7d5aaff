"""ctypes view of the C ABI in include/f2nerf_hip.h (libf2nerf_hip.so).

The signatures are parsed from the header itself, so the binding cannot drift from the boundary.
There is NO fallback: if the library is missing or a call returns a non-zero status this raises.
PyTorch is used by callers only as the owner of device memory (tensor.data_ptr()) and streams.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_PKG), "include", "f2nerf_hip.h")
LIB_PATH = os.path.join(_PKG, "lib", "libf2nerf_hip.so")

_CTYPES = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "uint32_t": ctypes.c_uint32,
    "float": ctypes.c_float,
}


class F2NError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every function the header declares."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    decls = {}
    for m in re.finditer(r"\b(int64_t|int|const char \*)\s*(f2n_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    params.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    toks = a.replace("const ", "").split()
                    params.append((_CTYPES[toks[0]], toks[-1]))
        restype = ctypes.c_char_p if "char" in ret else (
            ctypes.c_int64 if ret == "int64_t" else ctypes.c_int)
        decls[name] = (restype, params)
    return decls


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise F2NError(
                "libf2nerf_hip.so is not built (%s). Run `python -c \"import __graft_entry__ as g; "
                "g.build()\"` -- there is no non-HIP fallback." % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.decls = parse_header()
        for name, (ret, params) in self.decls.items():
            fn = getattr(self.cdll, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = [t for t, _ in params]
        if self.cdll.f2n_abi_version() != 2:
            raise F2NError("ABI version mismatch")

    def status_string(self, st):
        return self.cdll.f2n_status_string(st).decode()


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def option_keys(path=HEADER):
    """{'SHADE_FWD': 0, ...} from the header's #define F2N_OPT_* lines."""
    keys = {}
    for m in re.finditer(r"#define\s+F2N_OPT_(\w+)\s+(\d+)", open(path).read()):
        if m.group(1) != "COUNT":
            keys[m.group(1)] = int(m.group(2))
    return keys


def set_option(name, value):
    """f2n_set_option by name (e.g. set_option("SHADE_BWD", 1)); returns the previous value."""
    L = lib()
    prev = L.cdll.f2n_set_option(option_keys()[name.upper()], int(value))
    if prev < 0:
        raise F2NError("f2n_set_option(%s, %r) rejected" % (name, value))
    return prev


class option:
    """Context manager: `with capi.option("HASH_BWD", 2): ...` restores the previous route on exit."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


def call(name, *args, stream=None):
    """Call f2n_<name>(*args, stream) with tensors turned into device pointers; raise on error."""
    L = lib()
    full = name if name.startswith("f2n_") else "f2n_" + name
    fn = getattr(L.cdll, full)
    if stream is None:
        stream = current_stream()
    st = fn(*[_ptr(a) for a in args], stream)
    if st != 0:
        raise F2NError("%s failed: %s (%d)" % (full, L.status_string(st), st))
