"""ctypes binding of oracle/libf2n_oracle.so (the C restatement of the reference's 14 CUDA kernels)
with torch-CPU tensor wrappers.  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
import ctypes
import os
import subprocess

import torch

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libf2n_oracle.so")
_lib = None

c_p, c_i, c_i64, c_u32, c_f = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_float)

_SIGS = {
    "f2no_cast_f32_to_f16": (None, [c_p, c_p, c_i64]),
    "f2no_cast_f16_to_f32": (None, [c_p, c_p, c_i64]),
    "f2no_level_mul": (None, [c_i, c_p]),
    "f2no_hash_fwd": (None, [c_p] * 7 + [c_i64, c_i, c_i, c_u32, c_i64]),
    "f2no_hash_bwd": (None, [c_p] * 8 + [c_i64, c_i, c_i, c_u32, c_i64, c_f, c_i]),
    "f2no_div_inplace": (None, [c_p, c_i64, c_f]),
    "f2no_sh_encode": (None, [c_p, c_p, c_i64, c_i]),
    "f2no_seg_sum_fwd": (None, [c_p, c_p, c_p, c_i]),
    "f2no_seg_sum_bwd": (None, [c_p, c_p, c_p, c_i]),
    "f2no_seg_sum_vec_fwd": (None, [c_p, c_p, c_p, c_i, c_i]),
    "f2no_seg_sum_vec_bwd": (None, [c_p, c_p, c_p, c_i, c_i]),
    "f2no_seg_scan_fwd": (None, [c_p, c_p, c_p, c_i, c_i]),
    "f2no_seg_scan_bwd": (None, [c_p, c_p, c_p, c_i, c_i]),
    "f2no_weight_var_fwd": (None, [c_p, c_p, c_p, c_i]),
    "f2no_weight_var_bwd": (None, [c_p, c_p, c_p, c_p, c_i]),
    "f2no_scatter_idx": (None, [c_p, c_p, c_p, c_i]),
    "f2no_scatter_add_fwd": (None, [c_p, c_p, c_p, c_p, c_i64, c_i]),
    "f2no_scatter_add_bwd": (None, [c_p, c_p, c_p, c_i64, c_i, c_i]),
    "f2no_num_threads": (c_i, []),
    "f2no_set_num_threads": (None, [c_i]),
}


def build(force=False):
    src = os.path.join(_DIR, "f2n_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _DIR, "-B", "libf2n_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(t):
    assert t.dtype == torch.float32 and t.device.type == "cpu"
    return t.contiguous()


def _i32(t):
    assert t.dtype == torch.int32 and t.device.type == "cpu"
    return t.contiguous()


def num_threads():
    return lib().f2no_num_threads()


def set_num_threads(n):
    lib().f2no_set_num_threads(int(n))


def level_mul(L):
    out = torch.empty(L, dtype=torch.float32)
    lib().f2no_level_mul(L, _p(out))
    return out


def cast_f16(table_f32):
    """f32 tensor -> int16 tensor holding the f16 bit patterns (RNE)."""
    x = _f32(table_f32)
    out = torch.empty(x.shape, dtype=torch.int16)
    lib().f2no_cast_f32_to_f16(_p(x), _p(out), x.numel())
    return out


def hash_fwd(pts, table_f16, primes, bias, mul, L, F, T, level_stride, want_idx=False):
    pts = _f32(pts)
    n = pts.shape[0]
    out = torch.empty(n, L * F, dtype=torch.float32)
    idx = torch.empty(n, L, 8, dtype=torch.int32) if want_idx else None
    lib().f2no_hash_fwd(_p(pts), _p(table_f16.contiguous()), _p(_i32(primes)), _p(_f32(bias)),
                        _p(_f32(mul)), _p(out), _p(idx), n, L, F, T, level_stride)
    return (out, idx) if want_idx else out


def hash_bwd(pts, table_f16, primes, bias, mul, grad_out, table_numel, L, F, T, level_stride,
             grad_scale=128.0, need_pts_grad=False, parallel=False):
    pts = _f32(pts)
    n = pts.shape[0]
    g = _f32(grad_out)
    tg = torch.zeros(table_numel, dtype=torch.float32)
    pg = torch.zeros(n, 3, dtype=torch.float32) if need_pts_grad else None
    lib().f2no_hash_bwd(_p(pts), _p(table_f16.contiguous()), _p(_i32(primes)), _p(_f32(bias)),
                        _p(_f32(mul)), _p(g), _p(tg), _p(pg), n, L, F, T, level_stride,
                        grad_scale, int(parallel))
    lib().f2no_div_inplace(_p(tg), tg.numel(), grad_scale)
    return tg, pg


def sh_encode(dirs, degree=4):
    d = _f32(dirs)
    out = torch.empty(d.shape[0], degree * degree, dtype=torch.float32)
    lib().f2no_sh_encode(_p(d), _p(out), d.shape[0], degree)
    return out


def seg_sum_fwd(val, idx):
    val, idx = _f32(val), _i32(idx)
    R = idx.shape[0]
    if val.dim() == 1:
        out = torch.empty(R, dtype=torch.float32)
        lib().f2no_seg_sum_fwd(_p(val), _p(idx), _p(out), R)
    else:
        out = torch.empty(R, val.shape[1], dtype=torch.float32)
        lib().f2no_seg_sum_vec_fwd(_p(val), _p(idx), _p(out), R, val.shape[1])
    return out


def seg_sum_bwd(dsum, idx, n_all):
    dsum, idx = _f32(dsum), _i32(idx)
    R = idx.shape[0]
    if dsum.dim() == 1:
        out = torch.zeros(n_all, dtype=torch.float32)
        lib().f2no_seg_sum_bwd(_p(dsum), _p(idx), _p(out), R)
    else:
        out = torch.zeros(n_all, dsum.shape[1], dtype=torch.float32)
        lib().f2no_seg_sum_vec_bwd(_p(dsum), _p(idx), _p(out), R, dsum.shape[1])
    return out


def seg_scan_fwd(val, idx, include_this):
    val, idx = _f32(val), _i32(idx)
    out = torch.zeros_like(val)
    lib().f2no_seg_scan_fwd(_p(val), _p(idx), _p(out), idx.shape[0], int(include_this))
    return out


def seg_scan_bwd(dsum, idx, include_this):
    dsum, idx = _f32(dsum), _i32(idx)
    out = torch.zeros_like(dsum)
    lib().f2no_seg_scan_bwd(_p(dsum), _p(idx), _p(out), idx.shape[0], int(include_this))
    return out


def weight_var_fwd(w, idx):
    w, idx = _f32(w), _i32(idx)
    out = torch.empty(idx.shape[0], dtype=torch.float32)
    lib().f2no_weight_var_fwd(_p(w), _p(idx), _p(out), idx.shape[0])
    return out


def weight_var_bwd(w, idx, dvars):
    w, idx, dvars = _f32(w), _i32(idx), _f32(dvars)
    out = torch.zeros_like(w)
    lib().f2no_weight_var_bwd(_p(w), _p(idx), _p(dvars), _p(out), idx.shape[0])
    return out


def scatter_idx(n_all, idx, emb_idx):
    idx, emb_idx = _i32(idx), _i32(emb_idx)
    out = torch.zeros(n_all, dtype=torch.int32)
    lib().f2no_scatter_idx(_p(idx), _p(emb_idx), _p(out), idx.shape[0])
    return out


def scatter_add_fwd(emb, sidx, to_add):
    emb, sidx, to_add = _f32(emb), _i32(sidx), _f32(to_add)
    out = torch.empty_like(to_add)
    lib().f2no_scatter_add_fwd(_p(emb), _p(sidx), _p(to_add), _p(out), to_add.shape[0],
                               to_add.shape[1])
    return out


def scatter_add_bwd(sidx, dsum, n_emb):
    sidx, dsum = _i32(sidx), _f32(dsum)
    out = torch.zeros(n_emb, dsum.shape[1], dtype=torch.float32)
    lib().f2no_scatter_add_bwd(_p(sidx), _p(dsum), _p(out), dsum.shape[0], n_emb, dsum.shape[1])
    return out
