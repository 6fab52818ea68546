"""ctypes binding of oracle/libpc_oracle.so (TEST INFRASTRUCTURE ONLY -- see pc_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpc_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "pc_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "pc_math.h")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "libpc_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def _host_threads(cap=16):
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


_lib = None
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(_host_threads())
        L.orc_rans_encode.argtypes = [_i32p, _i32p, C.c_int64, _i32p, C.c_int, _i32p, _i32p, C.c_int,
                                      _u8p, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_rans_decode.argtypes = [_u8p, C.c_int64, _i32p, C.c_int64, _i32p, C.c_int, _i32p, _i32p, C.c_int, _i32p]
        L.orc_pmf_to_quantized_cdf.argtypes = [_f32p, C.c_int, C.c_int, _u32p]
        L.orc_build_indexes.argtypes = [_f32p, C.c_int64, _f32p, C.c_int, C.c_float, _i32p]
        L.orc_quantile.argtypes = [_f32p, C.c_int64, C.c_float]
        L.orc_quantile.restype = C.c_float
        L.orc_mask_point_based_std.argtypes = [_f32p, C.c_int, C.c_int64, C.c_double, _f32p, C.c_void_p]
        L.orc_quantize.argtypes = [_f32p, C.c_void_p, C.c_int64, _i32p]
        L.orc_unary.argtypes = [_f32p, C.c_int64, C.c_int]
        L.orc_conv_nhwc.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    _f32p, _i32p, _i32p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int,
                                    _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_win_attention.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, _f32p]
        for f in (L.orc_rans_encode, L.orc_rans_decode, L.orc_pmf_to_quantized_cdf):
            f.restype = C.c_int
        _lib = L
    return _lib


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Tables:
    """CDF tables of one entropy model: cdf [n_cdf, stride] int32, length, offset."""

    def __init__(self, cdf, length, offset):
        self.cdf = _c(cdf, np.int32)
        self.length = _c(length, np.int32).reshape(-1)
        self.offset = _c(offset, np.int32).reshape(-1)


def rans_encode(sym, idx, t: Tables) -> bytes:
    sym = _c(sym, np.int32).reshape(-1)
    idx = _c(idx, np.int32).reshape(-1)
    cap = 4 * (sym.size * 10 + 16) + 64
    out = np.empty(cap, np.uint8)
    n = C.c_int64(0)
    rc = lib().orc_rans_encode(sym, idx, sym.size, t.cdf, t.cdf.shape[1], t.length, t.offset, t.cdf.shape[0],
                               out, cap, C.byref(n))
    if rc:
        raise ValueError(f"orc_rans_encode rc={rc}")
    return out[: n.value].tobytes()


def rans_decode(data: bytes, idx, t: Tables) -> np.ndarray:
    idx = _c(idx, np.int32).reshape(-1)
    buf = np.frombuffer(data, np.uint8).copy()
    out = np.empty(idx.size, np.int32)
    rc = lib().orc_rans_decode(buf, buf.size, idx, idx.size, t.cdf, t.cdf.shape[1], t.length, t.offset,
                               t.cdf.shape[0], out)
    if rc:
        raise ValueError(f"orc_rans_decode rc={rc}")
    return out


def pmf_to_quantized_cdf(pmf, precision=16) -> np.ndarray:
    pmf = _c(pmf, np.float32).reshape(-1)
    cdf = np.zeros(pmf.size + 1, np.uint32)
    rc = lib().orc_pmf_to_quantized_cdf(pmf, pmf.size, precision, cdf)
    if rc:
        raise ValueError(f"orc_pmf_to_quantized_cdf rc={rc}")
    return cdf


def build_indexes(scale, table, bound=0.11) -> np.ndarray:
    s = _c(scale, np.float32)
    out = np.empty(s.shape, np.int32)
    lib().orc_build_indexes(s.reshape(-1), s.size, _c(table, np.float32), len(table), np.float32(bound), out.reshape(-1))
    return out


def quantile(v, q) -> np.float32:
    v = _c(v, np.float32).reshape(-1)
    return np.float32(lib().orc_quantile(v, v.size, np.float32(q)))


def mask_point_based_std(scale, pr) -> np.ndarray:
    """scale: [B, ...]; one quantile per leading-dim entry."""
    s = _c(scale, np.float32)
    out = np.empty(s.shape, np.float32)
    lib().orc_mask_point_based_std(s.reshape(-1), s.shape[0], s[0].size, float(pr), out.reshape(-1), None)
    return out


def quantize(x, mu=None) -> np.ndarray:
    x = _c(x, np.float32)
    out = np.empty(x.shape, np.int32)
    if mu is None:
        lib().orc_quantize(x.reshape(-1), None, x.size, out.reshape(-1))
    else:
        m = _c(mu, np.float32)
        lib().orc_quantize(x.reshape(-1), m.ctypes.data_as(C.c_void_p), x.size, out.reshape(-1))
    return out


UNARY = {"gelu": 0, "tanh": 1, "sigmoid": 2, "rsqrt": 3, "sqrt": 4, "exp": 5, "erf": 6}


def unary(x, op) -> np.ndarray:
    y = np.array(x, dtype=np.float32, order="C", copy=True)
    lib().orc_unary(y.reshape(-1), y.size, UNARY[op])
    return y


def conv_nhwc(x, w, taps, stride, Ho, Wo, out=None, out_hw=None, ostride=(1, 1), ooff=(0, 0), square=False):
    """x [B,H,W,Cin]; w [T,Cin,Cout]; taps list of (dy,dx).  Raw fmaf-chain accumulations, no bias."""
    x = _c(x, np.float32)
    w = _c(w, np.float32)
    B, H, W, Cin = x.shape
    T, Cin2, Cout = w.shape
    assert Cin2 == Cin and T == len(taps)
    dy = _c([t[0] for t in taps], np.int32)
    dx = _c([t[1] for t in taps], np.int32)
    if out is None:
        oh, ow = out_hw if out_hw else (Ho, Wo)
        out = np.zeros((B, oh, ow, Cout), np.float32)
    lib().orc_conv_nhwc(x.reshape(-1), B, H, W, Cin, Cin, w.reshape(-1), dy, dx, T, stride, Ho, Wo, Cout,
                        out.reshape(-1), out.shape[1], out.shape[2], ostride[0], ooff[0], ostride[1], ooff[1],
                        out.shape[3], 1 if square else 0)
    return out


def win_attention(qkv, bias, heads, ws, shift, scale):
    qkv = _c(qkv, np.float32)
    B, H, W, C3 = qkv.shape
    Cc = C3 // 3
    out = np.empty((B, H, W, Cc), np.float32)
    lib().orc_win_attention(qkv.reshape(-1), _c(bias, np.float32).reshape(-1), B, H, W, Cc, heads, ws, shift,
                            np.float32(scale), out.reshape(-1))
    return out
