"""ctypes binding for the C oracle (``oracle/oracle.c`` -> ``oracle/liboracle.so``).

Test infrastructure only; see the header of ``oracle/oracle.c``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    """Compile liboracle.so with gcc (seconds).  Returns its path."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        i64 = ctypes.c_int64
        for name in ("oracle_softmax_f64acc", "oracle_softmax_f32"):
            f = getattr(L, name)
            f.argtypes = [fp, fp, i64, i64]
            f.restype = None
        for name in ("oracle_attention_f64acc", "oracle_attention_f32"):
            f = getattr(L, name)
            f.argtypes = [fp, fp, fp, fp, i64, i64, i64, i64, ctypes.c_float]
            f.restype = None
        L.oracle_attention_causal_f64acc.argtypes = [fp, fp, fp, fp, ctypes.POINTER(i64), i64, i64, i64, i64, ctypes.c_float]
        L.oracle_attention_causal_f64acc.restype = None
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        L.oracle_abi_version.restype = ctypes.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def softmax_rows(x, f32_arith=False):
    x = _f32c(x)
    out = np.empty_like(x)
    fn = lib().oracle_softmax_f32 if f32_arith else lib().oracle_softmax_f64acc
    fn(_fp(x), _fp(out), x.shape[0], x.shape[1])
    return out


def attention(q, k, v, scale=1.0, f32_arith=False, causal_pos=None):
    """q: (..., Nq, d); k, v: (..., N, d) float32, k NOT transposed.  Returns float32 (..., Nq, d).

    Nq may differ from N (a sample of query rows against all keys).  causal_pos: the Nq
    sequence positions of the query rows; when given, row i sees keys 0..causal_pos[i]."""
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    assert k.shape == v.shape and q.ndim == k.ndim >= 2
    assert q.shape[:-2] == k.shape[:-2] and q.shape[-1] == k.shape[-1]
    nq, d = q.shape[-2:]
    n = k.shape[-2]
    heads = int(np.prod(q.shape[:-2], dtype=np.int64)) if q.ndim > 2 else 1
    out = np.empty_like(q)
    if causal_pos is not None:
        assert not f32_arith
        pos = np.ascontiguousarray(np.asarray(causal_pos, dtype=np.int64))
        assert pos.shape == (nq,)
        lib().oracle_attention_causal_f64acc(_fp(q), _fp(k), _fp(v), _fp(out),
                                             pos.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                             heads, nq, n, d, float(scale))
        return out
    fn = lib().oracle_attention_f32 if f32_arith else lib().oracle_attention_f64acc
    fn(_fp(q), _fp(k), _fp(v), _fp(out), heads, nq, n, d, float(scale))
    return out


def max_threads():
    return int(lib().oracle_max_threads())


def set_threads(n):
    lib().oracle_set_threads(int(n))
