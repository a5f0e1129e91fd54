"""ctypes loader for libmyfa_gfx950.so (the C ABI declared in include/myfa.h).

There is deliberately no fallback: if the shared library is missing or does not
export the expected ABI, importing the package fails.  Build it with
``python flash-attention_amd/build.py`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The package loads exactly this file and reads no environment variable; tools/ and the variant
# tests load the tuning flavour (TUNING_LIB_PATH) themselves, through tools/tuned.py.
LIB_PATH = os.path.join(_HERE, "libmyfa_gfx950.so")
TUNING_LIB_PATH = os.path.join(_HERE, "libmyfa_gfx950_tuning.so")

ABI_VERSION = 3
MYFA_F32, MYFA_F16, MYFA_BF16 = 0, 1, 2

# every symbol include/myfa.h declares
EXPORTS = (
    "myfa_abi_version",
    "myfa_last_error",
    "myfa_device_info",
    "myfa_softmax_f32",
    "myfa_softmax",
    "myfa_softmax_nd",
    "myfa_softmax_nd_ex",
    "myfa_softmax_nd_ws_bytes",
    "myfa_softmax_nd_ws",
    "myfa_attn_fwd_f32",
    "myfa_attn_fwd_f32_ws_bytes",
    "myfa_attn_fwd_f32_ws",
    "myfa_attn_fwd_bhnd",
    "myfa_attn_fwd_bhnd_ex",
    "myfa_attn_fwd_bhnd_kernel",
    "myfa_attn_bwd_bhnd",
    "myfa_attn_bwd_bhnd_static_ws",
    "myfa_matmul_f32",
)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X kernels are not built. "
            "Run `python flash-attention_amd/build.py` (needs hipcc); there is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, f32, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
    lib.myfa_abi_version.argtypes = []
    lib.myfa_abi_version.restype = ci
    lib.myfa_last_error.argtypes = []
    lib.myfa_last_error.restype = ctypes.c_char_p
    lib.myfa_device_info.argtypes = [ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci),
                                     ctypes.c_char_p, ci]
    lib.myfa_device_info.restype = ci
    lib.myfa_softmax_f32.argtypes = [vp, vp, i64, i64, vp]
    lib.myfa_softmax_f32.restype = ci
    lib.myfa_softmax.argtypes = [vp, vp, i64, i64, ci, vp]
    lib.myfa_softmax.restype = ci
    lib.myfa_softmax_nd.argtypes = [vp, vp, i64, i64, i64, ci, vp]
    lib.myfa_softmax_nd.restype = ci
    lib.myfa_softmax_nd_ex.argtypes = [vp, vp, i64, i64, i64, ci, f32, vp, i64, ci, vp]
    lib.myfa_softmax_nd_ex.restype = ci
    lib.myfa_softmax_nd_ws_bytes.argtypes = [i64, i64, i64, ci]
    lib.myfa_softmax_nd_ws_bytes.restype = i64
    lib.myfa_softmax_nd_ws.argtypes = [vp, vp, i64, i64, i64, ci, f32, vp, i64, ci, vp, i64, vp]
    lib.myfa_softmax_nd_ws.restype = ci
    lib.myfa_attn_fwd_f32.argtypes = [vp, vp, vp, vp, i64, i64, f32, vp]
    lib.myfa_attn_fwd_f32.restype = ci
    lib.myfa_attn_fwd_f32_ws_bytes.argtypes = [i64, i64]
    lib.myfa_attn_fwd_f32_ws_bytes.restype = i64
    lib.myfa_attn_fwd_f32_ws.argtypes = [vp, vp, vp, vp, i64, i64, f32, vp, i64, vp]
    lib.myfa_attn_fwd_f32_ws.restype = ci
    lib.myfa_attn_fwd_bhnd.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, f32, ci, vp]
    lib.myfa_attn_fwd_bhnd.restype = ci
    lib.myfa_attn_fwd_bhnd_ex.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i64, f32, ci, ci, vp]
    lib.myfa_attn_fwd_bhnd_ex.restype = ci
    lib.myfa_attn_fwd_bhnd_kernel.argtypes = [i64, ci, ci]
    lib.myfa_attn_fwd_bhnd_kernel.restype = ctypes.c_char_p
    lib.myfa_attn_bwd_bhnd.argtypes = [vp] * 10 + [i64, i64, i64, i64, f32, ci, ci, vp]
    lib.myfa_attn_bwd_bhnd.restype = ci
    lib.myfa_attn_bwd_bhnd_static_ws.argtypes = [vp] * 9 + [i64, i64, i64, i64, f32, ci, ci, vp]
    lib.myfa_attn_bwd_bhnd_static_ws.restype = ci
    lib.myfa_matmul_f32.argtypes = [vp, vp, vp, i64, i64, i64, vp]
    lib.myfa_matmul_f32.restype = ci
    got = lib.myfa_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, this package expects {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def check(status):
    """Turns a nonzero C status into RuntimeError (what the reference's TORCH_CHECK
    failures surface as in Python)."""
    if status != 0:
        msg = lib.myfa_last_error().decode("utf-8", "replace")
        raise RuntimeError(msg or f"libmyfa error {status}")


def device_info(device=0):
    cu, lds, wave = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    arch = ctypes.create_string_buffer(64)
    check(lib.myfa_device_info(int(device), ctypes.byref(cu), ctypes.byref(lds), ctypes.byref(wave), arch, 64))
    return {"cu_count": cu.value, "lds_bytes": lds.value, "wave_size": wave.value,
            "arch": arch.value.decode()}
