#!/usr/bin/env python3
"""Builds libmyfa_gfx950.so: hipcc, gfx950 only, in-tree.

Takes the place of the reference's setup.py CUDAExtension build
(setup.py:107-143 of the reference: nvcc, sm_75, ninja).  There is no
torch C++ extension here -- the library is a plain C-ABI shared object that the
Python host layer loads with ctypes -- so the build is one hipcc invocation per
translation unit and a link.  Objects are cached under csrc/.obj keyed on the
source and header mtimes.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(CSRC, ".obj")
PKG = os.path.join(HERE, "my_flash_attn_cuda")
LIB = os.path.join(PKG, "libmyfa_gfx950.so")

SOURCES = ["c_abi.hip", "softmax_rows.hip", "softmax_strided.hip", "attn_fwd_f32.hip", "attn_fwd_mfma.hip", "attn_bwd_mfma.hip", "matmul_f32.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + INCLUDE, "-I" + CSRC,
         "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]
# The backward kernels place every vector instruction by hand between MFMAs; the SLP vectoriser
# regroups them into packed float32 pairs with register shuffles around them.
PER_SOURCE_FLAGS = {"attn_bwd_mfma.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, extra):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    stamp = max(os.path.getmtime(path), _newest_header(), os.path.getmtime(__file__))
    if os.path.exists(obj) and os.path.getmtime(obj) >= stamp:
        return obj, False
    subprocess.check_call([_hipcc(), *FLAGS, *PER_SOURCE_FLAGS.get(src, []), *extra, "-c", path, "-o", obj])
    return obj, True


def build_library(force=False, verbose=True, extra_flags=()):
    """Compile every HIP source for gfx950 and link the shared library."""
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _compile(s, list(extra_flags)), SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB):
        subprocess.check_call([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs])
        if verbose:
            print(f"[myfa] linked {LIB}")
    elif verbose:
        print(f"[myfa] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
