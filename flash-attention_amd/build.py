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
# The same sources with -DMYFA_TUNING: launch knobs settable at run time through two extra
# exported functions (csrc/tuning.h).  For tools/ and the tests that compare kernel variants;
# the package never loads it.
LIB_TUNING = os.path.join(PKG, "libmyfa_gfx950_tuning.so")

SOURCES = ["c_abi.hip", "softmax_rows.hip", "softmax_strided.hip", "attn_fwd_f32.hip", "attn_fwd_dispatch.hip", "attn_fwd_mfma.hip",
           "attn_fwd_mfma16.hip", "attn_fwd_stream.hip", "attn_fwd_mfma256.hip", "attn_bwd_mfma.hip", "attn_bwd_wide.hip", "attn_bwd_f32.hip", "matmul_f32.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-I" + INCLUDE, "-I" + CSRC,
         "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]
# The backward kernels place every vector instruction by hand between MFMAs; the SLP vectoriser
# regroups them into packed float32 pairs with register shuffles around them.
PER_SOURCE_FLAGS = {"attn_bwd_mfma.hip": ["-fno-slp-vectorize"], "attn_fwd_mfma16.hip": ["-fno-slp-vectorize"],
                    "attn_fwd_stream.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, extra, objdir=OBJ):
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    cmd = [_hipcc(), *FLAGS, *PER_SOURCE_FLAGS.get(src, []), *extra, "-c", path, "-o", obj]
    # an object is reused only if it was built by the same command line from sources not newer than it
    tag = obj + ".cmd"
    stamp = max(os.path.getmtime(path), _newest_header(), os.path.getmtime(__file__))
    if os.path.exists(obj) and os.path.getmtime(obj) >= stamp and os.path.exists(tag) and open(tag).read() == " ".join(cmd):
        return obj, False
    subprocess.check_call(cmd)
    with open(tag, "w") as f:
        f.write(" ".join(cmd))
    return obj, True


def _build_one(lib, objdir, flags, force, verbose):
    os.makedirs(objdir, exist_ok=True)
    if force:
        for f in os.listdir(objdir):
            os.remove(os.path.join(objdir, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda s: _compile(s, list(flags), objdir), SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(lib):
        subprocess.check_call([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs])
        if verbose:
            print(f"[myfa] linked {lib}")
    elif verbose:
        print(f"[myfa] up to date: {lib}")
    return lib


def build_library(force=False, verbose=True, extra_flags=(), tuning=True):
    """Compile every HIP source for gfx950 and link the shared library (and, unless tuning=False,
    its tuning flavour)."""
    lib = _build_one(LIB, OBJ, list(extra_flags), force, verbose)
    if tuning:
        _build_one(LIB_TUNING, OBJ + "_tuning", [*extra_flags, "-DMYFA_TUNING"], force, verbose)
    return lib


def build_experiment(name, flags, verbose=True):
    """A tuning-flavour build with extra -D flags, for A/B runs of compile-time variants in one
    process (tools/ab_attn.py `lib=` specs): my_flash_attn_cuda/exp_<name>.so.  Not shipped."""
    lib = os.path.join(PKG, f"exp_{name}.so")
    return _build_one(lib, OBJ + "_exp_" + name, ["-DMYFA_TUNING", *flags], False, verbose)


if __name__ == "__main__":
    if "--exp" in sys.argv:
        i = sys.argv.index("--exp")
        build_experiment(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build_library(force="--force" in sys.argv)
