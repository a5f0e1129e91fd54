#!/usr/bin/env python3
"""Installs the MI355X drop-in for the reference's `my_flash_attn_cuda` extension.

Takes the place of the reference's setup.py (CUDAExtension "my_flash_attn_cuda", nvcc,
sm_75; setup.py:107-143 there).  Nothing here goes through torch.utils.cpp_extension:
the kernels are compiled by hipcc for gfx950 into one C-ABI shared library
(flash-attention_amd/build.py) that ships inside the pure-Python package of the same
import name.  No git submodules are fetched and no compiler is forced.

    python setup.py build_ext --inplace     # just build libmyfa_gfx950.so in-tree
    pip install -e .                        # development install
"""
import importlib.util
import os

from setuptools import Command, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.join(HERE, "flash-attention_amd")


def _build_library():
    spec = importlib.util.spec_from_file_location("_myfa_build", os.path.join(PKG_ROOT, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_library()


class BuildHip(Command):
    description = "compile the HIP kernels for gfx950 into libmyfa_gfx950.so"
    user_options = [("inplace", "i", "accepted for compatibility; the library is always built in-tree")]

    def initialize_options(self):
        self.inplace = 1

    def finalize_options(self):
        pass

    def run(self):
        _build_library()


class BuildPyWithHip(build_py):
    def run(self):
        _build_library()
        super().run()


setup(
    name="my_flash_attn",
    version="0.1.0",
    description="Row-softmax and flash-attention forward for AMD Instinct MI355X (gfx950), "
                "drop-in for my_flash_attn_cuda",
    package_dir={"": "flash-attention_amd"},
    packages=["my_flash_attn_cuda"],
    package_data={"my_flash_attn_cuda": ["libmyfa_gfx950.so"]},
    python_requires=">=3.9",
    cmdclass={"build_ext": BuildHip, "build_py": BuildPyWithHip},
)
