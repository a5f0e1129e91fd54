"""Register budgets of the resident-workgroup attention forward, checked on the gfx950 code hipcc
generates (CPU-only: hipcc cross-compiles).

The kernel's speed rests on two properties of its register allocation that a source edit can lose
without any test of results noticing: no scalar register is spilled in any build (scalar spills are
v_readlane / v_writelane traffic inside the tile loop: the causal builds lost 9 % to it before the
mask was written as VCC compare/select pairs), and the builds the dispatch uses by itself -- head
dim 128, and head dim 64 with one workgroup per CU -- spill no vector register to scratch memory
(at most a handful outside the loops in the causal build that carries both softmax forms).
"""
import importlib.util
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def kernel_resources(asm):
    """{kernel symbol: {field: int}} from the amdhsa.kernels metadata of an assembly listing (one list
    item per kernel, its fields in alphabetical order, so `.name` comes after some of them)."""
    out, cur = {}, None

    def close():
        if cur and "name" in cur:
            out[cur.pop("name")] = cur
    for line in asm.split("\n"):
        item = re.match(r"^  - \.(\w+):\s*(\S*)\s*$", line)          # first field of a kernel's entry
        field = item or re.match(r"^    \.(\w+):\s+(\S+)\s*$", line)
        if item:
            close()
            cur = {}
        if field and cur is not None:
            key, val = field.group(1), field.group(2)
            if key == "name":
                cur["name"] = val
            elif key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                         "group_segment_fixed_size", "private_segment_fixed_size"):
                cur[key] = int(val)
    close()
    return out


def test_metadata_parser():
    asm = ("  - .agpr_count:     0\n    .args:\n      - .address_space:  global\n        .size: 8\n"
           "    .group_segment_fixed_size: 4096\n    .name:           kern_a\n    .sgpr_count:     40\n    .vgpr_spill_count: 3\n"
           "  - .agpr_count:     0\n    .group_segment_fixed_size: 0\n    .name: kern_b\n    .vgpr_count: 128\n")
    res = kernel_resources(asm)
    assert res == {"kern_a": {"group_segment_fixed_size": 4096, "sgpr_count": 40, "vgpr_spill_count": 3},
                   "kern_b": {"group_segment_fixed_size": 0, "vgpr_count": 128}}


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_resident_workgroup_forward_register_budgets():
    haz = _load("check_mfma_hazards", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
    build = _load("myfa_build", os.path.join(ROOT, "flash-attention_amd", "build.py"))
    src = "attn_fwd_stream.hip"
    asm = haz.compile_to_asm(os.path.join(build.CSRC, src), build.PER_SOURCE_FLAGS.get(src, []))
    res = {k: v for k, v in kernel_resources(asm).items() if "attn_fwd_stream_kernel" in k and "vgpr_count" in v}
    # template arguments in the mangled name: <T, D, CAUSAL, SPEC, PRE, WIDE>
    seen = 0
    for name, r in res.items():
        m = re.search(r"attn_fwd_stream_kernelI(\w+?)Li(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)EEE", name)
        assert m, name
        d, causal, spec, pre, wide = (int(m.group(i)) for i in (2, 3, 4, 5, 6))
        seen += 1
        assert r["sgpr_spill_count"] == 0, (name, r)
        if d == 128:
            assert r["vgpr_count"] <= 256 and r["vgpr_spill_count"] <= (4 if causal and pre else 0), (name, r)
            assert r["group_segment_fixed_size"] <= 160 * 1024, (name, r)
        elif wide:
            assert r["vgpr_count"] <= 256 and r["vgpr_spill_count"] == 0, (name, r)
        else:
            assert r["vgpr_count"] <= 128, (name, r)                          # two workgroups per CU
            assert 2 * r["group_segment_fixed_size"] <= 160 * 1024, (name, r)
    assert seen >= 2 * (5 + 10), f"only {seen} builds of the kernel found"
