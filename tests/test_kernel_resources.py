"""Register budgets of the attention kernels whose speed depends on them, checked on the gfx950 code
hipcc generates (CPU-only: hipcc cross-compiles).

The resident-workgroup forward (head dims up to 128):

The kernel's speed rests on two properties of its register allocation that a source edit can lose
without any test of results noticing: no scalar register is spilled in any build (scalar spills are
v_readlane / v_writelane traffic inside the tile loop: the causal builds lost 9 % to it before the
mask was written as VCC compare/select pairs), and the builds the dispatch uses by itself -- head
dim 128, and head dim 64 with one workgroup per CU -- spill no vector register to scratch memory
(at most a handful outside the loops in the causal build that carries both softmax forms).

The one-wave-per-SIMD kernels for head dims above 128 (forward and backward): a spilled vector
register inside their tile loop costs far more than its two instructions, because scratch traffic
shares the counter (vmcnt) on which the LDS-DMA fetches of the coming tiles are waited for -- every
reload then waits for the fetches issued a moment ago.  The 256-column forward uses all 512
registers of a lane; the few values it spills live outside the loop (checked on the code: no
scratch instruction between the loop's barriers).
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
    # template arguments in the mangled name: <T, D, CAUSAL, SPEC, PRE, WIDE, LM>
    seen = 0
    for name, r in res.items():
        m = re.search(r"attn_fwd_stream_kernelI(\w+?)Li(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)EEE", name)
        assert m, name
        d, causal, spec, pre, wide, lm = (int(m.group(i)) for i in (2, 3, 4, 5, 6, 7))
        seen += 1
        assert r["sgpr_spill_count"] == 0, (name, r)
        if d == 128:
            assert r["vgpr_count"] <= 256 and r["vgpr_spill_count"] <= (4 if (causal and pre) or lm else 0), (name, r)
            assert r["group_segment_fixed_size"] <= 160 * 1024, (name, r)
        elif wide:
            assert r["vgpr_count"] <= 256 and r["vgpr_spill_count"] == 0, (name, r)
        else:
            assert r["vgpr_count"] <= 128, (name, r)                          # two workgroups per CU
            assert 2 * r["group_segment_fixed_size"] <= 160 * 1024, (name, r)
    assert seen >= 2 * (5 + 10), f"only {seen} builds of the kernel found"


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_wide_head_kernels_register_budgets():
    haz = _load("check_mfma_hazards", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
    build = _load("myfa_build", os.path.join(ROOT, "flash-attention_amd", "build.py"))
    seen = 0
    for src, kern in (("attn_fwd_mfma256.hip", "attn_fwd_mfma_wide_kernel"), ("attn_bwd_wide.hip", "attn_bwd_d")):
        asm = haz.compile_to_asm(os.path.join(build.CSRC, src), build.PER_SOURCE_FLAGS.get(src, []))
        res = {k: v for k, v in kernel_resources(asm).items() if kern in k and "vgpr_count" in v}
        for name, r in res.items():
            seen += 1
            assert r["sgpr_spill_count"] == 0, (name, r)
            assert r["vgpr_count"] <= 512 and r["group_segment_fixed_size"] <= 160 * 1024, (name, r)
            wide256_fwd = "attn_fwd_mfma_wide_kernel" in name and "Li256E" in name
            assert r["vgpr_spill_count"] <= (4 if wide256_fwd else 0), (name, r)
        if src == "attn_fwd_mfma256.hip":
            # the tile loop of every build is free of scratch traffic: between the first bare s_barrier
            # inside a loop and the loop's back edge there is no scratch instruction
            for body in re.split(r"\n\s+s_endpgm", asm):
                m = re.search(r"^(_ZN4myfa\S*attn_fwd_mfma_wide_kernel\S*):", body, re.M)
                if not m:
                    continue
                lines = body[m.end():].split("\n")
                labels = {ln.split(":")[0]: i for i, ln in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", ln)}
                loops = []
                for i, ln in enumerate(lines):
                    b = re.match(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
                    if b and labels.get(b.group(1), i + 1) < i:                  # a backward branch
                        loops.append((labels[b.group(1)], i))
                mfma_loops = [(a, z) for a, z in loops if sum("v_mfma" in ln for ln in lines[a:z]) >= 16]
                assert mfma_loops, m.group(1)
                for a, z in mfma_loops:
                    assert not any("scratch_" in ln for ln in lines[a:z]), (m.group(1), a, z)
    assert seen >= 8 + 24, f"only {seen} builds found"


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_float32_backward_register_budgets():
    """The float32 backward keeps its owned fragments and its accumulators in registers for the whole
    walk over the other side: no build up to 256 columns may spill, the builds up to head dim 128 must
    leave room for two workgroups per CU (256 registers), and the double-buffered blocks of the widest
    build must fit the CU's LDS."""
    haz = _load("check_mfma_hazards", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))
    build = _load("myfa_build", os.path.join(ROOT, "flash-attention_amd", "build.py"))
    src = "attn_bwd_f32.hip"
    asm = haz.compile_to_asm(os.path.join(build.CSRC, src), build.PER_SOURCE_FLAGS.get(src, []))
    res = {k: v for k, v in kernel_resources(asm).items() if "attn_bwd_f32_kernel" in k and "vgpr_count" in v}
    for name, r in res.items():
        m = re.search(r"attn_bwd_f32_kernelILi(\d+)E", name)
        assert m, name
        dpad = int(m.group(1))
        # (the 512-column build, for head dims 257..512, holds 256 registers of owned fragments and
        # spills some tens of values: it exists for completeness of the range, not for speed)
        assert r["sgpr_spill_count"] == 0 and r["vgpr_spill_count"] <= (0 if dpad <= 256 else 96), (name, r)
        assert r["vgpr_count"] <= (256 if dpad <= 128 else 512), (name, r)
        assert 4 * (4 * 16 * (dpad + 4) + 64) <= 160 * 1024
    assert len(res) == 5 * 8, f"{len(res)} builds of the kernel found"
