"""The hand-written VGPR-form MFMAs of the backward kernels (Mma::acc_v / first_v) are invisible
to the compiler's hazard recogniser; tools/check_mfma_hazards.py verifies on the generated gfx950
assembly that nothing touches such a result before the hardware's result latency has passed.
CPU-only: hipcc cross-compiles."""
import importlib.util
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


chk = _load("check_mfma_hazards", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))

PRODUCER = """
_Z6kernelv:
	;;#ASMSTART
	v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]
	;;#ASMEND
"""


def test_checker_flags_an_early_reader():
    seen, bad = chk.check(PRODUCER + "\tds_read_b128 v[20:23], v40\n\tv_mul_f32_e32 v30, v31, v7\n")
    assert seen == 1 and len(bad) == 1 and "v[0:15]" in bad[0]


def test_checker_accepts_chains_and_distance():
    chain = PRODUCER + "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_bf16 v[0:15], v[28:31], v[24:27], v[0:15]\n\t;;#ASMEND\n"
    assert chk.check(chain)[1] == []                                   # accumulating into the same registers
    two = PRODUCER + "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n" * 2 + "\tv_mul_f32_e32 v30, v31, v7\n"
    assert chk.check(two)[1] == []                                     # two MFMAs issued in between
    slots = PRODUCER + "\tv_add_f32_e32 v40, v41, v42\n" * 16 + "\tv_mul_f32_e32 v30, v31, v7\n"
    assert chk.check(slots)[1] == []                                   # sixteen issue slots in between
    nops = PRODUCER + "\ts_nop 15\n\tv_mul_f32_e32 v30, v31, v7\n"
    assert chk.check(nops)[1] == []
    one = PRODUCER + "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n\tv_exp_f32_e32 v3, v50\n"
    assert len(chk.check(one)[1]) == 1                                 # a write counts like a read


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_backward_kernels_keep_the_result_latency():
    sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
    build = _load("myfa_build", os.path.join(ROOT, "flash-attention_amd", "build.py"))
    src = "attn_bwd_mfma.hip"
    asm = chk.compile_to_asm(os.path.join(build.CSRC, src), build.PER_SOURCE_FLAGS.get(src, []))
    seen, bad = chk.check(asm)
    assert seen >= 8 * 32, "the kernels' hand-written MFMAs were not found in the assembly"
    assert bad == [], "\n".join(bad[:10])


def test_trans_use_checker():
    asm = "_Z1kv:\n\tv_exp_f32_e32 v5, v1\n\tv_max3_f32 v9, v5, v6, v7\n"
    assert chk.check_trans_use(asm) == (1, chk.check_trans_use(asm)[1]) and len(chk.check_trans_use(asm)[1]) == 1
    assert chk.check_trans_use("_Z1kv:\n\tv_exp_f32_e32 v5, v1\n\ts_nop 0\n\tv_max3_f32 v9, v5, v6, v7\n")[1] == []
    assert chk.check_trans_use("_Z1kv:\n\tv_exp_f32_e32 v5, v1\n\tv_add_f32_e32 v9, v6, v7\n\tv_mul_f32_e32 v2, v5, v5\n")[1] == []
    assert chk.check_trans_use("_Z1kv:\n\tv_exp_f32_e32 v5, v1\n\tv_exp_f32_e32 v5, v2\n")[1] == []      # overwritten, not read
    assert len(chk.check_trans_use("_Z1kv:\n\tv_rcp_f32_e32 v5, v1\n\tv_fma_f32 v0, v[4:7], v2, v3\n")[1]) == 1


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
@pytest.mark.parametrize("src,min_mfmas", [("attn_fwd_stream.hip", 0), ("attn_fwd_mfma16.hip", 0), ("attn_fwd_mfma256.hip", 8 * 48),
                                           ("attn_bwd_mfma.hip", 8 * 32), ("attn_bwd_wide.hip", 8 * 48)])
def test_attention_kernels_keep_both_rules(src, min_mfmas):
    """Every attention source with inline asm beside transcendental instructions or with hand-written
    MFMAs: no early reader of an MFMA result, and no instruction reading a transcendental result one
    issue slot behind it (an asm max on fresh exp2 results did exactly that in one build of the
    resident-workgroup forward: its go-back test read stale registers, float16 weights overflowed)."""
    build = _load("myfa_build", os.path.join(ROOT, "flash-attention_amd", "build.py"))
    asm = chk.compile_to_asm(os.path.join(build.CSRC, src), build.PER_SOURCE_FLAGS.get(src, []))
    seen, bad = chk.check(asm)
    assert seen >= min_mfmas and bad == [], "\n".join(bad[:10])
    tseen, tbad = chk.check_trans_use(asm)
    assert tseen > 0 and tbad == [], "\n".join(tbad[:10])
