"""The split softmax kernels hand partial (max, sum) pairs from one workgroup to another inside a
launch.  That is correct only if the generated code stores them with device scope, waits for the
stores in front of the barrier that precedes the arrival counter's increment, and reads them back
with device-scope loads -- properties of the gfx950 assembly, which tools/check_split_handoff.py
checks.  CPU-only: hipcc cross-compiles."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


chk = _load("check_split_handoff", os.path.join(ROOT, "tools", "check_split_handoff.py"))
haz = _load("check_mfma_hazards", os.path.join(ROOT, "tools", "check_mfma_hazards.py"))

NAME = "_ZN4myfa12_GLOBAL__N_119softmax_cols_streamIfLi16ELi4ELi256ELb0ELi1EEEvPKT_PS2_lllNS_8PrologueElilPjPf"
GOOD = [
    "global_load_dwordx4 v[0:3], v[4:5], off nt",
    "global_store_dword v[2:3], v70, off sc1",
    "global_store_dword v[2:3], v4, off offset:256 sc1",
    "s_waitcnt vmcnt(0)",
    "s_barrier",
    "global_atomic_add v1, v1, v2, s[6:7] sc0",
    "s_waitcnt vmcnt(0)",
    "s_barrier",
    "buffer_load_dword v5, v4, s[4:7], 0 offen sc1",
    "global_store_dword v[2:3], v4, off",
]


def _asm(lines, name=NAME):
    return name + ":\n" + "".join("\t" + s + "\n" for s in lines) + "\ts_endpgm\n"


def test_checker_accepts_the_expected_shape():
    assert chk.check(_asm(GOOD)) == (1, [])


@pytest.mark.parametrize("drop,what", [
    ("global_store_dword v[2:3], v70, off sc1|global_store_dword v[2:3], v70, off", "store without sc1"),
    ("s_waitcnt vmcnt(0)|s_nop 0", "no s_waitcnt vmcnt(0)"),
    ("buffer_load_dword v5, v4, s[4:7], 0 offen sc1|buffer_load_dword v5, v4, s[4:7], 0 offen", "merge load without sc1"),
])
def test_checker_flags_each_broken_rule(drop, what):
    old, new = drop.split("|")
    lines = list(GOOD)
    lines[lines.index(old)] = new      # (the first occurrence: the wait in front of the barrier)
    seen, bad = chk.check(_asm(lines))
    assert seen == 1 and any(what in b for b in bad), bad


def test_checker_ignores_the_other_phases():
    assert chk.check(_asm(GOOD, NAME.replace("ELi1EEEv", "ELi2EEEv"))) == (0, [])


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
@pytest.mark.parametrize("src,at_least", [("softmax_rows.hip", 4), ("softmax_strided.hip", 6)])
def test_split_kernels_keep_the_handoff_rules(src, at_least):
    asm = haz.compile_to_asm(os.path.join(ROOT, "flash-attention_amd", "csrc", src), [])
    seen, bad = chk.check(asm)
    assert seen >= at_least, f"only {seen} phase-1 kernels found in {src}"
    assert bad == [], "\n".join(bad[:10])
