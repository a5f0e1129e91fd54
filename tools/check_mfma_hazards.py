#!/usr/bin/env python3
"""Static check of the hand-written (inline asm) MFMAs in a HIP source.

`Mma::acc_v / first_v` (csrc/attn_mfma_common.h) write their accumulator to VECTOR registers
without the compiler knowing that the instruction is an MFMA, so the compiler inserts none of
the wait states the hardware asks for between an MFMA and a vector instruction that touches its
result (11 issue slots for the 8-pass 32x32x16 MFMA).  The kernels keep the distance by
construction: the planned position of every consumer is at least two MFMAs behind the producer.
The compiler may still move a vector instruction up within that window (nothing orders a pure
instruction against an MFMA it does not depend on), so this script compiles the source to gfx950
assembly and verifies the outcome on every kernel:

    for every inline-asm MFMA with result v[a:b], no instruction other than an MFMA that
    accumulates into the very same v[a:b] may name a register of v[a:b] until two more MFMAs
    (each holds the issue port of the matrix core for its 8 passes) or 16 other instructions
    (one issue slot each at least; the rule asks for 11) have been issued.

    python tools/check_mfma_hazards.py flash-attention_amd/csrc/attn_bwd_mfma.hip [extra hipcc flags]

The same run checks a second rule the compiler cannot keep for inline asm (`check_trans_use`): no
instruction reads the result of a transcendental instruction issued immediately in front of it.

Exit status 0 and a one-line summary if every kernel passes; 1 and the offending lines otherwise.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIN_MFMAS_BETWEEN = 2
MIN_SLOTS_BETWEEN = 16
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_to_asm(src, extra):
    """gfx950 assembly of `src` (device side).  Kept in a per-user cache under the temp directory,
    keyed by the contents of the source, of every header beside it and the flags: several checks
    read the same listing and a compile takes up to a minute."""
    import glob
    import hashlib
    h = hashlib.sha256(" ".join(extra).encode())
    csrc = os.path.dirname(os.path.abspath(src))
    for f in [src] + sorted(glob.glob(os.path.join(csrc, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    cache_dir = os.path.join(tempfile.gettempdir(), f"myfa_asm_cache_{os.getuid()}")
    cached = os.path.join(cache_dir, os.path.basename(src) + "." + h.hexdigest()[:24] + ".s")
    if os.path.exists(cached):
        with open(cached) as f:
            return f.read()
    text = _compile_to_asm(src, extra)
    try:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = cached + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, cached)
    except OSError:
        pass
    return text


def _compile_to_asm(src, extra):
    hipcc = "/opt/rocm/bin/hipcc"
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "flash-attention_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src, *extra]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    with open(out) as f:
        text = f.read()
    os.unlink(out)
    return text


def check(asm):
    """Returns (number of inline-asm MFMAs seen, list of violations)."""
    seen, bad = 0, []
    kernel = None
    in_asm = False
    open_results = []       # [first, last, mfmas issued since, line text, issue slots since]
    for n, line in enumerate(asm.split("\n"), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, open_results = m.group(1), []
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        code = s.split(";")[0]
        is_mfma = code.startswith("v_mfma")
        ops = code.split(None, 1)[1] if " " in code else ""
        dst = vregs(ops.split(",")[0]) if is_mfma else set()
        touched = vregs(ops)
        for r in open_results:
            lo, hi = r[0], r[1]
            if not any(lo <= v <= hi for v in touched):
                continue
            same_chain = is_mfma and dst == set(range(lo, hi + 1))
            if not same_chain:
                bad.append(f"{kernel}: line {n}: `{code}` touches v[{lo}:{hi}] {r[2]} MFMA(s) / {r[4]} slots after `{r[3]}`")
        nop = re.match(r"s_nop\s+(\d+)", code)
        for r in open_results:
            r[4] += int(nop.group(1)) + 1 if nop else 1
            if is_mfma:
                r[2] += 1
        open_results = [r for r in open_results if r[2] < MIN_MFMAS_BETWEEN and r[4] < MIN_SLOTS_BETWEEN
                        and not (is_mfma and dst and dst == set(range(r[0], r[1] + 1)))]
        if is_mfma and in_asm and dst:
            seen += 1
            open_results.append([min(dst), max(dst), 0, code, 0])
    return seen, bad


TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def check_trans_use(asm):
    """A second rule of the same kind.  The result of a transcendental instruction (v_exp_f32 and its
    kin) arrives one issue slot late: the compiler puts a wait state (or another instruction) between
    such an instruction and a vector instruction that reads its result -- but only for instructions
    it knows; the text of an inline asm statement gets none.  Returns (transcendental instructions
    seen, violations): an instruction that names the result register of the instruction issued
    immediately in front of it, when that one is transcendental."""
    seen, bad = 0, []
    kernel, prev = None, None       # prev = (result registers, text) of the last issued instruction if transcendental
    for n, line in enumerate(asm.split("\n"), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, prev = m.group(1), None
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        code = s.split(";")[0]
        ops = code.split(None, 1)[1] if " " in code else ""
        if prev is not None and code.startswith("v_"):
            srcs = vregs(",".join(ops.split(",")[1:]))
            if srcs & prev[0]:
                bad.append(f"{kernel}: line {n}: `{code.strip()}` reads the result of `{prev[1].strip()}` issued right in front of it")
        prev = None
        if code.startswith(TRANS):
            seen += 1
            prev = (vregs(ops.split(",")[0]), code)
    return seen, bad


def main():
    src = sys.argv[1]
    asm = compile_to_asm(src, sys.argv[2:])
    seen, bad = check(asm)
    tseen, tbad = check_trans_use(asm)
    for b in bad + tbad:
        print(b)
    print(f"{os.path.basename(src)}: {seen} hand-written MFMAs, {len(bad)} result-latency violations; "
          f"{tseen} transcendental instructions, {len(tbad)} read one slot behind")
    return 1 if bad or tbad or not seen else 0


if __name__ == "__main__":
    sys.exit(main())
