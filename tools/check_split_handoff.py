#!/usr/bin/env python3
"""Static check of the cross-workgroup hand-off in the split softmax kernels, on the gfx950 assembly.

The first launch of the split row / column softmax (`softmax_rows_stream<..., 1>`,
`softmax_cols_stream<..., 1>`) leaves one (max, sum) pair per workgroup in a workspace; the
workgroup that counts itself in last merges the pairs of its row or column block.  The pairs cross
XCDs inside one launch, so the hand-off holds only if, in the code hipcc generates,

  1. every store in front of the counter increment carries the device-scope bit (`sc1`) -- the
     kernel stores nothing else in that launch;
  2. the storing waves wait for those stores (`s_waitcnt vmcnt(0)`) in front of the workgroup
     barrier (`s_barrier`) that precedes the increment (`global_atomic_add`): the barrier alone
     waits for no memory operation;
  3. the merging workgroup reads the pairs with loads that carry `sc1` (they bypass the CU's
     vector cache, which no other CU's store refreshes).

    python tools/check_split_handoff.py            # compiles both sources and checks them
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = re.compile(r"^(_ZN4myfa\S*softmax_(?:rows|cols)_stream\S*):")
# mangled template list ends with the PHASE argument: ...ELi1EEEv...
PHASE1 = re.compile(r"ELi1EEEv")


def kernels(asm):
    """Yields (name, [instruction lines]) for every split-softmax stream kernel in the assembly."""
    name, body = None, []
    for line in asm.split("\n"):
        m = KERNEL.match(line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        s = line.strip()
        if s.startswith("s_endpgm"):
            yield name, body
            name = None
            continue
        if s and not s.startswith((";", ".")) and not re.match(r"^\.?\w+:", s):
            body.append(s)


def check_kernel(body):
    """Violations (strings) of the three rules in one phase-1 kernel's instruction list."""
    bad = []
    atomics = [i for i, s in enumerate(body) if s.startswith(("global_atomic_add", "buffer_atomic_add", "flat_atomic_add"))]
    if len(atomics) != 1:
        return [f"expected exactly one counter increment, found {len(atomics)}"]
    at = atomics[0]
    stores = [i for i, s in enumerate(body[:at]) if re.match(r"^(global|buffer|flat)_store", s)]
    if not stores:
        bad.append("no store of partial results in front of the counter increment")
    for i in stores:
        if not re.search(r"\bsc1\b", body[i]):
            bad.append(f"store without sc1 in front of the increment: {body[i]}")
    barriers = [i for i in range(at) if body[i].startswith("s_barrier")]
    if not barriers or (stores and barriers[-1] < stores[-1]):
        bad.append("no s_barrier between the partial stores and the counter increment")
    elif stores:
        between = body[stores[-1] + 1:barriers[-1]]
        if not any(re.match(r"^s_waitcnt\b.*vmcnt\(0\)", s) for s in between):
            bad.append("no s_waitcnt vmcnt(0) between the last partial store and the barrier in front of the increment")
    loads = [s for s in body[at + 1:] if re.match(r"^(global|buffer|flat)_load", s)]
    if not loads:
        bad.append("no merge loads behind the counter increment")
    for s in loads:
        if not re.search(r"\bsc1\b", s):
            bad.append(f"merge load without sc1: {s}")
    return bad


def check(asm):
    """Returns (number of phase-1 kernels seen, list of violations)."""
    seen, bad = 0, []
    for name, body in kernels(asm):
        if not PHASE1.search(name):
            continue
        seen += 1
        bad += [f"{name}: {b}" for b in check_kernel(body)]
    return seen, bad


def main():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from check_mfma_hazards import compile_to_asm
    total = 0
    for src in ("softmax_rows.hip", "softmax_strided.hip"):
        seen, bad = check(compile_to_asm(os.path.join(ROOT, "flash-attention_amd", "csrc", src), []))
        print(f"{src}: {seen} phase-1 kernels, {len(bad)} violations")
        for b in bad[:20]:
            print("  " + b)
        total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
