#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR/AGPR/SGPR/scratch/LDS/occupancy) for a HIP source.

    python tools/kernel_usage.py flash-attention_amd/csrc/attn_fwd_mfma.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "flash-attention_amd", "csrc"),
           "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:"):
            cur = {"name": txt.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in txt:
            k, v = txt.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"myfa::\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name)
        print(f"{name:58s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>4} sgpr={r.get('TotalSGPRs','?'):>4} "
              f"scratch={r.get('ScratchSize [bytes/lane]','?'):>5} lds={r.get('LDS Size [bytes/block]','?'):>6} "
              f"occ={r.get('Occupancy [waves/SIMD]','?')}")


if __name__ == "__main__":
    main()
