#!/usr/bin/env python3
"""Static resource table of every gfx950 kernel in poly_amd/csrc (no GPU needed).

    python scripts/kernel_resources.py > profiles/r01_kernel_resources.md

Recompiles each .hip device-side with -Rpass-analysis=kernel-resource-usage (same flags as
poly_amd/build.py) and prints VGPRs / SGPRs / scratch / static LDS / compiler occupancy per
kernel.  Dynamic LDS (most kernels here size their LDS at launch) is not in the static figure;
DESIGN.md states it per kernel.
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from poly_amd import build  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.splitlines()
    short = []
    for d in out:
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*$", "", d)  # drop the parameter list
        d = d.replace("polyhip::", "").replace("(anonymous namespace)::", "")
        short.append(d)
    return short


def main():
    rows = []
    for src in sorted(glob.glob(os.path.join(build.CSRC, "*.hip"))):
        cmd = [build._hipcc()] + build.CXXFLAGS + ["--cuda-device-only", "-c", src, "-o", "/dev/null",
                                                    "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
            if not m:
                continue
            t = m.group(1).strip()
            if t.startswith("Function Name:"):
                cur = {"file": os.path.basename(src), "name": t.split(":", 1)[1].strip()}
                rows.append(cur)
            elif cur is not None and ":" in t:
                k, v = t.rsplit(":", 1)
                cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    print("# Static kernel resources (gfx950, hipcc -O3; scripts/kernel_resources.py)\n")
    print("Occupancy is the compiler's waves/SIMD from registers and static LDS only; kernels that take")
    print("dynamic LDS (profiles, tables, sketch tiles) are further bounded at launch -- see DESIGN.md.\n")
    print("| file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | static LDS B | waves/SIMD |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|")
    for r, n in zip(rows, names):
        print(f"| {r['file']} | `{n}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | "
              f"{r.get('ScratchSize [bytes/lane]')} | {r.get('VGPRs Spill')} | {r.get('LDS Size [bytes/block]')} | "
              f"{r.get('Occupancy [waves/SIMD]')} |")


if __name__ == "__main__":
    main()
