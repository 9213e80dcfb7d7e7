#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel of the library, as the compiler reports them
(`hipcc -Rpass-analysis=kernel-resource-usage`). Cross-compiles, no GPU needed.
usage: python3 scripts/resource_table.py [file.hip ...]      (default: every .hip under floxer_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "floxer_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("flx::", "").replace("(anonymous namespace)::", "") for o in out]


def table(path):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark: [^ ]+ +(?:Function )?Name: (\S+)", line) or re.search(r"Name: (\S+) \[-Rpass", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = n
    return rows


def main():
    files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>14s} {'static LDS B':>12s} {'waves/SIMD':>10s}")
    for f in files:
        print(f"-- {os.path.basename(f)}")
        for r in table(f):
            print(f"{r['name'][:58]:58s} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('scratch', 0):14d} {r.get('lds', 0):12d} {r.get('occ', 0):10d}")


if __name__ == "__main__":
    main()
