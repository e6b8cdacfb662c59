#!/usr/bin/env python3
"""Register / spill / scratch report per kernel of one HIP source file (hipcc -Rpass-analysis=kernel-resource-usage, gfx950).
usage: tools/kernel_regs.py <file.hip under qat-vit_amd/csrc> [substring of the demangled kernel name]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src, "-o", "/tmp/kernel_regs.o"] + sys.argv[3:] + [
                    "-Rpass-analysis=kernel-resource-usage"], cwd=os.path.join(ROOT, "qat-vit_amd", "csrc"), capture_output=True, text=True)
cur, rows = None, {}
for ln in r.stderr.splitlines():
    m = re.search(r"remark:\s*Function Name: (\S+)", ln)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
if r.returncode != 0:
    print(r.stderr[-3000:])
    sys.exit(1)
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    g = v.get
    print(f"VGPR {g('VGPRs', 0):4d} AGPR {g('AGPRs', 0):4d} spill {g('VGPRs Spill', 0):4d} scratch {g('ScratchSize', 0):5d} SGPR {g('TotalSGPRs', 0):4d} occ {g('Occupancy', 0)}  {name[:130]}")
