#!/bin/bash
# SQ / GRBM counters of EVERY kernel of the C2 step (bench.py, batch 256), one counter set per rocprofv3 pass (--kernel-trace + --pmc only: no
# other tracing domain; the program itself follows `--`).  usage: tools/pmc_step_sq.sh <tag> [extra bench.py arguments]
# Writes gpurun_out/pmc_sq_<tag>/summary.txt: per kernel the share of wave cycles spent waiting (s_waitcnt / barrier), issue-stalled and issuing,
# the VALU / LDS issue shares and the MFMA pipe's busy fraction (SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?tag}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_sq_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/s$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-rates --no-extras --no-kernel-legs "$@" > "$OUT/s$i.log" 2>&1 || { echo "set $i failed"; tail -5 "$OUT/s$i.log"; exit 1; }
  echo "pass $i done"
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, collections, sys
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "qv::" not in k and "k_" not in k:
            continue
        tot[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not tot:
    print("no counter rows found"); sys.exit(1)
rows = []
for k, v in tot.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    n = len(v.get("SQ_WAVE_CYCLES", []))
    w = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    rows.append((gui * n, k, n, m, w, gui))
print("# per launch means; wave-cycle shares: wait = s_waitcnt / barrier, stall = issue stall, issue = issuing (VALU / LDS shares of all wave cycles)")
print("# cyc/XCD = GRBM_GUI_ACTIVE / 8; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cyc/XCD x 1024 SIMDs)")
for _, k, n, m, w, gui in sorted(rows, reverse=True):
    cyc = gui / 8.0
    mfma = 100.0 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024) if cyc else 0.0
    print(f"{n:4d} x cyc/XCD {cyc:9.0f}  wait {100*m.get('SQ_WAIT_ANY',0)/w:4.0f}%  stall {100*m.get('SQ_WAIT_INST_ANY',0)/w:4.0f}%  issue {100*m.get('SQ_ACTIVE_INST_ANY',0)/w:4.0f}%"
          f"  VALU {100*m.get('SQ_ACTIVE_INST_VALU',0)/w:4.0f}%  LDS {100*m.get('SQ_ACTIVE_INST_LDS',0)/w:4.0f}%  VMEM {100*m.get('SQ_ACTIVE_INST_VMEM',0)/w:4.0f}%"
          f"  MFMA busy {mfma:5.1f}%  ldsconf {m.get('SQ_LDS_BANK_CONFLICT',0):9.0f}  insts_valu {m.get('SQ_INSTS_VALU',0):11.0f}  insts_lds {m.get('SQ_INSTS_LDS',0):10.0f}  lds_idx {m.get('SQ_LDS_IDX_ACTIVE',0):10.0f}  {k[:110]}")
PY
