#!/bin/bash
# LDS / wait counters of the int8 strip kernel on the micro-benchmark (tools/bench_i8strip.py), one counter set per rocprofv3 pass.
# usage: tools/pmc_i8strip.sh <tag> [variant list, default 0]
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?tag}; VARS=${2:-0}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_strip_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*" | sort -u > "$OUT/available.txt" || true
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  BENCH_ROUNDS=2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/s$i" -- python3 tools/bench_i8strip.py "$VARS" > "$OUT/s$i.log" 2>&1 || { echo "set $i failed"; tail -5 "$OUT/s$i.log"; }
done
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, collections, sys
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_i8_strip" not in k: continue
        tot[(k.split("(")[0].replace("void ", ""), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(tot.items()):
    print(k)
    for c in sorted(v): print(f"    {c:28s} {sum(v[c])/len(v[c]):14.0f}")
PY
