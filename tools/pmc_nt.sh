#!/bin/bash
# PMC passes over the NT GEMM micro-benchmark (one counter set per run; no tracing domains besides kernel-trace)
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_nt
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  BENCH_N=3 SKIP_TN=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python tools/bench_gemm.py > $OUT/s$i.log 2>&1 || { echo "set $i failed"; tail -5 $OUT/s$i.log; }
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_nt/s*/")):
    for f in glob.glob(d+"*/*counter_collection.csv"):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "gemm_nt" not in k: continue
            acc[(k[:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
