#!/bin/bash
# SQ counters of the three attention kernels at the step's shape (one counter set per run; no tracing domains besides kernel-trace)
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  BENCH_N=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python3 tools/bench_attn.py > $OUT/s$i.log 2>&1 || { echo "set $i failed"; tail -5 $OUT/s$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_attn/s*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_attn" not in k: continue
            acc[k.split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            for c, x in v.items(): tot[k][c] = sum(x) / len(x)
for k, v in tot.items():
    print(k)
    for c in sorted(v): print(f"    {c:34s} {v[c]:16.0f}")
    if "SQ_BUSY_CYCLES" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        # SQ_BUSY_CYCLES counts per SE-quad; derive ratios against wave cycles instead
        pass
    if "SQ_WAVE_CYCLES" in v:
        w = v["SQ_WAVE_CYCLES"]
        print("    -> of wave cycles: waiting (s_waitcnt/barrier) %.0f%%, issue-stalled %.0f%%, issuing %.0f%%; VALU-issuing %.0f%%, LDS-issuing %.0f%%" % (
            100 * v.get("SQ_WAIT_ANY", 0) / w, 100 * v.get("SQ_WAIT_INST_ANY", 0) / w, 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / w,
            100 * v.get("SQ_ACTIVE_INST_VALU", 0) / w, 100 * v.get("SQ_ACTIVE_INST_LDS", 0) / w))
    if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles are summed over all SIMDs (256 CUs x 4)
        print("    -> MFMA pipe busy %.1f%% of SIMD cycles" % (100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
