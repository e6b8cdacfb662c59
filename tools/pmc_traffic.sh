#!/bin/bash
# HBM-side traffic of the GEMM kernels from rocprofv3 PMC counters, one counter per pass (MI355X_MICROARCH.md, HBM section:
# FETCH_SIZE / WRITE_SIZE in KiB-units of 1024 B; gfx950: FETCH_SIZE counts half of a wide streaming read -> doubled below).
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_traffic
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  BENCH_N=3 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python tools/bench_gemm.py > $OUT/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $OUT/$c.log; exit 1; }
done
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
order = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
    for f in glob.glob(f"gpurun_out/pmc_traffic/{c}/*/*counter_collection.csv"):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = r["Kernel_Name"]
            if "gemm_" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
# bench_gemm.py launches, per shape, 2 warm-up + BENCH_N timed calls in a fixed order: group consecutive launches of a kernel by 5
out = {}
for k, v in acc.items():
    n = len(v["FETCH_SIZE"])
    shapes = n // 5
    per = []
    for s in range(shapes):
        sl = slice(5 * s, 5 * s + 5)
        m = lambda c: sum(v[c][sl]) / 5
        per.append({"fetch_MB": round(2 * m("FETCH_SIZE") * 1024 / 1e6, 1), "write_MB": round(m("WRITE_SIZE") * 1024 / 1e6, 1),
                    "l2_hit_rate": round(m("TCC_HIT_sum") / max(1.0, m("TCC_HIT_sum") + m("TCC_MISS_sum")), 3)})
    out[k] = per
json.dump(out, open("gpurun_out/pmc_traffic/summary.json", "w"), indent=1)
for k, per in out.items():
    print(k)
    for p in per:
        print("   ", p)
PY
