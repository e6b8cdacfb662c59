#!/bin/bash
# rocprofv3 kernel stats of the default bench workload (C2): tools/prof_step.sh <tag> [extra bench args]
# writes gpurun_out/prof_<tag>/ and prints per-kernel time per STEP (steps in the trace = launches of k_kd_ce: warm-up + timed + per-class legs)
set -eu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?tag}; shift
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c2 -- python3 $ROOT/bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-kernel-rates --no-extras "$@" > $OUT/bench.log 2>&1
echo "rc=$?"
grep '"metric"' $OUT/bench.log | cut -c1-200
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = float(sum(int(r["Calls"]) for r in rows if "k_kd_ce" in r["Name"]) or 29)
print(f"steps in the trace: {steps:.0f}")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time per step: {tot/steps/1e6:.3f} ms   ({f})")
for r in rows[:32]:
    print(f'{float(r["TotalDurationNs"])/steps/1e3:9.1f} us/step {int(r["Calls"])/steps:6.1f} calls/step {float(r["AverageNs"])/1e3:8.1f} us avg  {r["Name"][:110]}')
PY
