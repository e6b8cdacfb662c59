#!/bin/bash
# A/B of environment knobs on the default bench workload, one process per arm, same box: tools/ab_env.sh "K1=V1 K2=V2" "K3=V3" ...
for arm in "" "$@"; do
  r=$(env $arm python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-rates --no-extras 2>/dev/null | grep '"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k[:34]: (v['avg_us_per_launch'], v['ms_per_step']) for k, v in d['mfma_gemms'].items()})")
  echo "[${arm:-default}] ms/step img/s nt-split-us others: $r"
done
