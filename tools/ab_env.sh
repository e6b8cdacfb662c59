#!/bin/bash
# A/B of environment knobs on ONE box: alternates `env A` and `env B` under the quick bench (C2, no extras), prints ms/step per arm and round.
# usage (inside one gpurun call): tools/ab_env.sh <rounds> "<VAR=.. VAR=..>" "<VAR=.. VAR=..>" ["<third arm>" ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=$1; shift
cd "$ROOT"
for r in $(seq "$ROUNDS"); do
  for arm in "$@"; do
    ms=$(env $arm timeout -k 10 300 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-rates --no-extras 2>/dev/null | python3 -c 'import sys,json; print(json.loads([l for l in sys.stdin if l.startswith("{")][-1])["ms_per_step"])')
    echo "[$arm] $ms"
  done
done
