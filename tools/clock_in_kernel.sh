#!/bin/bash
# In-kernel clock of the k-loops of the step's MFMA kernels (s_memtime / s_memrealtime stamps, development builds ab/libqatvit_st<mode>.so:
# gemm.hip compiled with -DQV_NT_EXPERIMENTS=<mode>, linked with the remaining build/*.o).  usage (one gpurun call): tools/clock_in_kernel.sh 18 100 101
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
cp qat-vit_amd/libqatvit.so ab/libqatvit_ship.so
trap 'cp ab/libqatvit_ship.so qat-vit_amd/libqatvit.so' EXIT
for m in "$@"; do
  cp ab/libqatvit_st$m.so qat-vit_amd/libqatvit.so
  echo "== stamps of mode $m"
  STAMP_STEPS=60 timeout -k 10 300 python3 tools/stamp_nt.py 2>/dev/null | grep -v '^{' | grep -E "workgroups|block 0 wave [04]:|k-loop" | cut -c1-250 | head -7
done
