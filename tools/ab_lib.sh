#!/bin/bash
# A/B of two builds of libqatvit.so on ONE box (boxes of the pool differ by several per cent): alternates the two libraries under a command.
# usage (inside one gpurun call):  tools/ab_lib.sh <dir with libqatvit_old.so> <rounds> <command...>
#   e.g. tools/ab_lib.sh ab 3 python3 tools/bench_attn.py        - prints the last line of the command's output per arm and round
# The current qat-vit_amd/libqatvit.so is the "new" arm (saved and restored); build the other arm by compiling the old source of the changed
# file(s) into objects of their own and linking them with the remaining build/*.o into <dir>/libqatvit_old.so (the directory is git-ignored).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
DIR=$1; ROUNDS=$2; shift 2
cd "$ROOT"
cp qat-vit_amd/libqatvit.so "$DIR/libqatvit_new.so"
trap 'cp "$DIR/libqatvit_new.so" qat-vit_amd/libqatvit.so' EXIT
for r in $(seq "$ROUNDS"); do
  for v in new old; do
    cp "$DIR/libqatvit_$v.so" qat-vit_amd/libqatvit.so
    echo "$v: $(timeout -k 10 300 "$@" 2>/dev/null | tail -1)"
  done
done
