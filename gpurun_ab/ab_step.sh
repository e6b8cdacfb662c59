set -e
for r in 1 2; do
  for v in new old; do
    cp gpurun_ab/libqatvit_$v.so qat-vit_amd/libqatvit.so
    echo "$v: $(timeout -k 10 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-rates --no-extras 2>/dev/null | python3 -c 'import sys,json; r=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"])')"
  done
done
cp gpurun_ab/libqatvit_new.so qat-vit_amd/libqatvit.so
