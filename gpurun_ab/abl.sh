set -e
for r in 1 2; do
  for v in new abl528 abl1040 abl2064; do
    cp gpurun_ab/libqatvit_$v.so qat-vit_amd/libqatvit.so
    echo "$v: $(timeout -k 10 100 python3 tools/bench_attn.py 2>/dev/null | tail -1)"
  done
done
cp gpurun_ab/libqatvit_new.so qat-vit_amd/libqatvit.so
