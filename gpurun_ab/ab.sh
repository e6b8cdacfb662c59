set -e
cp qat-vit_amd/libqatvit.so gpurun_ab/libqatvit_new.so
for r in 1 2 3; do
  for v in new old; do
    cp gpurun_ab/libqatvit_$v.so qat-vit_amd/libqatvit.so
    echo "$v: $(timeout -k 10 100 python3 tools/bench_attn.py 2>/dev/null | tail -1)"
  done
done
cp gpurun_ab/libqatvit_new.so qat-vit_amd/libqatvit.so
