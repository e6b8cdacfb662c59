#!/bin/bash
# One-GPU rehearsal of the data-parallel step under rocprofv3 (kernel + memory-copy trace of rank 0): two ranks of the real engine on device 0,
# gloo as the transport (RCCL refuses two ranks per device; gloo's CUDA all-reduce = D2H copy on its own stream, host reduce, H2D copy).
# Shows WHERE in the backward each gradient bucket's collective starts - the schedule is transport-independent host code (engine.py:
# staged_backward_allreduce).  Output: gpurun_out/dp_trace/ + the overlap report of tools/dp_overlap_report.py.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/dp_trace
rm -rf $OUT && mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 BENCH_ONE_DEVICE=1 BENCH_DIST_BACKEND=gloo
ARGS="--gpus 2 --steps 4 --warmup 2 --batch ${BATCH:-64} --no-cpu-baseline --no-kernel-rates --no-extras"
RANK=1 LOCAL_RANK=1 timeout -k 10 400 python3 $ROOT/bench.py $ARGS > $OUT/rank1.log 2>&1 &
R1=$!
RANK=0 LOCAL_RANK=0 timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o r0 -- python3 $ROOT/bench.py $ARGS > $OUT/rank0.log 2>&1
echo "rank0 rc=$?"
wait $R1
echo "rank1 rc=$?"
grep '"metric"' $OUT/rank0.log | cut -c1-300
find $OUT -name "*.csv" | head
python3 $ROOT/tools/dp_overlap_report.py $OUT | tee $OUT/report.txt
