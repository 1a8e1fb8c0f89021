#!/bin/bash
# round 5, session b: batched-fit schedules after the panel (72 KB) / thin-last-row / interleaved-launch changes
TAG=${1:-r05b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batched or fit_batch or split or edge or golden or device_resident or headline_size or full_size" > $OUT/pytest_batch.log 2>&1; echo "pytest rc=$?" >> $OUT/summary.txt
tail -3 $OUT/pytest_batch.log >> $OUT/summary.txt
V="4,1,-1;6,1,-1;6,2,-1;4,3,-1;5,3,-1;6,3,-1;8,3,-1;6,4,-1"
BATCH_AB="$V" BATCH_TUNE="potrf_thin_last=0" timeout 600 python tools/batched_fit_ab.py 4096 16 27 5 > $OUT/ab_4096_thin0.txt 2>&1
BATCH_AB="$V" timeout 600 python tools/batched_fit_ab.py 4096 16 27 5 > $OUT/ab_4096_thin1.txt 2>&1
BATCH_AB="$V" timeout 300 python tools/batched_fit_ab.py 2048 16 27 7 > $OUT/ab_2048_thin1.txt 2>&1
BATCH_AB="4,1,-1;6,3,-1" timeout 300 python tools/batched_fit_ab.py 1024 8 27 9 > $OUT/ab_1024_thin1.txt 2>&1
FIT_REPS=10 python tools/fit_only.py > $OUT/fit_only.txt 2>&1
cat $OUT/ab_4096_thin0.txt $OUT/ab_4096_thin1.txt $OUT/ab_2048_thin1.txt $OUT/ab_1024_thin1.txt $OUT/fit_only.txt >> $OUT/summary.txt
for v in "6,1,-1"; do
  name=$(echo $v | tr ',' '_')
  BATCH_AB="$v" timeout 300 rocprofv3 --kernel-trace -d $OUT/trace_$name -o t -- python tools/batched_fit_ab.py 4096 16 27 2 > $OUT/trace_$name.log 2>&1
  python tools/batch_trace.py $OUT/trace_$name 60 > $OUT/trace_$name.txt 2>&1
  find $OUT/trace_$name -size +5M -delete
  head -2 $OUT/trace_$name.txt >> $OUT/summary.txt
done
