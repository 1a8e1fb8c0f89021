#!/bin/bash
# round 5, session a: batched-fit schedules (A/B + kernel traces)
TAG=${1:-r05a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batched or fit_batch or split" > $OUT/pytest_batch.log 2>&1; echo "pytest rc=$?" >> $OUT/summary.txt
tail -3 $OUT/pytest_batch.log >> $OUT/summary.txt
timeout 600 python tools/batched_fit_ab.py 4096 16 27 5 > $OUT/batched_ab_4096.txt 2>&1; echo "ab4096 rc=$?" >> $OUT/summary.txt
timeout 300 python tools/batched_fit_ab.py 2048 16 27 7 > $OUT/batched_ab_2048.txt 2>&1; echo "ab2048 rc=$?" >> $OUT/summary.txt
cat $OUT/batched_ab_4096.txt $OUT/batched_ab_2048.txt >> $OUT/summary.txt
for v in "6,1,-1" "6,2,-1" "6,3,-1"; do
  name=$(echo $v | tr ',' '_')
  BATCH_AB="$v" timeout 300 rocprofv3 --kernel-trace -d $OUT/trace_$name -o t -- python tools/batched_fit_ab.py 4096 16 27 2 > $OUT/trace_$name.log 2>&1
  python tools/batch_trace.py $OUT/trace_$name 120 > $OUT/trace_$name.txt 2>&1
  find $OUT/trace_$name -size +5M -delete
  head -2 $OUT/trace_$name.txt >> $OUT/summary.txt
done
