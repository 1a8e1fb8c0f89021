#!/bin/bash
# round 5, session e: where a batched fit at N = 2048 (config 3's model size) spends its time
TAG=${1:-r05e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for v in "4,1,-1" "4,3,-1"; do
  name=$(echo $v | tr ',' '_')
  BATCH_AB="$v" timeout 300 rocprofv3 --kernel-trace -d $OUT/trace_$name -o t -- python tools/batched_fit_ab.py 2048 16 26 2 > $OUT/trace_$name.log 2>&1
  python tools/batch_trace.py $OUT/trace_$name 400 > $OUT/trace_$name.txt 2>&1
  find $OUT/trace_$name -size +5M -delete
  head -2 $OUT/trace_$name.txt >> $OUT/summary.txt
done
BATCH_AB="4,1,-1;4,2,-1;4,3,-1;3,3,-1;2,3,-1" timeout 300 python tools/batched_fit_ab.py 2048 16 26 9 > $OUT/ab_2048.txt 2>&1
BATCH_AB="4,1,-1;4,3,-1;3,3,-1" timeout 300 python tools/batched_fit_ab.py 1536 16 26 9 > $OUT/ab_1536.txt 2>&1
BATCH_AB="4,1,-1;4,3,-1;3,3,-1" timeout 300 python tools/batched_fit_ab.py 3072 16 26 7 > $OUT/ab_3072.txt 2>&1
grep "round 1" $OUT/ab_2048.txt $OUT/ab_1536.txt $OUT/ab_3072.txt >> $OUT/summary.txt
