#!/bin/bash
# HBM-traffic counters for the bench kernels: separate rocprofv3 passes per counter group
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass), kernel-trace only.
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES"; do
  name=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/$name -o pmc -- python bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  echo "$C rc=$?" >> $OUT/summary.txt
done
python tools/rocpd_pmc.py $OUT > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt >> $OUT/summary.txt
find $OUT -size +30M -delete
