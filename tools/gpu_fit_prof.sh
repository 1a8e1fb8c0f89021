#!/bin/bash
TAG=${1:-r02x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python tools/fit_only.py > $OUT/fit_only.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof -o fit -- python tools/fit_only.py > $OUT/prof.log 2>&1
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/fit_kernel_stats.csv; done
find $OUT/prof -size +5M -delete
cat $OUT/fit_only.txt; head -12 $OUT/fit_kernel_stats.csv
