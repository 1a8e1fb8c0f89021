#!/bin/bash
# kernel trace of the fit at several N (leading-dimension / stride experiment)
TAG=${1:-r02x}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for n in "$@"; do
  FIT_N=$n FIT_REPS=6 python tools/fit_only.py > $OUT/fit_$n.txt 2>&1
  FIT_N=$n FIT_REPS=4 rocprofv3 --kernel-trace -d $OUT/prof_$n -o fit -- python tools/fit_only.py > $OUT/prof_$n.log 2>&1
  python tools/fit_trace.py $OUT/prof_$n > $OUT/trace_$n.txt 2>&1
  find $OUT/prof_$n -size +5M -delete
  echo "N=$n"; cat $OUT/fit_$n.txt; tail -2 $OUT/trace_$n.txt
done
