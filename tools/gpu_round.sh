#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace.  Everything worth
# keeping goes to gpurun_out/ (merged back by gpurun).   usage: tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
echo "== device" | tee $OUT/summary.txt
rocm-smi --showproductname 2>/dev/null | head -8 >> $OUT/summary.txt
echo "== smoke" | tee -a $OUT/summary.txt
timeout 300 python __graft_entry__.py smoke >> $OUT/summary.txt 2>&1
echo "smoke rc=$?" | tee -a $OUT/summary.txt
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -rA -s --durations=20 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.json >> $OUT/summary.txt
tail -5 $OUT/bench.err >> $OUT/summary.txt
echo "== rocprofv3 kernel trace" | tee -a $OUT/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*stats*" | head >> $OUT/summary.txt
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -30 $f >> $OUT/summary.txt; done
# the raw kernel trace can be large; keep the stats, drop traces above 20 MB
find $OUT/prof -size +20M -delete
echo "== done" | tee -a $OUT/summary.txt
