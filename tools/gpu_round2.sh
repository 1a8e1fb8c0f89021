#!/bin/bash
# One GPU-box session of round 2: the bench lines of every BASELINE configuration, the RCCL path on one rank,
# a kernel trace of the headline command, PMC passes (one counter group per pass, kernel-trace only).
TAG=${1:-r02}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt
timeout 600 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.txt
: > $OUT/configs.jsonl
for c in c2 c3 c4 c5; do
  timeout 600 python bench.py --gpus 1 --config $c --no-cpu-baseline >> $OUT/configs.jsonl 2>> $OUT/configs.err; echo "$c rc=$?" >> $OUT/summary.txt
done
ROBO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --no-cpu-baseline > $OUT/forcedist.json 2> $OUT/forcedist.err; echo "forcedist rc=$?" >> $OUT/summary.txt
ROBO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --config c3 > $OUT/forcedist_c3.json 2>> $OUT/forcedist.err; echo "forcedist c3 rc=$?" >> $OUT/summary.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --gpus 1 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" >> $OUT/summary.txt
python tools/rocpd_stats.py $OUT/prof/bench_results.db > $OUT/bench_kernel_stats.csv 2>> $OUT/prof.err
find $OUT/prof -size +20M -delete
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc/$name -o pmc -- python bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $C rc=$?" >> $OUT/summary.txt
done
python tools/rocpd_pmc.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
find $OUT/pmc -size +30M -delete
cat $OUT/summary.txt; head -c 600 $OUT/bench.json; echo; cut -c1-400 $OUT/configs.jsonl; head -14 $OUT/bench_kernel_stats.csv; grep -i "trsm_step_gen\|potrf_step\|gram_kernel" $OUT/pmc_summary.txt | head -20
