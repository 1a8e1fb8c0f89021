#!/bin/bash
# round 5, session d: gram-per-sub-batch A/B; block-row solve: two rows per launch (trsm_pair) timing + PMC traffic
TAG=${1:-r05d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batched or fit_batch or split or device_resident or mcmc" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/summary.txt
tail -2 $OUT/pytest.log >> $OUT/summary.txt
V="4,1,-1;4,3,-1;6,3,-1"
for gs in 0 1; do
  BATCH_AB="$V" BATCH_TUNE="potrf_gram_split=$gs" timeout 600 python tools/batched_fit_ab.py 4096 16 27 7 > $OUT/ab_4096_gram$gs.txt 2>&1
  BATCH_AB="$V" BATCH_TUNE="potrf_gram_split=$gs" timeout 600 python tools/batched_fit_ab.py 2048 16 26 9 > $OUT/ab_2048_gram$gs.txt 2>&1
  echo "== potrf_gram_split=$gs" >> $OUT/summary.txt; grep "round 1" $OUT/ab_4096_gram$gs.txt $OUT/ab_2048_gram$gs.txt >> $OUT/summary.txt
done
# block-row solve, one vs two rows per launch: same-session timing (20 steps each, twice) ...
for rep in 1 2; do for pair in 0 1; do
  ROBO_TRSM_PAIR=$pair timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --lean --no-cpu-baseline > $OUT/bench_pair${pair}_$rep.json 2>> $OUT/bench_pair.err
  python -c "
import json; d=json.load(open('$OUT/bench_pair${pair}_$rep.json')); print('trsm_pair=$pair rep $rep: ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['kernel'], 'argmax', d['argmax'])" >> $OUT/summary.txt
done; done
# ... and HBM traffic of both (separate PMC passes, one step each)
for pair in 0 1; do
  for C in FETCH_SIZE WRITE_SIZE; do
    ROBO_TRSM_PAIR=$pair timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_pair$pair/$C -o pmc -- python bench.py --gpus 1 --steps 1 --warmup 0 --lean --no-cpu-baseline > /dev/null 2> $OUT/pmc_pair${pair}_$C.err
    echo "pmc pair=$pair $C rc=$?" >> $OUT/summary.txt
  done
  python tools/rocpd_pmc.py $OUT/pmc_pair$pair > $OUT/pmc_summary_pair$pair.txt 2>&1
  find $OUT/pmc_pair$pair -size +30M -delete
done
python tools/make_traffic_json.py $OUT/pmc_summary_pair0.txt $(cat .git_head 2>/dev/null || echo unknown) trsm_step_gen_kernel headline 4096 16 65536 > $OUT/trsm_traffic_pair0.json 2>> $OUT/summary.txt
python tools/make_traffic_json.py $OUT/pmc_summary_pair1.txt $(cat .git_head 2>/dev/null || echo unknown) trsm_pair_gen_kernel headline 4096 16 65536 > $OUT/trsm_traffic_pair1.json 2>> $OUT/summary.txt
grep -h "bytes_per_launch\|launches" $OUT/trsm_traffic_pair0.json $OUT/trsm_traffic_pair1.json >> $OUT/summary.txt
