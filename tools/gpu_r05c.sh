#!/bin/bash
# round 5, session c: full GPU suite + smoke + default bench + inproc forms on one GPU (two contexts on device 0)
TAG=${1:-r05c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 2400 python -m pytest tests -m gpu -q -x -rP --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/summary.txt
grep -h "headline parity\|config5:\|config3:\|guard sweep" $OUT/pytest_gpu.log >> $OUT/summary.txt
tail -25 $OUT/pytest_gpu.log >> $OUT/summary.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt; tail -1 $OUT/smoke.txt >> $OUT/summary.txt
timeout 900 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.txt
python - <<PY >> $OUT/summary.txt 2>&1
import json; d=json.load(open('$OUT/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'fit', d['gp_fit_ms'], d.get('gp_fit_ms_min'), d.get('gp_fit_data_resident_ms'), 'batched', d['gp_fit_batched'])
print('hyper', json.dumps(d.get('hyper_inference'))[:1500])
print('bo', json.dumps(d.get('bo_iteration'))[:800])
PY
for cfg in "headline --scaling strong" "c3 --steps 2 --warmup 1" "c4 --scaling strong --m 8192"; do
  timeout 600 python bench.py --gpus 2 --launcher inproc --devices 0,0 --no-cpu-baseline --lean --config $cfg >> $OUT/inproc.jsonl 2>> $OUT/inproc.err; echo "inproc $cfg rc=$?" >> $OUT/summary.txt
done
cut -c1-400 $OUT/inproc.jsonl >> $OUT/summary.txt
