#!/bin/bash
# quick fit-kernel iteration on the GPU box: diag timeline, the fit-related parity tests, bench side figures
TAG=${1:-r02x}
OUT=gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp
python tools/diag_timeline.py > $OUT/diag_timeline.txt 2>&1
(timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_ref_parity.py -m gpu -q -x -k "golden or headline or ill_cond or multi_panel or fit_batch or batched or edge or gp_class or mcmc_chain or grad or error" > $OUT/pytest.log 2>&1; echo rc=$? >> $OUT/pytest.log)
timeout 300 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/diag_timeline.txt; tail -4 $OUT/pytest.log
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['gp_fit_ms'], d['gp_fit_phases_ms'], d['gp_fit_batched'], d['roofline']['frac'])"
