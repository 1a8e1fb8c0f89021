#!/bin/bash
# One GPU-box session of round 6.  usage: tools/gpu_r06.sh <tag> [tests] [smoke] [bench] [sustained] [configs] [dist] [inproc] [prof] [pmc] [pmcconf] [pmcbatch] [fit] [small] [batched]
TAG=${1:-r06}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for what in "$@"; do case $what in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -x -rP --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/summary.txt
  grep -h "headline parity\|config5:\|guard sweep" $OUT/pytest_gpu.log >> $OUT/summary.txt
  tail -30 $OUT/pytest_gpu.log >> $OUT/summary.txt ;;
smoke)
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt ;;
bench)
  timeout 600 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.txt
  python - <<PY >> $OUT/summary.txt 2>&1
import json; d=json.load(open('$OUT/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'fit median/min', d['gp_fit_ms'], d.get('gp_fit_ms_min'), 'resident', d.get('gp_fit_data_resident_ms'), d.get('gp_fit_data_resident_ms_min'), d['gp_fit_phases_ms'], 'k1', d['k_assembly'], 'small', d['small_batch_latency_ms'], 'batched', d['gp_fit_batched'])
h = d.get('hyper_inference', {})
for k in ('n_train_2048', 'n_train_4096'):
    if k in h: print('hyper', k, {kk: h[k].get(kk) for kk in ('first_iteration_ms', 'later_iteration_ms', 'likelihoods_per_s', 'frac_of_fp64_mfma_peak')}, 'cpu ms/likelihood', (h[k].get('cpu_port') or {}).get('ms_per_likelihood'))
print('hyper n=200', {kk: h.get(kk) for kk in ('first_iteration_ms', 'later_iteration_ms', 'likelihoods_per_s')})
print('bo_iteration', d.get('bo_iteration', {}).get('ms'), 'gp_mcmc_n2048', d.get('bo_iteration', {}).get('gp_mcmc_n2048', {}).get('ms'))
PY
  ;;
configs)
  : > $OUT/configs.jsonl
  # c2 / c4 steps last 1.4 / 5.4 ms: the default 5 timed steps (7 / 27 ms behind 2 warm-up steps) end before the part has
  # left its idle power state (c2: 0.59 of peak over 5 steps, 0.65 over 20, 0.69 over 100) -- sustained runs for those two
  for c in "c2" "c3" "c4" "c5"; do          # (c2 / c4 default to 100 / 50 timed steps: sustained rates)
    timeout 900 python bench.py --gpus 1 --config $c >> $OUT/configs.jsonl 2>> $OUT/configs.err; echo "$c rc=$?" >> $OUT/summary.txt
  done
  timeout 300 python bench.py --gpus 1 --config c2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/c2_default_5_steps.json 2>> $OUT/configs.err
  timeout 900 python bench.py --gpus 1 --config c5 --m 1048576 --steps 2 --warmup 1 --no-cpu-baseline >> $OUT/configs.jsonl 2>> $OUT/configs.err; echo "c5 full rc=$?" >> $OUT/summary.txt
  cut -c1-300 $OUT/configs.jsonl >> $OUT/summary.txt ;;
dist)
  # the RCCL path on ONE rank, both scalings (torchrun-style process group) + the self-launcher's one-rank form (no torch)
  for sc in weak strong; do
    ROBO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --scaling $sc --no-cpu-baseline --lean > $OUT/forcedist_$sc.json 2> $OUT/forcedist_$sc.err; echo "forcedist $sc rc=$?" >> $OUT/summary.txt
  done
  RDV=$(mktemp -d); RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 ROBO_BENCH_RENDEZVOUS=$RDV timeout 600 python bench.py --gpus 1 --scaling strong --no-cpu-baseline --lean > $OUT/spawn1_strong.json 2> $OUT/spawn1.err; echo "spawn1 rc=$?" >> $OUT/summary.txt
  ROBO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --config c3 --no-cpu-baseline > $OUT/forcedist_c3.json 2>> $OUT/forcedist.err; echo "forcedist c3 rc=$?" >> $OUT/summary.txt
  # what --gpus 2 does on a one-GPU box: both ranks fail or run on device 0/1 -- must end non-zero, not hang
  timeout 300 python bench.py --gpus 2 --lean --no-cpu-baseline > $OUT/gpus2_on_one_gpu.json 2> $OUT/gpus2_on_one_gpu.err; echo "gpus2-on-1-gpu rc=$? (expected non-zero)" >> $OUT/summary.txt
  cut -c1-260 $OUT/forcedist_weak.json $OUT/forcedist_strong.json $OUT/spawn1_strong.json $OUT/forcedist_c3.json >> $OUT/summary.txt; tail -3 $OUT/forcedist_strong.err >> $OUT/summary.txt ;;
prof)
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --gpus 1 --no-cpu-baseline --lean > $OUT/prof_bench.json 2> $OUT/prof.err; echo "rocprof rc=$?" >> $OUT/summary.txt
  python tools/rocpd_stats.py $OUT/prof/bench_results.db > $OUT/bench_kernel_stats.csv 2>> $OUT/prof.err
  find $OUT/prof -size +20M -delete
  head -16 $OUT/bench_kernel_stats.csv >> $OUT/summary.txt ;;
pmc)
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc/$name -o pmc -- python bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --lean > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
    echo "pmc $C rc=$?" >> $OUT/summary.txt
  done
  python tools/rocpd_pmc.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
  find $OUT/pmc -size +30M -delete
  grep -i "trsm_step\|potrf_step\|gram_kernel\|potrf_panel" $OUT/pmc_summary.txt | head -24 >> $OUT/summary.txt ;;
pmcconf)
  # HBM traffic of the dominant kernel of the other configurations: FETCH_SIZE and WRITE_SIZE passes, one step each
  for cfg in "c2 trsm_step_gen_kernel 1024 8 65536" "c3 trsm_step_gen_kernel 2048 16 65536" "c4 winv_row_kernel 4096 11 8192" "c5 trsm_step_kernel 8192 64 131072"; do
    set -- $cfg
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$1/$C -o pmc -- python bench.py --gpus 1 --config $1 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$1_$C.err
      echo "pmc $1 $C rc=$?" >> $OUT/summary.txt
    done
    python tools/rocpd_pmc.py $OUT/pmc_$1 > $OUT/pmc_summary_$1.txt 2>&1
    python tools/make_traffic_json.py $OUT/pmc_summary_$1.txt $(cat .git_head 2>/dev/null || echo unknown) $2 $1 $3 $4 $5 > $OUT/trsm_traffic_$1.json 2>> $OUT/summary.txt
    find $OUT/pmc_$1 -size +30M -delete
    grep bytes_per_launch $OUT/trsm_traffic_$1.json >> $OUT/summary.txt
  done ;;
fit)
  python tools/diag_timeline.py > $OUT/diag_timeline.txt 2>&1
  timeout 600 bash tools/gpu_fit_trace.sh $TAG 4096 > $OUT/fit_trace.log 2>&1
  cat $OUT/diag_timeline.txt >> $OUT/summary.txt; tail -3 $OUT/trace_4096.txt >> $OUT/summary.txt 2>/dev/null ;;
sustained)
  # the headline under steady power: 200 timed steps, the shader clock sampled during one more step at the end
  timeout 600 python bench.py --gpus 1 --steps 200 --warmup 5 --lean --no-cpu-baseline > $OUT/bench_sustained.json 2> $OUT/bench_sustained.err; echo "sustained rc=$?" >> $OUT/summary.txt
  python - <<PY >> $OUT/summary.txt 2>&1
import json; d=json.load(open('$OUT/bench_sustained.json'))
print('sustained: value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])
PY
  ;;
k1pmc)
  # K1 alone: VALU instruction counts and busy cycles of the gram kernel over 4 headline fits (separate passes per group)
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $C | tr ' ' '_')
    FIT_REPS=4 timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/k1pmc/$name -o pmc -- python tools/fit_only.py > $OUT/k1pmc_$name.txt 2> $OUT/k1pmc_$name.err
    echo "k1pmc $C rc=$?" >> $OUT/summary.txt
  done
  python tools/rocpd_pmc.py $OUT/k1pmc > $OUT/k1pmc_summary.txt 2>&1
  find $OUT/k1pmc -size +30M -delete
  grep -i "gram_kernel" $OUT/k1pmc_summary.txt >> $OUT/summary.txt ;;
inproc)
  # ONE process driving several contexts (robo_amd/csrc/multi.hip).  A one-GPU box has one device: two contexts on device 0
  # exercise the worker threads, peer copies and reductions on hardware; the scaling itself needs a multi-GPU node
  : > $OUT/inproc.jsonl
  for cfg in "headline --scaling strong" "c3 --steps 2 --warmup 1" "c4 --scaling strong --m 8192" "c5 --scaling strong --m 131072 --steps 2 --warmup 1"; do
    timeout 900 python bench.py --gpus 2 --launcher inproc --devices 0,0 --no-cpu-baseline --lean --config $cfg >> $OUT/inproc.jsonl 2>> $OUT/inproc.err; echo "inproc $cfg rc=$?" >> $OUT/summary.txt
  done
  timeout 600 python bench.py --gpus 1 --launcher inproc --no-cpu-baseline --lean >> $OUT/inproc.jsonl 2>> $OUT/inproc.err; echo "inproc 1 device rc=$?" >> $OUT/summary.txt
  cut -c1-330 $OUT/inproc.jsonl >> $OUT/summary.txt ;;
batched)
  # the batched factorisation: one stream vs three sub-batch streams, same session, several N
  : > $OUT/batched_fit_ab.txt
  for cfg in "4096 16 27 7" "3072 16 26 7" "2048 16 26 9" "1536 16 26 9" "1024 8 26 9"; do
    BATCH_AB="4,1,-1;4,3,-1;0,3,-1" timeout 600 python tools/batched_fit_ab.py $cfg >> $OUT/batched_fit_ab.txt 2>&1
  done
  grep "batched fit\|round 1" $OUT/batched_fit_ab.txt >> $OUT/summary.txt
  for v in "4,1,-1" "4,3,-1"; do
    name=$(echo $v | tr ',' '_')
    BATCH_AB="$v" timeout 300 rocprofv3 --kernel-trace -d $OUT/batch_trace_$name -o t -- python tools/batched_fit_ab.py 4096 16 27 2 > $OUT/batch_trace_$name.log 2>&1
    python tools/batch_trace.py $OUT/batch_trace_$name 400 > $OUT/batch_trace_$name.txt 2>&1
    find $OUT/batch_trace_$name -size +5M -delete
    head -1 $OUT/batch_trace_$name.txt >> $OUT/summary.txt
  done ;;
pmcbatch)
  # MFMA busy of the batched factorisation's kernels (27 thetas, N = 4096), one stream and three
  for sp in 1 3; do
    BATCH_AB="4,$sp,-1" timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_batch$sp/busy -o pmc -- python tools/batched_fit_ab.py 4096 16 27 2 > /dev/null 2> $OUT/pmc_batch$sp.err
    echo "pmcbatch split=$sp rc=$?" >> $OUT/summary.txt
    python tools/rocpd_pmc.py $OUT/pmc_batch$sp > $OUT/pmc_summary_batch$sp.txt 2>&1
    find $OUT/pmc_batch$sp -size +30M -delete
    grep -i "potrf_step\|potrf_panel\|potrf_diag" $OUT/pmc_summary_batch$sp.txt | head -12 >> $OUT/summary.txt
  done ;;
pairpmc)
  # the block-row solve with two rows per launch (trsm_pair = 1): timing beside the default and HBM traffic of both
  for rep in 1 2; do for pair in 0 1; do
    ROBO_TRSM_PAIR=$pair timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --lean --no-cpu-baseline > $OUT/bench_pair${pair}_$rep.json 2>> $OUT/bench_pair.err
    python -c "
import json; d=json.load(open('$OUT/bench_pair${pair}_$rep.json')); print('trsm_pair=$pair rep $rep: ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['kernel'], 'argmax', d['argmax'])" >> $OUT/summary.txt
  done; done
  for C in FETCH_SIZE WRITE_SIZE; do
    ROBO_TRSM_PAIR=1 timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_pair1/$C -o pmc -- python bench.py --gpus 1 --steps 1 --warmup 0 --lean --no-cpu-baseline > /dev/null 2> $OUT/pmc_pair1_$C.err
  done
  python tools/rocpd_pmc.py $OUT/pmc_pair1 > $OUT/pmc_summary_pair1.txt 2>&1
  find $OUT/pmc_pair1 -size +30M -delete
  python tools/make_traffic_json.py $OUT/pmc_summary_pair1.txt $(cat .git_head 2>/dev/null || echo unknown) trsm_pair_gen_kernel headline 4096 16 65536 > $OUT/trsm_traffic_pair1.json 2>> $OUT/summary.txt
  grep -h "bytes_per_launch\|launches" $OUT/trsm_traffic_pair1.json >> $OUT/summary.txt ;;
trace2048)
  # the batched factorisation where the default front end lives: 26 thetas at N = 2048 (one ensemble half-step), kernel trace
  for v in "0,3,-1" "0,1,-1"; do
    name=$(echo $v | tr ',' '_')
    BATCH_AB="$v" timeout 300 rocprofv3 --kernel-trace -d $OUT/batch2048_$name -o t -- python tools/batched_fit_ab.py 2048 16 26 2 > $OUT/batch2048_$name.log 2>&1
    python tools/batch_trace.py $OUT/batch2048_$name 400 > $OUT/batch2048_trace_$name.txt 2>&1
    find $OUT/batch2048_$name -size +5M -delete
    head -1 $OUT/batch2048_trace_$name.txt >> $OUT/summary.txt
  done
  BATCH_AB="0,3,-1;0,1,-1" timeout 300 python tools/batched_fit_ab.py 2048 16 26 7 > $OUT/batch2048_ab.txt 2>&1; grep "round 1" $OUT/batch2048_ab.txt >> $OUT/summary.txt ;;
storefloor)
  # what a pure store stream of K1's size costs on this part (torch fill of 67.6 MB, HIP events): the floor under gram_kernel
  timeout 300 python - > $OUT/storefloor.txt 2>&1 <<PY
import torch
n = 67649536 // 8
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n // 2, dtype=torch.float64, device="cuda")
for name, t in (("fill 67.6 MB", x), ("fill 33.8 MB (fp32-sized K)", y)):
    for _ in range(5): t.fill_(1.0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); t.fill_(2.0); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print("%s: median %.1f us min %.1f us -> %.2f TB/s" % (name, ts[len(ts)//2], ts[0], t.numel() * 8 / ts[len(ts)//2] / 1e6))
PY
  cat $OUT/storefloor.txt >> $OUT/summary.txt ;;
follow)
  # the follower form of the single-theta factorisation: hardware parity test first (hand-off across XCDs), then the A/B
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "panel_followers" > $OUT/follow_test.log 2>&1; echo "follow test rc=$?" >> $OUT/summary.txt
  tail -5 $OUT/follow_test.log >> $OUT/summary.txt
  for cfg in "4096 16" "2048 16" "1024 8" "3000 16"; do
    FOLLOW_FROM="-1" FOLLOW_ROWS="64,128,-1" timeout 300 python tools/follow_ab.py $cfg 15 >> $OUT/follow_ab.txt 2>&1
  done
  FOLLOW_FROM="0,8,16,31" FOLLOW_ROWS="64,128,-1" timeout 600 python tools/follow_ab.py 8192 64 9 >> $OUT/follow_ab.txt 2>&1
  grep "single-theta\|round 1" $OUT/follow_ab.txt >> $OUT/summary.txt
  for f in 0 1; do
    ROBO_POTRF_FOLLOW=$f ROBO_POTRF_FOLLOW_FROM=-1 ROBO_POTRF_FOLLOW_ROWS=-1 FIT_N=4096 FIT_REPS=4 timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_follow$f -o fit -- python tools/fit_only.py > $OUT/prof_follow$f.log 2>&1
    python tools/fit_trace.py $OUT/prof_follow$f > $OUT/trace_4096_follow$f.txt 2>&1
    find $OUT/prof_follow$f -size +5M -delete
    tail -2 $OUT/trace_4096_follow$f.txt >> $OUT/summary.txt
  done ;;
k1ab)
  # K1 (gram_kernel) between builds: robo_amd/librobo_hip.so against every robo_amd/librobo_hip_*.so variant shipped beside it
  timeout 600 python tools/k1_ab.py default $(ls robo_amd/librobo_hip_*.so | grep -v diag) > $OUT/k1_ab.txt 2>&1; cat $OUT/k1_ab.txt >> $OUT/summary.txt
  timeout 300 python tools/k1_ab.py default $(ls robo_amd/librobo_hip_*.so | grep -v diag) --n 2048 --d 16 > $OUT/k1_ab_2048.txt 2>&1; grep "round 1" $OUT/k1_ab_2048.txt >> $OUT/summary.txt ;;
chain)
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "device_resident_chain or mcmc" > $OUT/chain_test.log 2>&1; echo "chain tests rc=$?" >> $OUT/summary.txt; tail -3 $OUT/chain_test.log >> $OUT/summary.txt
  timeout 600 python tools/chain_ab.py > $OUT/chain_ab.txt 2>&1; cat $OUT/chain_ab.txt >> $OUT/summary.txt ;;
bfollow)
  # batched fits: diagonal block + panel in one launch (potrf_batch_follow) against the launch-per-phase form, one and three streams
  : > $OUT/bfollow_ab.txt
  for cfg in "2048 16 26 7" "4096 16 27 5" "1024 8 26 9" "3072 16 26 5"; do
    for bf in 0 1; do
      BATCH_TUNE="potrf_batch_follow=$bf" BATCH_AB="0,3,-1;0,1,-1;0,2,-1" timeout 600 python tools/batched_fit_ab.py $cfg 2>&1 | sed "s/^round/follow=$bf round/" >> $OUT/bfollow_ab.txt
    done
  done
  grep "batched fit\|round 1" $OUT/bfollow_ab.txt >> $OUT/summary.txt ;;
broll)
  # batched merged launch: 150-KB image against the 80-KB rolling layout (two workgroups per CU), against the launch-per-phase form
  : > $OUT/broll_ab.txt
  for cfg in "2048 16 26 7" "4096 16 27 5" "1024 8 26 9" "3072 16 26 5"; do
    for tune in "potrf_batch_follow=0" "potrf_batch_follow=1,potrf_batch_roll=0" "potrf_batch_follow=1,potrf_batch_roll=1"; do
      BATCH_TUNE="$tune" BATCH_AB="0,3,-1;0,1,-1" timeout 600 python tools/batched_fit_ab.py $cfg 2>&1 | sed "s/^round/$tune round/" >> $OUT/broll_ab.txt
    done
  done
  grep "batched fit\|round 1" $OUT/broll_ab.txt >> $OUT/summary.txt ;;
small)
  timeout 300 python tools/small_m_timing.py > $OUT/small_m.txt 2>&1; cat $OUT/small_m.txt >> $OUT/summary.txt ;;
*) echo "unknown step $what" >> $OUT/summary.txt ;;
esac; done
echo "commit $(cat .git_head 2>/dev/null || echo unknown)" >> $OUT/summary.txt
cat $OUT/summary.txt
