#!/usr/bin/env python
"""Copy one GPU session's results (gpurun_out/<tag>/, written by tools/gpu_r05.sh) into profiles/<tag>_* -- the tracked
evidence set -- and refresh the PMC traffic files bench.py reads (profiles/trsm_traffic*.json) from THAT session's counter
passes; the `roofline.traffic` fields of the session's own bench lines are then patched to the session's own measurement
(they were filled from the previous set while the session ran; the patch is recorded in each line as `traffic_patched_from`).

    python tools/collect_evidence.py r05z
"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
commit = open(os.path.join(ROOT, ".git_head")).read().strip() if os.path.exists(os.path.join(ROOT, ".git_head")) else "unknown"

FILES = ["summary.txt", "bench.json", "bench_sustained.json", "bench_kernel_stats.csv", "prof_bench.json", "configs.jsonl",
         "c2_default_5_steps.json", "forcedist_weak.json", "forcedist_strong.json", "forcedist_c3.json", "spawn1_strong.json",
         "gpus2_on_one_gpu.err", "inproc.jsonl", "pmc_summary.txt", "pmc_summary_c2.txt", "pmc_summary_c3.txt",
         "pmc_summary_c4.txt", "pmc_summary_c5.txt", "pmc_summary_batch1.txt", "pmc_summary_batch3.txt", "pmc_summary_pair1.txt",
         "diag_timeline.txt", "trace_4096.txt", "small_m.txt", "smoke.txt", "batched_fit_ab.txt", "batch_trace_4_1_-1.txt",
         "batch_trace_4_3_-1.txt"]
for f in FILES:
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, f)))
if os.path.exists(os.path.join(src, "pytest_gpu.log")):
    shutil.copy(os.path.join(src, "pytest_gpu.log"), os.path.join(dst, "%s_pytest_gpu.txt" % tag))


def traffic(summary, kernel, config, n, d, m, out):
    p = os.path.join(src, summary)
    if not os.path.exists(p):
        return None
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"), p, commit, kernel, config,
                          str(n), str(d), str(m)], capture_output=True, text=True)
    if txt.returncode != 0:
        print("traffic", config, "failed:", txt.stderr[-300:])
        return None
    obj = json.loads(txt.stdout)
    obj["source"] = "profiles/%s_%s (rocprofv3 --kernel-trace --pmc <one group per pass>, bench.py --steps 1)" % (tag, summary)
    json.dump(obj, open(os.path.join(dst, out), "w"), indent=1)
    return obj


head = traffic("pmc_summary.txt", "trsm_step_gen_kernel", "headline", 4096, 16, 65536, "trsm_traffic.json")
others = {"c2": traffic("pmc_summary_c2.txt", "trsm_step_gen_kernel", "c2", 1024, 8, 65536, "trsm_traffic_c2.json"),
          "c3": traffic("pmc_summary_c3.txt", "trsm_step_gen_kernel", "c3", 2048, 16, 65536, "trsm_traffic_c3.json"),
          "c4": traffic("pmc_summary_c4.txt", "winv_row_kernel", "c4", 4096, 11, 8192, "trsm_traffic_c4.json"),
          "c5": traffic("pmc_summary_c5.txt", "trsm_step_kernel", "c5", 8192, 64, 131072, "trsm_traffic_c5.json")}
pair = traffic("pmc_summary_pair1.txt", "trsm_pair_gen_kernel", "headline", 4096, 16, 65536, "%s_trsm_traffic_two_rows_per_launch.json" % tag)


def patch(path, obj, jsonl=False, pick=None):
    if obj is None or not os.path.exists(path):
        return
    lines = [json.loads(l) for l in open(path) if l.strip()]
    for d in lines:
        cfg = pick(d) if pick else True
        t = obj if not isinstance(obj, dict) or "bytes_per_launch" in obj else obj.get(cfg)
        if t and isinstance(d.get("roofline"), dict) and t["kernel"] in str(d["roofline"].get("kernel", "")):
            d["roofline"]["traffic"] = t["bytes_per_launch"]
            d["roofline"]["traffic_patched_from"] = t["source"]
    with open(path, "w") as f:
        for d in lines:
            f.write(json.dumps(d) + "\n")


def which_config(d):
    w = d.get("config", {}).get("workload", "")
    for c in ("config 2", "config 3", "config 4", "config 5"):
        if c in w:
            return "c" + c[-1]
    return "c2" if d.get("config", {}).get("n_train") == 1024 else None


patch(os.path.join(dst, "%s_bench.json" % tag), head)
patch(os.path.join(dst, "%s_bench_sustained.json" % tag), head)
patch(os.path.join(dst, "%s_configs.jsonl" % tag), others, pick=which_config)
print("collected", tag, "at", commit, "->", len([f for f in os.listdir(dst) if f.startswith(tag + "_")]), "files")
if head:
    print("headline traffic per launch %.4g B, mfma busy %s" % (head["bytes_per_launch"], head.get("mfma_busy_fraction")))
if pair:
    print("two rows per launch: %.4g B per launch x %d launches" % (pair["bytes_per_launch"], pair["launches"]))
