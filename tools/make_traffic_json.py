#!/usr/bin/env python
"""profiles/trsm_traffic.json (read by bench.py for roofline.traffic) from a PMC summary of tools/gpu_round2.sh.

    python tools/make_traffic_json.py gpurun_out/<tag>/pmc_summary.txt <commit> > profiles/trsm_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size (factor 2,
MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as reported.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 128
SIMD groups the counter is replicated on per GRBM_GUI_ACTIVE cycle (same normalisation as round 1).
"""
import json
import re
import sys


def main(path, commit):
    sums, rows = {}, {}
    for line in open(path):
        m = re.match(r"\s*(\S+)\s+(\S+)\s+rows=(\d+)\s+sum=(\S+)", line)
        if m and "trsm_step_gen_kernel" in m.group(1):
            sums[m.group(2)] = float(m.group(4))
            rows[m.group(2)] = int(m.group(3))
    n = rows["FETCH_SIZE"]
    per_launch = (2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0 / n
    out = {"kernel": "trsm_step_gen_kernel", "workload": {"n_train": 4096, "dim": 16, "candidates_per_gpu": 65536},
           "launches": n, "FETCH_SIZE_KB_sum": sums["FETCH_SIZE"], "WRITE_SIZE_KB_sum": sums["WRITE_SIZE"],
           "correction": "FETCH_SIZE x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md section HBM); "
                         "WRITE_SIZE as reported",
           "bytes_per_launch": per_launch,
           "SQ_VALU_MFMA_BUSY_CYCLES_sum": sums.get("SQ_VALU_MFMA_BUSY_CYCLES"),
           "GRBM_GUI_ACTIVE_sum": sums.get("GRBM_GUI_ACTIVE"),
           "mfma_busy_fraction": sums["SQ_VALU_MFMA_BUSY_CYCLES"] / sums["GRBM_GUI_ACTIVE"] / 128.0,
           "commit": commit, "source": path + " (rocprofv3 --kernel-trace --pmc <one group per pass>, bench.py --steps 1)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
