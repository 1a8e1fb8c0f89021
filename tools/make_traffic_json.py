#!/usr/bin/env python
"""profiles/trsm_traffic*.json (read by bench.py for roofline.traffic) from PMC summaries (tools/rocpd_pmc.py).

    python tools/make_traffic_json.py <pmc_summary.txt> <commit> [kernel-substring] [config] [n_train dim candidates]

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size (factor 2,
MI355X_MICROARCH.md section HBM); WRITE_SIZE is taken as reported.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 128
SIMD groups the counter is replicated on per GRBM_GUI_ACTIVE cycle (same normalisation as round 1).
"""
import json
import re
import sys


def main(path, commit, kernel="trsm_step_gen_kernel", config="headline", n=4096, d=16, m=65536):
    sums, rows = {}, {}
    for line in open(path):
        mt = re.match(r"\s*(\S+)\s+(\S+)\s+rows=(\d+)\s+sum=(\S+)", line)
        if mt and kernel in mt.group(1):
            sums[mt.group(2)] = sums.get(mt.group(2), 0.0) + float(mt.group(4))
            rows[mt.group(2)] = rows.get(mt.group(2), 0) + int(mt.group(3))
    nl = rows["FETCH_SIZE"]
    per_launch = (2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0 / nl
    out = {"kernel": kernel, "config": config,
           "workload": {"n_train": int(n), "dim": int(d), "candidates_per_gpu": int(m)},
           "launches": nl, "FETCH_SIZE_KB_sum": sums["FETCH_SIZE"], "WRITE_SIZE_KB_sum": sums["WRITE_SIZE"],
           "correction": "FETCH_SIZE x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md section HBM); "
                         "WRITE_SIZE as reported",
           "bytes_per_launch": per_launch,
           "commit": commit, "source": path + " (rocprofv3 --kernel-trace --pmc <one group per pass>, bench.py --steps 1)"}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in sums and "GRBM_GUI_ACTIVE" in sums:
        out["SQ_VALU_MFMA_BUSY_CYCLES_sum"] = sums["SQ_VALU_MFMA_BUSY_CYCLES"]
        out["GRBM_GUI_ACTIVE_sum"] = sums["GRBM_GUI_ACTIVE"]
        out["mfma_busy_fraction"] = sums["SQ_VALU_MFMA_BUSY_CYCLES"] / sums["GRBM_GUI_ACTIVE"] / 128.0
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
