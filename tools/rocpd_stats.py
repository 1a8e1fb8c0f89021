#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) as a per-kernel stats table (CSV on stdout).

    python tools/rocpd_stats.py gpurun_out/<tag>/prof/bench_results.db > profiles/<name>_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    stats = {}
    for name, s, e in rows:
        d = stats.setdefault(name, [])
        d.append(e - s)
    total = sum(sum(v) for v in stats.values())
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for name, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (name.split("(")[0], len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                              100.0 * sum(v) / total))


if __name__ == "__main__":
    main(sys.argv[1])
