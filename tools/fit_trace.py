#!/usr/bin/env python
"""Per-launch durations and gaps of the LAST fit in a rocprofv3 rocpd database (kernel trace):
one line per step = panel kernel, step kernel, and the idle gaps in front of each."""
import glob
import sqlite3
import sys

import numpy as np


def main(path):
    db = glob.glob(path + "/**/*results.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start"
                     % (kd, ks)).fetchall()
    last = max(i for i, r in enumerate(rows) if "gram_" in r[0] and "cross" not in r[0])
    rows = rows[last:]
    t0 = rows[0][1]
    prev = rows[0][1]
    steps, panels, gaps = [], [], []
    for n, s, e in rows:
        short = n.split("(")[0].split("::")[-1][:28]
        gap = (s - prev) / 1e3
        print("%-28s start %8.1f  dur %6.1f  gap %5.1f" % (short, (s - t0) / 1e3, (e - s) / 1e3, gap))
        if "potrf_step" in n: steps.append((e - s) / 1e3)
        if "potrf_panel" in n: panels.append((e - s) / 1e3)
        gaps.append(gap)
        prev = e
    print("total %.1f us; step kernels %.1f, panels %.1f, gaps %.1f" %
          ((rows[-1][2] - t0) / 1e3, sum(steps), sum(panels), sum(gaps)))
    print("steps:", np.round(steps, 1).tolist())


if __name__ == "__main__":
    main(sys.argv[1])
