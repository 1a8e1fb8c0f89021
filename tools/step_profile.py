#!/usr/bin/env python
"""Per-block-row durations of the posterior step kernel from a rocprofv3 rocpd database:
median over predict calls of the duration of launch i, and the fit  t_i = a + b * i."""
import glob
import sqlite3
import sys

import numpy as np


def main(path, nb=32, pat="trsm_step"):
    db = glob.glob(path + "/**/*results.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start"
                     % (kd, ks)).fetchall()
    d = np.array([(e - s) / 1e3 for n, s, e in rows if pat in n])
    d = d[-(len(d) // nb) * nb:].reshape(-1, nb)
    m = np.median(d, axis=0)
    print("us per block row:", np.round(m, 1).tolist())
    i = np.arange(nb)
    ideal = 2.0 * 128 * 128 * 128 * 512 / 78.6e12 * 1e6
    print("sum %.1f us; ideal per 128-column block %.2f us" % (m.sum(), ideal))
    print("overhead over ideal:", np.round(m - ideal * (i + 1), 0).tolist())


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:3]))
