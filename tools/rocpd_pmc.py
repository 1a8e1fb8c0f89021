#!/usr/bin/env python
"""Per-kernel PMC sums from the rocpd databases written by tools/gpu_pmc.sh."""
import glob
import os
import sqlite3
import sys


def main(root):
    for db_path in sorted(glob.glob(os.path.join(root, "*", "*.db"))):
        db = sqlite3.connect(db_path)
        cur = db.cursor()
        views = [r[0] for r in cur.execute("select name from sqlite_master where type='view'")]
        print("==", db_path)
        if "counters_collection" not in views:
            print("  no counters_collection view:", views)
            continue
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
        ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
        vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
        q = "select %s, %s, count(*), sum(%s) from counters_collection group by 1, 2 order by 4 desc" % (kcol, ccol, vcol)
        for k, c, n, v in cur.execute(q):
            print("  %-28s %-28s rows=%-6d sum=%.6g" % (k.split("(")[0][-28:], c, n, v))


if __name__ == "__main__":
    main(sys.argv[1])
