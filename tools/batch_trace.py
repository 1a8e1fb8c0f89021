#!/usr/bin/env python
"""Timeline of the LAST batched factorisation in a rocprofv3 rocpd database (kernel trace): which launches of the
sub-batch streams overlapped.  Prints the time the chip spent with (a) only latency-phase kernels resident (diagonal
blocks, panels, 32-row column updates), (b) at least one chip-wide trailing update resident, (c) two or more kernels
resident, and a compact per-launch listing (first 80 launches)."""
import glob
import sqlite3
import sys


def main(path, show=80):
    db = glob.glob(path + "/**/*results.db", recursive=True)[0]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    extra = [x for x in ("queue_id", "stream_id", "grid_size_x", "grid_size_y", "workgroup_size_x") if x in cols]
    q = "select s.kernel_name, d.start, d.end%s from %s d join %s s on d.kernel_id=s.id order by d.start" % (
        "".join(", d." + x for x in extra), kd, ks)
    rows = c.execute(q).fetchall()
    last = max(i for i, r in enumerate(rows) if "gram_" in r[0] and "cross" not in r[0])
    rows = rows[last:]
    t0 = rows[0][1]
    ev = []
    for r in rows:
        n = r[0]
        big = "potrf_step_kernel<4" in n
        ev.append((r[1], 1, big))
        ev.append((r[2], -1, big))
    ev.sort()
    active = nbig = 0
    prev = t0
    only_small = with_big = multi = idle = 0
    for t, d, big in ev:
        dt = t - prev
        if active == 0: idle += dt
        elif nbig == 0: only_small += dt
        else: with_big += dt
        if active >= 2: multi += dt
        active += d
        if big: nbig += d
        prev = t
    total = rows[-1][2] - t0
    print("launches %d, wall %.1f us: idle %.1f, only latency-phase kernels resident %.1f, a chip-wide update resident %.1f; "
          ">= 2 kernels resident %.1f" % (len(rows), total / 1e3, idle / 1e3, only_small / 1e3, with_big / 1e3, multi / 1e3))
    print("columns: name start dur", " ".join(extra))
    for r in rows[:show]:
        short = r[0].split("(")[0].split("::")[-1][:34]
        print("%-34s %9.1f %8.1f  %s" % (short, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, " ".join(str(x) for x in r[3:])))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80)
