#!/usr/bin/env python
"""Concurrency of kernels in a rocprofv3 --kernel-trace result (rocpd sqlite): how many kernels are resident at once,
time-weighted, over the steady-state part of the run.   python tools/rocprof_overlap.py x_results.db"""
import sqlite3
import sys

import numpy as np


def main(path):
    c = sqlite3.connect(path)
    rows = sorted(c.execute("select start, end, queue_id from kernels").fetchall())
    rows = rows[len(rows) // 2:]  # steady state: the multi-stream region comes last in bench.py
    ev = []
    for s, e, q in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    t_at = {}
    cur, last = 0, ev[0][0]
    for t, d in ev:
        t_at[cur] = t_at.get(cur, 0) + (t - last)
        cur += d
        last = t
    wall = sum(t_at.values())
    dur = np.array([e - s for s, e, q in rows], np.float64)
    print("kernels %d on %d queues; mean duration %.2f us; wall %.2f ms" % (len(rows), len(set(q for _, _, q in rows)), dur.mean() / 1e3, wall / 1e6))
    for k in sorted(t_at):
        print("  %d kernel(s) resident: %5.1f %% of the time" % (k, 100.0 * t_at[k] / wall))
    print("  time-weighted mean concurrency: %.2f" % (sum(k * v for k, v in t_at.items()) / wall))


if __name__ == "__main__":
    main(sys.argv[1])
