#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace result (rocpd sqlite):
python tools/rocprof_gaps.py x_results.db   -> distribution of start(i+1) - end(i) over the steady-state part."""
import sqlite3
import sys

import numpy as np


def main(path):
    c = sqlite3.connect(path)
    rows = sorted(c.execute("select start, end, name from kernels").fetchall())
    rows = rows[len(rows) // 3:]  # skip tuning / warm-up
    st = np.array([r[0] for r in rows], np.int64)
    en = np.array([r[1] for r in rows], np.int64)
    gaps = (st[1:] - en[:-1]) / 1e3
    dur = (en - st) / 1e3
    inside = gaps[(gaps > -50) & (gaps < 50)]  # between forwards the host is in the way
    print("kernels %d  mean duration %.2f us" % (len(rows), dur.mean()))
    print("gap between consecutive kernels (us): mean %.2f  median %.2f  p10 %.2f  p90 %.2f  (n=%d)"
          % (inside.mean(), np.median(inside), np.percentile(inside, 10), np.percentile(inside, 90), len(inside)))
    print("share of wall time idle between kernels: %.1f %%" % (100.0 * inside.sum() / (inside.sum() + dur.sum())))


if __name__ == "__main__":
    main(sys.argv[1])
