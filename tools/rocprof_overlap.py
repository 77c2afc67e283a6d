#!/usr/bin/env python
"""Concurrency of kernels in a rocprofv3 --kernel-trace result (rocpd sqlite): how many kernels are resident at once,
time-weighted, over the steady-state part of the run.   python tools/rocprof_overlap.py x_results.db"""
import sqlite3
import sys

import numpy as np


def main(path):
    c = sqlite3.connect(path)
    rows = sorted(c.execute("select start, end, queue_id from kernels").fetchall())
    # the region with several forwards in flight = the time span in which the queues of the EXTRA streams carry kernels
    # (everything else bench.py runs — the one-at-a-time region, the host-buffer paths — stays on the busiest queue)
    per_q = {}
    for s, e, q in rows:
        per_q[q] = per_q.get(q, 0) + 1
    main_q = max(per_q, key=per_q.get)
    extra = [(s, e) for s, e, q in rows if q != main_q]
    if extra:
        # the longest contiguous burst on the extra queues (warm-up and timed region run back to back; gaps of more than 5 ms
        # separate bursts)
        extra.sort()
        bursts, cur = [], [extra[0]]
        for se in extra[1:]:
            if se[0] - max(e for _, e in cur[-8:]) < 5e6:
                cur.append(se)
            else:
                bursts.append(cur)
                cur = [se]
        bursts.append(cur)
        best = max(bursts, key=len)  # the bursts of a few kernels are stray copies / fills on other queues
        t0, t1 = best[0][0], max(e for _, e in best)
        rows = [(max(s, t0), min(e, t1), q) for s, e, q in rows if e > t0 and s < t1]
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
