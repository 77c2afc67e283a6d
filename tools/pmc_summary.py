#!/usr/bin/env python
"""Summarise rocprofv3 --pmc results (rocpd sqlite) per kernel name: sums of each counter."""
import collections
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    q = """select k.name, p.name, sum(e.value), count(*) from pmc_events e
           join pmc_info p on e.pmc_id = p.id join kernels k on e.event_id = k.event_id group by k.name, p.name"""
    try:
        rows = c.execute(q).fetchall()
    except Exception as ex:  # schema differences: dump what exists
        print("query failed:", ex)
        for t in ("pmc_events", "pmc_info", "kernels", "counters_collection"):
            print(t, [d[1] for d in c.execute("pragma table_info('%s')" % t)])
        return
    agg = collections.OrderedDict()
    for k, cn, v, n in rows:
        agg.setdefault(k, {})[cn] = (v, n)
    for k, d in agg.items():
        print(k[:100])
        for cn, (v, n) in sorted(d.items()):
            print("    %-32s sum=%16.0f  n=%d  avg=%14.1f" % (cn, v, n, v / n))


if __name__ == "__main__":
    main(sys.argv[1])
