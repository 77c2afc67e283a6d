#!/usr/bin/env python
"""Per kernel (name, grid): HBM-side bytes per dispatch from the FETCH_SIZE / WRITE_SIZE PMC passes (rocpd sqlite), with the
gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x2, KB units).   python tools/pmc_per_shape.py fetch.db write.db"""
import re
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_names  # noqa: E402


def table(path, counter):
    c = sqlite3.connect(path)
    out = {}
    for name, grid, wg, v, n in c.execute(
            "select kernel_name, grid_size, workgroup_size, sum(value), count(*) from counters_collection where counter_name = ? "
            "group by kernel_name, grid_size, workgroup_size", (counter,)):
        out[(name, grid // max(wg, 1))] = (v / n, n)
    return out


def short(name):
    return kernel_names.label(name)


def main(fdb, wdb):
    f, w = table(fdb, "FETCH_SIZE"), table(wdb, "WRITE_SIZE")
    rows = []
    for k in f:
        fb = 2.0 * f[k][0] * 1024
        wb = w.get(k, (0, 0))[0] * 1024
        rows.append((f[k][1] * (fb + wb), k, fb, wb, f[k][1]))
    tot = sum(r[0] for r in rows)
    print("# share  n   fetch MB  write MB  kernel  workgroups")
    for t, k, fb, wb, n in sorted(rows, reverse=True)[:30]:
        print("%5.1f%% %4d %8.2f %8.2f  %s  %d" % (100 * t / tot, n, fb / 1e6, wb / 1e6, short(k[0]), k[1]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
