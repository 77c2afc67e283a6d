#!/usr/bin/env python
"""Aggregate a bench.py --breakdown file (plan text + per-launch hipEvent table) by GEMM shape."""
import collections
import sys


def main(path, top=40):
    rows, plan = [], {}
    for ln in open(path):
        f = ln.rstrip("\n").split("\t")
        if ln.startswith("#") or f[0] == "idx":
            continue
        if len(f) == 4:
            plan[int(f[0])] = f[2]
        if len(f) == 7:
            rows.append((int(f[0]), f[1], float(f[2]), float(f[3]), f[6]))
    agg = collections.OrderedDict()
    for i, k, us, gf, lab in rows:
        a = agg.setdefault((k, plan.get(i, "")), [0, 0.0, 0.0, lab])
        a[0] += 1
        a[1] += us
        a[2] += gf
    tot = sum(a[1] for a in agg.values())
    print("# total %.1f us, %.1f GFLOP -> %.1f TFLOP/s" % (tot, sum(a[2] for a in agg.values()), sum(a[2] for a in agg.values()) / tot * 1e3))
    for (k, p), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%5.1f%% %7.1fus n=%2d avg %6.1fus %6.1f TF/s  %s %s  e.g. %s" % (
            100 * a[1] / tot, a[1], a[0], a[1] / a[0], a[2] / a[1] * 1e3 if a[1] else 0, k, p, a[3][:36]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
