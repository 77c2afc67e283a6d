#!/usr/bin/env python3
"""LDS bank conflicts per kernel from a rocprofv3 PMC pass (SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array
cycles; MI355X_MICROARCH.md, LDS):

    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d out -o l -- python bench.py --streams 1 --no-graph ...
    python tools/pmc_lds_conflicts.py out/.../l_results.db

Round 5 found the Winograd kernel's fragment reads at 8 LDS cycles instead of 4 this way (and by the lane-group table of the guide)."""
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import kernel_names  # noqa: E402


def main(db):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    agg = {}
    for k, n, v, cnt in rows:
        a = agg.setdefault(kernel_names.label(k) if ("conv_gemm" in k or "wino" in k or "ws1x1" in k or "stem7x7" in k) else k[:90], {})
        a[n] = a.get(n, 0.0) + v
        a["_n"] = max(a.get("_n", 0), cnt)
    print("%-86s %10s %16s %16s %8s" % ("kernel", "dispatches", "LDS_IDX_ACTIVE", "BANK_CONFLICT", "share"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0)):
        act, conf = a.get("SQ_LDS_IDX_ACTIVE", 0.0), a.get("SQ_LDS_BANK_CONFLICT", 0.0)
        if act <= 0:
            continue
        print("%-86s %10d %16.0f %16.0f %7.1f%%" % (k, a["_n"], act, conf, 100.0 * conf / act))


if __name__ == "__main__":
    main(sys.argv[1])
