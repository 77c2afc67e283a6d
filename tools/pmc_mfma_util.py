#!/usr/bin/env python
"""Per-kernel MFMA utilisation table from a rocprofv3 --pmc pass (rocpd sqlite):
   python tools/pmc_mfma_util.py <results.db> "<description line>" > profiles/<name>.txt
Counters needed: SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE."""
import collections
import sqlite3
import sys


def main(path, desc):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, grid_size, workgroup_size, counter_name, sum(value), count(*), sum(duration) "
                     "from counters_collection group by kernel_name, grid_size, workgroup_size, counter_name").fetchall()
    agg = collections.OrderedDict()
    for k, g, wg, cn, v, n, d in rows:
        agg.setdefault((k, g, wg), {})[cn] = (v, n, d)
    print("# " + desc)
    print("# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs): share of ALL matrix pipes busy while the")
    print("#            kernel runs (64 busy cycles per v_mfma_f32_32x32x2_f32, summed over SIMDs; GRBM_GUI_ACTIVE is summed over 8 XCDs).")
    print("# Profiled passes serialise dispatches and clock lower: durations are longer than in the --kernel-trace --stats summaries.")
    print("%-36s %6s %5s %9s %9s %9s %9s" % ("kernel<BM,BN,BK,WR,WC,WK,PF>", "WGs", "n", "us/launch", "MfmaUtil", "WAIT_ANY", "WAIT_INST"))
    tm = tg = 0.0
    for (k, g, wg), d in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", (0,))[0]):
        if "conv_gemm" not in k:
            continue
        n = d["GRBM_GUI_ACTIVE"][1]
        gui = d["GRBM_GUI_ACTIVE"][0] / 8.0
        mf = d["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        wc = d["SQ_WAVE_CYCLES"][0]
        tm += mf
        tg += gui
        print("%-36s %6d %5d %9.1f %8.1f%% %8.1f%% %8.1f%%" % (k[k.index("<"):k.index(">") + 1], g // wg, n, d["GRBM_GUI_ACTIVE"][2] / n / 1e3,
                                                            100 * mf / (gui * 1024), 100 * d["SQ_WAIT_ANY"][0] / wc, 100 * d["SQ_WAIT_INST_ANY"][0] / wc))
    print("# all conv_gemm dispatches (conv1..conv5 + heads): MfmaUtil = %.1f%% of the 1024 matrix pipes over the kernels' own run time" % (100 * tm / (tg * 1024)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
