#!/usr/bin/env python
"""Per-kernel MFMA utilisation table from a rocprofv3 --pmc pass (rocpd sqlite):
   python tools/pmc_mfma_util.py <results.db> "<description line>" [plan.txt [bench.json]] > profiles/<name>.txt
With a plan file (bench.py --breakdown output, or Net.plan_text()) a second table groups the launches by network
stage (conv1 / res2 / res3 / res4 / res5 / heads): dispatches are matched to plan lines by their order in a forward.
Counters needed: SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE."""
import collections
import sqlite3
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_names  # noqa: E402
import sys


def stage_of(label):
    for key, name in (("res2", "res2 (conv2_x)"), ("res3d", "heads"), ("res3", "res3 (conv3_x)"), ("res4", "res4 (conv4_x)"),
                      ("res5c_up", "heads"), ("res5", "res5 (conv5_x)"), ("conv1", "conv1")):
        if label.startswith(key):
            return name
    return "heads"


def per_stage(c, plan_path):
    labels, unprof = [], []
    for ln in open(plan_path):
        f = ln.rstrip("\n").split("\t")
        if len(f) == 4 and f[0].isdigit() and ("conv_gemm" in f[1] or "wino_f23<" in f[1] or "wino_h23<" in f[1] or "ws1x1<" in f[1] or "ws1x1f<" in f[1] or "ws7x7f<" in f[1] or "stem7x7<" in f[1]):  # conv_gemm< and conv_gemm_mp< (group plans)
            labels.append(f[3])
        if len(f) == 7 and f[0].isdigit() and ("conv_gemm" in f[1] or "wino_f23<" in f[1] or "wino_h23<" in f[1] or "ws1x1<" in f[1] or "ws1x1f<" in f[1] or "ws7x7f<" in f[1] or "stem7x7<" in f[1]):
            unprof.append(float(f[2]))  # bench.py --breakdown: hipEvent microseconds of this launch, nothing profiled
    n = len(labels)
    rows = c.execute("select dispatch_id, counter_name, value, duration from counters_collection where (kernel_name like '%conv_gemm%' or kernel_name like '%wino__23%' or kernel_name like '%ws1x1%' or kernel_name like '%stem7x7%') "
                     "and counter_name in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES') order by dispatch_id").fetchall()
    ids = sorted(set(r[0] for r in rows))
    if n == 0 or len(ids) % n:
        print("# per-stage table skipped: %d conv_gemm dispatches is not a multiple of the plan's %d" % (len(ids), n))
        return
    pos = {d: i % n for i, d in enumerate(ids)}
    have_u = len(unprof) == n
    agg = collections.OrderedDict()
    for d, cn, v, dur in rows:
        a = agg.setdefault(stage_of(labels[pos[d]]), {"GRBM_GUI_ACTIVE": 0.0, "SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "n": 0, "ns": 0.0, "uns": 0.0})
        a[cn] += v
        if cn == "GRBM_GUI_ACTIVE":
            a["n"] += 1
            a["ns"] += dur
            if have_u:
                a["uns"] += unprof[pos[d]] * 1e3
    print("# by network stage (%d forwards of %d conv_gemm launches):" % (len(ids) // n, n))
    if have_u:
        print("# MfmaUtil(u) = the same busy cycles / (the launch's UNPROFILED duration x 2.4 GHz x 1024 SIMDs): hipEvents around each launch of")
        print("#            a plain run (bench.py --breakdown, same tiles) - a counter-collecting dispatch takes 15-25 % longer than it does in")
        print("#            production, the MFMA instructions it issues are the same")
    print("%-18s %9s %12s %9s %12s %12s" % ("stage", "launches", "busy share", "MfmaUtil", "MfmaUtil(t)", "MfmaUtil(u)"))
    tot = sum(a["GRBM_GUI_ACTIVE"] for a in agg.values())
    for st in ("conv1", "res2 (conv2_x)", "res3 (conv3_x)", "res4 (conv4_x)", "res5 (conv5_x)", "heads"):
        if st in agg:
            a = agg[st]
            print("%-18s %9d %11.1f%% %8.1f%% %11.1f%% %11s" % (st, a["n"] * n // len(ids), 100 * a["GRBM_GUI_ACTIVE"] / tot,
                                                               100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8.0 * 1024),
                                                               100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["ns"] * 2.4 * 1024),
                                                               "%.1f%%" % (100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["uns"] * 2.4 * 1024)) if have_u else "-"))


def main(path, desc, plan_path=None, bench_path=None):
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
    print("# MfmaUtil(t) = the same busy cycles / (dispatch duration x 2.4 GHz x 1024 SIMDs): GRBM_GUI_ACTIVE of a counter-collecting dispatch")
    print("#            also covers its set-up and drain (~20 % more cycles than the timestamps), which the timestamps do not.")
    print("%-36s %6s %5s %9s %9s %11s %9s %9s" % ("kernel<BM,BN,BK,WR,WC,WK,PF>", "WGs", "n", "us/launch", "MfmaUtil", "MfmaUtil(t)", "WAIT_ANY", "WAIT_INST"))
    tm = tg = 0.0
    for (k, g, wg), d in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", (0,))[0]):
        if "conv_gemm" not in k and "wino_f23" not in k and "wino_h23" not in k and "ws1x1" not in k and "stem7x7" not in k:
            continue
        n = d["GRBM_GUI_ACTIVE"][1]
        gui = d["GRBM_GUI_ACTIVE"][0] / 8.0
        mf = d["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        wc = d["SQ_WAVE_CYCLES"][0]
        tm += mf
        tg += gui
        dur_cycles = d["GRBM_GUI_ACTIVE"][2] * 2.4  # ns x 2.4 cycles/ns
        label = kernel_names.label(k)
        print("%-36s %6d %5d %9.1f %8.1f%% %10.1f%% %8.1f%% %8.1f%%" % (label, g // wg, n, d["GRBM_GUI_ACTIVE"][2] / n / 1e3,
                                                                     100 * mf / (gui * 1024), 100 * mf / (dur_cycles * 1024),
                                                                     100 * d["SQ_WAIT_ANY"][0] / wc, 100 * d["SQ_WAIT_INST_ANY"][0] / wc))
    print("# all convolution / deconvolution dispatches (conv1..conv5 + heads): MfmaUtil = %.1f%% of the 1024 matrix pipes over the kernels' own run time" % (100 * tm / (tg * 1024)))
    if plan_path:
        per_stage(c, plan_path)
        in_flight(c, plan_path, bench_path)


def in_flight(c, plan_path, bench_path):
    """Counter collection serialises dispatches, so the regime `value` is measured in (several forwards in flight) cannot
    be profiled per dispatch.  But a forward issues the same MFMA instructions however it is scheduled: its busy cycles,
    counted here, times the forwards per second of the unprofiled bench line, is the share of the wall time the 1024
    matrix pipes are busy in that regime (at the 2.4 GHz the profiled clock is NOT: an upper bound on the clock, so a
    lower bound on the share)."""
    n = sum(1 for ln in open(plan_path) if len(ln.split("\t")) == 4 and ln.split("\t")[0].isdigit()
            and ("conv_gemm" in ln.split("\t")[1] or "wino_f23<" in ln or "wino_h23<" in ln or "ws1x1<" in ln or "ws1x1f<" in ln or "ws7x7f<" in ln or "stem7x7<" in ln))
    busy, disp = c.execute("select sum(value), count(*) from counters_collection where counter_name = 'SQ_VALU_MFMA_BUSY_CYCLES' "
                           "and (kernel_name like '%conv_gemm%' or kernel_name like '%wino__23%' or kernel_name like '%ws1x1%' or kernel_name like '%stem7x7%')").fetchone()
    if not n or not disp or disp % n:
        return
    per_fwd = busy / (disp // n)
    print("# MFMA busy cycles per forward (all %d convolution launches, summed over the 1024 SIMDs): %.3e" % (n, per_fwd))
    if not bench_path:
        return
    import json

    b = json.loads(open(bench_path).read().strip().splitlines()[-1])
    batch = b["config"].get("per_gpu_batch", 1)
    rows = [("one forward at a time", b.get("one_forward_at_a_time", {}).get("value")),
            ("%s forwards in flight (`value`)" % b["config"].get("forwards_in_flight", "?"), b.get("value"))]
    print("# share of the wall time the matrix pipes are busy = busy cycles per forward x forwards/s / (2.4e9 x 1024):")
    for name, v in rows:
        if v:
            print("#   %-36s %7.1f images/s -> %.1f%%" % (name, v, 100 * per_fwd * (v / batch) / (2.4e9 * 1024)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[3] if len(sys.argv) > 3 else None,
         sys.argv[4] if len(sys.argv) > 4 else None)
