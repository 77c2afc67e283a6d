#!/usr/bin/env python
"""Turn a rocprofv3 rocpd sqlite result (--kernel-trace [--stats]) into the text summary kept under
profiles/:  python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.txt
Durations are computed from the per-dispatch start/end timestamps (ns)."""
import collections
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_names  # noqa: E402


def main(path):
    c = sqlite3.connect(path)
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    assert "start" in cols and "end" in cols and "name" in cols, cols
    agg = collections.OrderedDict()
    for name, s, e in c.execute("select name, start, end from kernels"):
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    print("# rocprofv3 --kernel-trace --stats summary of %s (from per-dispatch start/end, ns)" % path)
    print("%-112s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        shown = kernel_names.label(name) if ("conv_gemm_kernel" in name or "wino_f23" in name or "wino_h23" in name or "ws1x1_kernel" in name or "ws1x1f_kernel" in name or "stem7x7_kernel" in name) else name
        print("%-112s %8d %12.1f %10.2f %7.2f" % (shown[:112], n, t / 1e3, t / n / 1e3, 100.0 * t / tot))
    print("# total kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(a[0] for a in agg.values())))
    n = sum(a[0] for k, a in agg.items() if "conv_gemm_kernel" in k or "wino_f23_kernel" in k or "wino_h23_kernel" in k or "ws1x1_kernel" in k or "ws1x1f_kernel" in k or "stem7x7_kernel" in k)
    t = sum(a[1] for k, a in agg.items() if "conv_gemm_kernel" in k or "wino_f23_kernel" in k or "wino_h23_kernel" in k or "ws1x1_kernel" in k or "ws1x1f_kernel" in k or "stem7x7_kernel" in k)
    if n:
        print("# conv_gemm_kernel + wino_f23_kernel (every convolution / deconvolution launch): %d dispatches, %.3f ms total, average %.2f us per launch" % (n, t / 1e6, t / n / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
