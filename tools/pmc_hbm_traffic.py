#!/usr/bin/env python
"""HBM traffic per convolution launch (conv_gemm, wino_f23 / wino_h23, ws1x1 / ws1x1f, stem7x7) from two rocprofv3 --pmc passes (rocpd sqlite), as MI355X_MICROARCH.md's HBM
section prescribes: FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes, both in KB, FETCH_SIZE x2 on gfx950 for
wide (16 B/lane) coalesced reads, WRITE_SIZE taken as is.
   python tools/pmc_hbm_traffic.py fetch.db write.db "<description>" [minimum bytes per forward] > profiles/<name>.json
The minimum defaults to the float32 batch-1 forward at 544x736 (2.07 GB of activations + 0.263 GB of filters)."""
import json
import sqlite3
import sys


def per_launch(path, counter):
    c = sqlite3.connect(path)
    v, n = c.execute("select sum(value), count(*) from counters_collection where counter_name = ? and (kernel_name like '%conv_gemm%' or kernel_name like '%wino__23%' or kernel_name like '%ws1x1%' or kernel_name like '%stem7x7%')",
                     (counter,)).fetchone()
    return v / n, n


def family(path, counter, like):
    c = sqlite3.connect(path)
    v, n = c.execute("select sum(value), count(*) from counters_collection where counter_name = ? and kernel_name like ?",
                     (counter, like)).fetchone()
    return (v or 0.0) / max(n, 1), n


def main(fetch_db, write_db, desc, min_bytes_per_forward=2.07e9 + 0.263e9):
    f, nf = per_launch(fetch_db, "FETCH_SIZE")
    w, nw = per_launch(write_db, "WRITE_SIZE")
    fam = {}
    for name, like in (("wino_f23", "%wino_f23%"), ("wino_h23", "%wino_h23%"), ("ws1x1", "%ws1x1%"), ("stem7x7", "%stem7x7%"), ("conv_gemm", "%conv_gemm%")):
        ff, n = family(fetch_db, "FETCH_SIZE", like)
        ww, _ = family(write_db, "WRITE_SIZE", like)
        fam[name] = {"dispatches": n, "hbm_bytes_per_launch": (2.0 * ff + ww) * 1024.0}
    launches_per_forward = 158.0
    print(json.dumps({
        "by_kernel_family": fam,
        "source": desc,
        "conv_gemm_dispatches": nf,
        "fetch_kb_per_launch_raw": f,
        "write_kb_per_launch_raw": w,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced "
                      "(16 B/lane) streaming reads -> x2; WRITE_SIZE uncalibrated, taken as is; both counters are KB",
        "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
        "algorithmic_min_bytes_per_launch": min_bytes_per_forward / launches_per_forward,
    }, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "", float(sys.argv[4]) if len(sys.argv) > 4 else 2.07e9 + 0.263e9)
