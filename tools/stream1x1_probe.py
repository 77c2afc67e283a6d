#!/usr/bin/env python3
"""The streaming form of the float16 dense 1x1 layers (csrc/stream1x1.hip, "ws1x1") against the gather-GEMM tiles, on the 1x1 expansion
layers (+ BatchNorm/Scale + shortcut + ReLU) of the 544x736 batch-8 forward (BASELINE configs[2]'s 1.0-scale member):
  1. the forced streaming launch against the forced direct launch (bit for bit: the epilogues are the same instruction sequence) and
     against a float64 NumPy evaluation of the same float16 operands;
  2. the autotuner's isolated timings (five launches back to back, best of two bursts) of every candidate, with the bytes the layer has to move.

    python tools/stream1x1_probe.py [--batch 8] [--shapes res4c,res3c,res2c,res5c,res2a1]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import numpy as np

# cin, cout, h, w, shortcut
SHAPES = {"res4c": (256, 1024, 34, 46, True), "res3c": (128, 512, 68, 92, True), "res2c": (64, 256, 136, 184, True),
          "res5c": (512, 2048, 34, 46, True), "res2a1": (64, 256, 136, 184, False)}


def net_text(n, cin, cout, h, w, shortcut):
    L = ['name: "s"', 'input: "data"'] + ["input_dim: %d" % d for d in (n, cin, h, w)]
    if shortcut:
        L += ['input: "sc"'] + ["input_dim: %d" % d for d in (n, cout, h, w)]
    L.append('layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: %d kernel_size: 1 bias_term: false } }' % cout)
    L.append('layer { name: "bn" type: "BatchNorm" bottom: "c" top: "c" batch_norm_param { use_global_stats: true } }')
    L.append('layer { name: "scale" type: "Scale" bottom: "c" top: "c" scale_param { bias_term: true } }')
    if shortcut:
        L.append('layer { name: "sum" type: "Eltwise" bottom: "sc" bottom: "c" top: "sum" }')
        L.append('layer { name: "relu" type: "ReLU" bottom: "sum" top: "sum" }')
    return "\n".join(L) + "\n"


def make(caffe, a, name, rs, env, **kw):
    cin, cout, h, w, shortcut = SHAPES[name]
    for k in ("DC_STREAM1X1", "DC_AUTOTUNE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    net = caffe.Net(net_text(a.batch, cin, cout, h, w, shortcut), caffe.TEST, from_text=True, dtype=a.dtype, **kw)
    return net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--shapes", default="res4c,res3c,res2c,res5c,res2a1")
    ap.add_argument("--dtype", default="f16", choices=("f16", "f32"), help="f32: the float32 form (csrc/stream1x1_f32.hip, 'ws1x1f': K = 64 / 128 / 256 / 512, not bit-identical to the tiles)")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--stamps", action="store_true", help="only the forced streaming launch of every shape, graph off (run with DC_DEBUG_TIMING=0)")
    a = ap.parse_args()
    import caffe

    caffe.set_mode_gpu()
    caffe.set_device(0)
    for name in a.shapes.split(","):
        cin, cout, h, w, shortcut = SHAPES[name]
        rs = np.random.RandomState(len(name) + cin)
        x = rs.randn(a.batch, cin, h, w).astype(np.float32)
        sc = rs.randn(a.batch, cout, h, w).astype(np.float32)
        wt = (rs.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float32)
        mean, var = rs.randn(cout).astype(np.float32) * 0.1, rs.uniform(0.5, 1.5, cout).astype(np.float32)
        ga, be = rs.uniform(0.5, 1.5, cout).astype(np.float32), rs.randn(cout).astype(np.float32) * 0.1
        out = "sum" if shortcut else "c"

        def fill(net):
            net.params["c"][0].data[...] = wt
            net.params["bn"][0].data[...] = mean
            net.params["bn"][1].data[...] = var
            net.params["bn"][2].data[...] = 1.0
            net.params["scale"][0].data[...] = ga
            net.params["scale"][1].data[...] = be
            net.blobs["data"].data[...] = x
            if shortcut:
                net.blobs["sc"].data[...] = sc

        if a.stamps:
            net = make(caffe, a, name, rs, {"DC_STREAM1X1": "1", "DC_AUTOTUNE": "0"}, hipgraph=0)
            fill(net)
            print(name, flush=True)
            net.forward()
            continue
        if not a.no_check:
            res = {}
            for mode in ("1", "0"):
                net = make(caffe, a, name, rs, {"DC_STREAM1X1": mode, "DC_AUTOTUNE": "0"})
                fill(net)
                net.forward()
                plan = net.plan_text()
                assert ("ws1x1" in plan) == (mode == "1"), plan
                res[mode] = net.blobs[out].data.copy()
                del net
            # float64 evaluation of the float16 operands (the library scales each filter row by a power of two before rounding: exact)
            ht = np.float16 if a.dtype == "f16" else np.float32
            xh = x.astype(ht).astype(np.float64)
            wh = wt.reshape(cout, cin).astype(ht).astype(np.float64)
            ref = np.einsum("nchw,oc->nohw", xh, wh)
            a_ = (ga.astype(np.float64) / np.sqrt(var.astype(np.float64) + 1e-5))
            ref = ref * a_[None, :, None, None] + (be.astype(np.float64) - mean.astype(np.float64) * a_)[None, :, None, None]
            if shortcut:
                ref = np.maximum(ref + sc.astype(ht).astype(np.float64), 0.0)
            err = float(np.abs(res["1"] - ref).max())
            same = bool(np.array_equal(res["1"], res["0"]))
            print("%s batch %d: ws1x1 vs float64 of the float16 operands max|d| = %.3e (range %.1f) | vs the direct tile: %s (max|d| %.3e)" % (
                name, a.batch, err, float(np.abs(ref).max()), "bit-identical" if same else "DIFFERENT", float(np.abs(res["1"] - res["0"]).max())), flush=True)
        net = make(caffe, a, name, rs, {})
        fill(net)
        net.forward()
        M = a.batch * h * w
        mbytes = (2.0 if a.dtype == "f16" else 4.0) * (M * cin + M * cout * (2 if shortcut else 1) + cin * cout) / 1e6
        for e in net.tune_report():
            timed = sorted(e["timed"], key=lambda t: t[1])
            direct = [t for t in timed if not t[0].startswith("ws1x1")]
            ws = [t for t in timed if t[0].startswith("ws1x1")]
            gflop = 2.0 * M * cin * cout / 1e9
            print("%s %dx%dx%d %d->%d%s (%.1f MB, %.3f GFLOP): chosen %s | best direct %s %.2f us (%.2f TB/s, %.1f TFLOP/s) | %s" % (
                name, a.batch, h, w, cin, cout, " +shortcut" if shortcut else "", mbytes, gflop, e["tile"], direct[0][0], direct[0][1], mbytes / direct[0][1],
                gflop / direct[0][1] * 1e3, " ".join("%s %.2f us (%.2f TB/s, %.1f TFLOP/s)" % (t[0], t[1], mbytes / t[1], gflop / t[1] * 1e3) for t in ws)), flush=True)
        del net


if __name__ == "__main__":
    main()
