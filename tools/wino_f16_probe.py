#!/usr/bin/env python3
"""The float16 Winograd form (csrc/wino_f16.hip) against the direct tiles, layer by layer, on the 3x3 shapes of the 544x736 batch-8
forward (BASELINE configs[2]'s 1.0-scale member): the autotuner's isolated timings (five launches back to back, best of two
bursts) of every candidate, then — with DC_DEBUG_TIMING=0 in the environment — the phase stamps of the forced Winograd launch.

    python tools/wino_f16_probe.py [--batch 8] [--shapes res4,res3,res2,res5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import numpy as np

SHAPES = {"res4": (256, 256, 34, 46, 1), "res3": (128, 128, 68, 92, 1), "res2": (64, 64, 136, 184, 1), "res5": (512, 512, 34, 46, 2)}


def net_text(n, cin, cout, h, w, dil):
    L = ['name: "w"', 'input: "data"'] + ["input_dim: %d" % d for d in (n, cin, h, w)]
    L.append('layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: %d kernel_size: 3 '
             'pad: %d dilation: %d bias_term: false } }' % (cout, dil, dil))
    L.append('layer { name: "relu" type: "ReLU" bottom: "c" top: "c" }')
    return "\n".join(L) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--shapes", default="res4,res3,res2,res5")
    ap.add_argument("--stamps", action="store_true", help="only the forced Winograd launch of every shape (run with DC_DEBUG_TIMING=0)")
    a = ap.parse_args()
    import caffe

    caffe.set_mode_gpu()
    caffe.set_device(0)
    rs = np.random.RandomState(0)
    for name in a.shapes.split(","):
        cin, cout, h, w, dil = SHAPES[name]
        x = rs.randn(a.batch, cin, h, w).astype(np.float32)
        wt = (rs.randn(cout, cin, 3, 3) / np.sqrt(9.0 * cin)).astype(np.float32)
        os.environ.pop("DC_WINOGRAD", None)
        os.environ.pop("DC_AUTOTUNE", None)
        if a.stamps:
            os.environ["DC_WINOGRAD"] = "1"
            os.environ["DC_AUTOTUNE"] = "0"
            f = caffe.Net(net_text(a.batch, cin, cout, h, w, dil), caffe.TEST, from_text=True, dtype="f16", hipgraph=0)
            f.params["c"][0].data[...] = wt
            f.blobs["data"].data[...] = x
            print(name, flush=True)
            f.forward()
            continue
        net = caffe.Net(net_text(a.batch, cin, cout, h, w, dil), caffe.TEST, from_text=True, dtype="f16")
        net.params["c"][0].data[...] = wt
        net.blobs["data"].data[...] = x
        net.forward()
        flops = 2.0 * a.batch * h * w * cout * cin * 9
        for e in net.tune_report():
            timed = sorted(e["timed"], key=lambda t: t[1])
            wino = [t for t in timed if t[0].startswith("wino")]
            print("%s %dx%dx%d %d->%d d%d: chosen %s | best direct %s %.2f us (%.0f TF/s) | %s" % (
                name, a.batch, h, w, cin, cout, dil, e["tile"], [t for t in timed if not t[0].startswith("wino")][0][0],
                [t for t in timed if not t[0].startswith("wino")][0][1], flops / [t for t in timed if not t[0].startswith("wino")][0][1] / 1e6,
                " ".join("%s %.2f us (%.0f TF/s direct-form)" % (t[0], t[1], flops / t[1] / 1e6) for t in wino)), flush=True)


if __name__ == "__main__":
    main()
