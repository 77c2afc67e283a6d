#!/usr/bin/env python3
"""Where the float32 streaming 1x1 form (csrc/stream1x1_f32.hip, "ws1x1f") stands inside the 544x736 batch-1 forward (BASELINE configs[1]):
the autotuner's candidates for every signature that has the form among them, and one forward at a time with the form on / off / chosen
per shape (hipGraph replay, 200 forwards).      python tools/stream1x1f_in_net.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import numpy as np


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--pin", default="", help="only this: tune, put the conv4_x expansion on this tile ('ws1x1f' or 'tuned'), run --forwards forwards (for rocprofv3 --kernel-trace --stats)")
    ap.add_argument("--forwards", type=int, default=100)
    a = ap.parse_args()
    import caffe
    import tempfile

    from deepcut_tools import deepercut_prototxt, synth_weights, write_caffemodel

    caffe.set_mode_gpu()
    caffe.set_device(0)
    path = os.path.join(tempfile.mkdtemp(), "synth152.caffemodel")
    write_caffemodel(path, "ResNet-152", synth_weights(152, seed=0))
    proto = deepercut_prototxt(152, 544, 736, 1)
    img = np.random.RandomState(0).rand(1, 3, 544, 736).astype(np.float32)
    def clock(net, label):
        for _ in range(5):
            net.forward()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            n = 100
            for _ in range(n):
                net.forward()
            best = min(best, (time.perf_counter() - t0) / n)
        took = sum("ws1x1f" in ln for ln in net.plan_text().splitlines())
        print("%s: %d launches on ws1x1f, %.3f ms per forward (%.1f images/s)" % (label, took, best * 1e3, 1.0 / best), flush=True)

    os.environ.pop("DC_STREAM1X1", None)
    net = caffe.Net(proto, path, caffe.TEST, from_text=True, hipgraph=1)
    net.blobs["data"].data[...] = img
    if a.pin:
        net.forward()
        if a.pin != "tuned":
            for e in net.tune_report():
                if e["signature"].startswith("1564/1024/256/"):
                    net.set_tile(e["signature"], a.pin)
        for _ in range(a.forwards):
            net.forward()
        print("%d forwards, %d launches on ws1x1f" % (a.forwards, sum("ws1x1f" in ln for ln in net.plan_text().splitlines())))
        return
    clock(net, "as tuned")
    sigs = []
    for e in net.tune_report():
        if any(t[0] == "ws1x1f" for t in e["timed"]):
            timed = sorted(e["timed"], key=lambda t: t[1])
            sigs.append((e["signature"], e["tile"]))
            print("   %s x%d: chosen %s | %s" % (e["signature"], e["launches"], e["tile"], " ".join("%s %.2f" % t for t in timed[:4])), flush=True)
    for sig, tile in sigs:
        net.set_tile(sig, "ws1x1f")
        clock(net, "ws1x1f on %s" % sig)
        net.set_tile(sig, tile)
        clock(net, "back on %s" % tile)


if __name__ == "__main__":
    main()
