#!/usr/bin/env python3
"""Phase stamps of ONE conv4_x expansion launch inside the 544x736 batch-1 float32 forward (cold filters: DC_DEBUG_TIMING_INSITU=1), as the
gather-GEMM tile and as the streaming form.   python tools/stream1x1f_insitu.py --find   prints the launch index of --layer;
DC_DEBUG_TIMING=<idx> DC_DEBUG_TIMING_INSITU=1 DC_AUTOTUNE=0 DC_STREAM1X1=0|1 python tools/stream1x1f_insitu.py   runs two forwards."""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--find", action="store_true")
    ap.add_argument("--layer", default="res4b20_branch2c")
    a = ap.parse_args()
    import caffe
    from deepcut_tools import deepercut_prototxt, synth_weights, write_caffemodel

    caffe.set_mode_gpu()
    caffe.set_device(0)
    path = os.path.join(tempfile.mkdtemp(), "synth152.caffemodel")
    write_caffemodel(path, "ResNet-152", synth_weights(152, seed=0))
    net = caffe.Net(deepercut_prototxt(152, 544, 736, 1), path, caffe.TEST, from_text=True, hipgraph=0)
    net.blobs["data"].data[...] = np.random.RandomState(0).rand(1, 3, 544, 736).astype(np.float32)
    if a.find:
        for ln in net.plan_text().splitlines():
            f = ln.split("\t")
            if len(f) >= 4 and f[3].startswith(a.layer):
                print(f[0])
                return
        raise SystemExit("no launch of " + a.layer)
    net.forward()
    net.forward()


if __name__ == "__main__":
    main()
