#!/usr/bin/env python3
"""conv1 of a float16 net (7x7 / stride 2 / pad 3, 3 -> 64, + BatchNorm / Scale / ReLU) at 544x736: the autotuner's isolated timings of the
stem kernel (csrc/stem_f16.hip, "stem7x7") and of the row-tap gather-GEMM tiles, batch 1 and 8.

    python tools/stem_probe.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python"), os.path.join(ROOT,"tests")):
    sys.path.insert(0, p)
import numpy as np
import caffe
import test_gpu_stem as T
caffe.set_mode_gpu(); caffe.set_device(0)
for n in (1, 8):
    net = caffe.Net(T._net_text(n, 544, 736, True, True), caffe.TEST, from_text=True, dtype="f16")
    net.blobs["data"].data[...] = np.random.RandomState(0).randn(n,3,544,736).astype(np.float32)
    net.forward()
    for e in net.tune_report():
        timed = sorted(e["timed"], key=lambda t: t[1])
        print("batch", n, "chosen", e["tile"], " | ".join("%s %.2f us" % (t[0], t[1]) for t in timed[:4]), flush=True)
