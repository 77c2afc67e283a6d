"""ON THE GPU BOX: what dc_forward_batch (the in-process multi-executor forward, csrc/multi_gpu.cpp) delivers on ONE GPU — BASELINE
configs[3]'s 64 images of 544x736 dealt to 1 / 2 / 4 / 8 executors that share GPU 0 over the loop-back transport, host arrays in, host
maps out (all three).  On one GPU this measures the machinery (threads, staging, batches of 64 / E images, gather copies, host copies),
not scaling: the executors share one device.  Twice: the caller's arrays pageable (staged / scattered through pinned buffers by the executors'
threads) and pinned (caffe.pinned_empty: moved in place by the DMA engines).  usage: comm_bench.py [f32|f16] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import caffe  # noqa: E402
from deepcut_tools import deepercut_prototxt, synth_weights  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
caffe.set_mode_gpu()
caffe.set_device(0)
net = caffe.Net(deepercut_prototxt(152, 544, 736, 8), caffe.TEST, from_text=True, hipgraph=1, dtype=dtype)
for name, _t, blobs in synth_weights(152, seed=0):
    for p, b in zip(net.params[name], blobs):
        p.data[...] = b
imgs = [(np.random.RandomState(100 + i).randn(3, 544, 736) * 50).astype(np.float32) for i in range(64)]
pimgs = []
for x in imgs:
    a = caffe.pinned_empty(x.shape)
    a[...] = x
    pimgs.append(a)
for what, src, pin in (("pageable", imgs, False), ("PINNED", pimgs, True)):
    print("# dc_forward_batch, 64 images 3x544x736 (%s), executors sharing GPU 0 (DC_COMM_PEER), host in / host out, all three maps, caller's arrays %s" % (dtype, what))
    for E in (1, 2, 4, 8):
        nets = [net] + [net.clone() for _ in range(E - 1)]
        comm = caffe.Comm(nets, devices=[0] * E, transport="peer")
        out = comm.forward(src, pinned=pin)  # lowers / tunes / allocates the batch shape on every executor; the result arrays are kept
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            comm.forward(src, pinned=pin, out=out)
            ts.append(time.perf_counter() - t0)
        del out
        t = sorted(ts)[len(ts) // 2]
        print("%d executor(s), %2d images each: %.1f ms per 64 images = %.1f images/s" % (E, 64 // E, t * 1e3, 64 / t))
        del comm, nets
