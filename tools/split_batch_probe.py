#!/usr/bin/env python
"""ONE batch-8 request at 544x736 as two concurrent half-batches (a 2-member caffe.NetGroup in two lanes: nothing merged, two
streams) against the plain batch-8 forward — does intra-request concurrency pay for a single-scale batch?  (diagnostics)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import caffe  # noqa: E402
from deepcut_tools import deepercut_prototxt, synth_weights  # noqa: E402

caffe.set_mode_gpu()
caffe.set_device(0)
layers = synth_weights(152, seed=0)
H, W = 544, 736
for dtype in sys.argv[1:] or ["f16", "f32"]:
    net = caffe.Net(deepercut_prototxt(152, H, W, 8), caffe.TEST, from_text=True, hipgraph=1, dtype=dtype)
    for name, _t, blobs in layers:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    dev = torch.device("cuda", 0)
    x = (torch.randn(8, 3, H, W) * 50).to(dev)
    halves = [x[:4].contiguous(), x[4:].contiguous()]
    quarters = [x[i:i + 2].contiguous() for i in range(0, 8, 2)]
    st = torch.cuda.Stream(dev)
    forms = {"plain batch 8": lambda: net.forward_device(x.data_ptr(), 8, H, W, stream=st.cuda_stream)}
    g2 = caffe.NetGroup.for_shapes(net.clone(), [(4, H, W)] * 2, lanes=2)
    forms["2 lanes x batch 4"] = lambda: g2.forward_device([h.data_ptr() for h in halves], [(4, H, W)] * 2, stream=st.cuda_stream)
    g4 = caffe.NetGroup.for_shapes(net.clone(), [(2, H, W)] * 4, lanes=2)
    forms["2 lanes x (2 merged batch-2 members)"] = lambda: g4.forward_device([q.data_ptr() for q in quarters], [(2, H, W)] * 4, stream=st.cuda_stream)
    for name, fn in forms.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
                torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 10 * 1e3)
        print("%s %-40s %.3f ms per 8 images (min %.3f) = %.0f images/s" % (dtype, name, sorted(ts)[2], min(ts), 8 / sorted(ts)[2] * 1e3), flush=True)
