"""Stress test of executors in flight: Pipeline(depth 3) vs sequential forwards, many rounds (diagnostics)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import caffe  # noqa: E402
from deepcut_tools import Pipeline, deepercut_prototxt, synth_weights  # noqa: E402

caffe.set_mode_gpu()
caffe.set_device(0)
h, w = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "104,136").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
layers = synth_weights(152, seed=0)
bad = 0
for trial in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):  # a fresh net (fresh tuning) per trial
    net = caffe.Net(deepercut_prototxt(152, h, w), caffe.TEST, from_text=True, hipgraph=1)
    for name, _t, blobs in layers:
        for pb, b in zip(net.params[name], blobs):
            pb.data[...] = b
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(trial)
    imgs = [torch.from_numpy((rs.randn(1, 3, h, w) * 50).astype(np.float32)).to(dev) for _ in range(7)]
    ref = [net.forward_batch(im.cpu().numpy()) for im in imgs]
    pipe = Pipeline(net, depth=3)
    outs = [[torch.empty(1, c, h // 8, w // 8, device=dev) for c in (14, 28, 364)] for _ in imgs]
    torch.cuda.synchronize()
    for r in range(rounds):
        for o in outs:
            for t in o:
                t.zero_()
        torch.cuda.synchronize()
        for i, im in enumerate(imgs):
            pipe.submit(im.data_ptr(), 1, h, w, outs[i][0].data_ptr(), outs[i][1].data_ptr(), outs[i][2].data_ptr(), tag=i)
        pipe.drain()
        for i in range(len(imgs)):
            for k, t in zip(("prob", "loc_pred", "next_pred"), outs[i]):
                e = float(np.abs(t.cpu().numpy() - ref[i][k]).max())
                if e > 1e-5:
                    bad += 1
                    print("trial %d round %d request %d %s: max err %g" % (trial, r, i, k, e))
    kinds = sorted(set(ln.split("\t")[1] for ln in net.plan_text().splitlines() if "\t" in ln))
    print("trial %d done, %d bad so far; kernels: %s" % (trial, bad, " ".join(kinds)[:400]))
print("BAD" if bad else "OK", bad)
