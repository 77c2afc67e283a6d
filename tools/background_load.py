"""ON THE GPU BOX: N batch-1 forwards (544x736 float32 ResNet-152) kept in flight on N executors for a number of seconds — the
neighbours of a profiled process (tools/pmc_in_flight.sh), the load under a clock probe (tools/clock_under_load.sh).
usage: background_load.py [executors] [seconds] [f32|f16] [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import caffe  # noqa: E402
from deepcut_tools import deepercut_prototxt, synth_weights  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
caffe.set_mode_gpu()
caffe.set_device(0)
dev = torch.device("cuda", 0)
H, W = 544, 736
net = caffe.Net(deepercut_prototxt(152, H, W, B), caffe.TEST, from_text=True, hipgraph=1, dtype=dtype)
for name, _t, blobs in synth_weights(152, seed=0):
    for p, b in zip(net.params[name], blobs):
        p.data[...] = b
nets = [net] + [net.clone() for _ in range(S - 1)]
for e in nets:
    e.reserve(B, H, W)
if S > 1:
    caffe.choose_streams(nets)
x = (torch.randn(B, 3, H, W) * 50).to(dev)
outs = [[torch.empty(B, c, H // 8, W // 8, device=dev) for c in (14, 28, 364)] for _ in nets]
print("background load: %d executors ready" % S, flush=True)
open(os.environ.get("DC_LOAD_READY", "/tmp/dc_load_ready"), "w").write("1")
t_end, n = time.time() + secs, 0
t_mark, n_mark = time.time(), 0
while time.time() < t_end:
    for k, e in enumerate(nets):
        e.forward_device(x.data_ptr(), B, H, W, outs[k][0].data_ptr(), outs[k][1].data_ptr(), outs[k][2].data_ptr(), stream="own")
    for e in nets:
        e.synchronize()
    n += S * B
    now = time.time()
    if now - t_mark >= 3.0:  # a line every 3 s (the caller may kill this process: what it did is on record)
        print("background load: t=%.1f  %.1f images/s over the last %.1f s" % (now, (n - n_mark) / (now - t_mark), now - t_mark), flush=True)
        t_mark, n_mark = now, n
print("background load: %d forwards in %.0f s = %.1f images/s" % (n, secs, n / secs), flush=True)
