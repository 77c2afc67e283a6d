"""How much does WHICH streams the forwards in flight run on matter?  (diagnostics; DESIGN 11 open end)  S executors of the
float32 batch-1 544x736 forward, each on one of P candidate streams: every S-subset of the candidates is timed (a burst of
forwards round-robin over the executors), best / median / worst subset printed — what `Pipeline` or `bench.py` would gain by
choosing their streams by measurement, as the lanes of a group do (NetGroup::choose_lane_streams).
usage: stream_subset_probe.py [S=4] [P=8] [forwards per timing=24]"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import caffe  # noqa: E402
from deepcut_tools import deepercut_prototxt, synth_weights  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 24
H, W = 544, 736
caffe.set_mode_gpu()
caffe.set_device(0)
dev = torch.device("cuda", 0)
net = caffe.Net(deepercut_prototxt(152, H, W), caffe.TEST, from_text=True, hipgraph=1)
for name, _t, blobs in synth_weights(152, seed=0):
    for pb, b in zip(net.params[name], blobs):
        pb.data[...] = b
nets = [net] + [net.clone() for _ in range(S - 1)]
streams = [torch.cuda.Stream(dev) for _ in range(P)]
x = torch.from_numpy((np.random.RandomState(0).randn(1, 3, H, W) * 50).astype(np.float32)).to(dev)
outs = [[torch.empty(1, c, H // 8, W // 8, device=dev) for c in (14, 28, 364)] for _ in nets]
for ex, o in zip(nets, outs):  # lowered, tuned, captured
    ex.forward_device(x.data_ptr(), 1, H, W, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=streams[0].cuda_stream)
    torch.cuda.synchronize(dev)


def burst(subset):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(NF):
        e = i % S
        o = outs[e]
        nets[e].forward_device(x.data_ptr(), 1, H, W, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), stream=streams[subset[e]].cuda_stream)
    torch.cuda.synchronize(dev)
    return NF / (time.perf_counter() - t0)


rows = []
for subset in itertools.combinations(range(P), S):
    burst(subset)
    rows.append((max(burst(subset), burst(subset)), subset))
rows.sort(reverse=True)
vals = [r[0] for r in rows]
print("%d executors on %d-subsets of %d candidate streams (%d subsets, %d forwards per timing, best of 2):" % (S, S, P, len(rows), NF))
print("  best   %.1f images/s on streams %s" % rows[0])
print("  median %.1f" % vals[len(vals) // 2])
print("  worst  %.1f on streams %s" % rows[-1])
print("  first S created (what a caller gets without choosing): %.1f" % [v for v, s in rows if s == tuple(range(S))][0])
hist = np.histogram(vals, bins=8)
for c, lo, hi in zip(hist[0], hist[1][:-1], hist[1][1:]):
    print("  %6.1f - %6.1f images/s: %d subsets" % (lo, hi, c))
# the best subsets, re-timed with a long burst (the short bursts rank, this one measures)
NF = 6 * NF
for v, subset in rows[:3] + [(0, tuple(range(S)))]:
    burst(subset)
    print("  long burst on %s: %.1f images/s" % (subset, burst(subset)))
