"""Soak test of the executors of one model on concurrent host threads (diagnostics; the plain-net companion of stress_groups.py):
T threads, each with a clone of its own, cycle through the three entries (device-resident batch, host batch, uint8 images with the
pose decoded on the device) over a list of shapes — some met in the sequential first pass, some new to everybody (lowering, tile
timing, buffer growth, graph capture while the other threads run) — and every result must equal, bit for bit, what the FIRST
executor produced for the same input sequentially.  usage: stress_clones.py [f16|f32] [rounds] [threads]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import caffe  # noqa: E402
from deepcut_tools import deepercut_prototxt, synth_weights  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nthreads = int(sys.argv[3]) if len(sys.argv) > 3 else 3
caffe.set_mode_gpu()
caffe.set_device(0)
dev = torch.device("cuda", 0)
SHAPES = [(1, 64, 80), (2, 40, 56), (1, 104, 136), (3, 72, 104)]          # met by executor 0 before the threads start
LATE = [(2, 88 + 8 * k, 120 + 16 * k) for k in range(3)]                 # first met inside the threads (by all of them at once)
layers = synth_weights(152, seed=0)
net = caffe.Net(deepercut_prototxt(152, 64, 80, 1), caffe.TEST, from_text=True, hipgraph=1, dtype=dtype)
for name, _t, blobs in layers:
    for pb, b in zip(net.params[name], blobs):
        pb.data[...] = b
nets = [net] + [net.clone() for _ in range(nthreads - 1)]
rs = np.random.RandomState(3)
host_in = {s: (rs.randn(s[0], 3, s[1], s[2]) * 50).astype(np.float32) for s in SHAPES + LATE}
dev_in = {s: torch.from_numpy(a).to(dev) for s, a in host_in.items()}
imgs = {s: rs.randint(0, 256, size=(s[0], s[1], s[2], 3), dtype=np.uint8) for s in SHAPES + LATE}
KEYS = ("prob", "loc_pred", "next_pred")


def dev_outs(s):
    n, h, w = s
    return [torch.zeros(n, c, h // 8, w // 8, device=dev) for c in (14, 28, 364)]


def three_ways(ex, s, stream, outs):
    """-> (device-entry maps, host-entry maps, image-entry pose + prob) of executor ex at shape s"""
    n, h, w = s
    ex.forward_device(dev_in[s].data_ptr(), n, h, w, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    a = [t.cpu().numpy().copy() for t in outs]
    hb = ex.forward_batch(host_in[s])
    b = [hb[k].copy() for k in KEYS]
    im = ex.forward_images(imgs[s], 1.0, want=("prob",), pose=True)
    c = [np.array(im["pose"]).copy(), im["prob"].copy()]
    return a, b, c


streams = [torch.cuda.Stream(dev) for _ in nets]
ref = {}
with torch.cuda.stream(streams[0]):
    for s in SHAPES + LATE:
        ref[s] = three_ways(nets[0], s, streams[0], dev_outs(s))
# the LATE shapes are new to every OTHER executor (their plans, buffers and graphs; the tile choices are the model's)
bad = [0] * nthreads
done = [0] * nthreads
errors = []


def same(x, y):
    return all(np.array_equal(p, q) for p, q in zip(x, y))


def worker(t):
    try:
        caffe.set_mode_gpu()
        caffe.set_device(0)
        ex = nets[t]
        order = SHAPES + LATE
        with torch.cuda.stream(streams[t]):
            for r in range(rounds):
                s = order[(r * (t + 1) + t) % len(order)] if r > 2 else LATE[(r + t) % len(LATE)]
                got = three_ways(ex, s, streams[t], dev_outs(s))
                for x, y in zip(got, ref[s]):
                    if not same(x, y):
                        bad[t] += 1
                done[t] += 1
    except Exception as e:  # noqa: BLE001
        errors.append("thread %d round %d: %r" % (t, done[t], e))


t0 = time.time()
threads = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
for th in threads:
    th.start()
for th in threads:
    th.join()
dt = time.time() - t0
for e in errors:
    print(e)
ok = not errors and sum(bad) == 0 and all(d == rounds for d in done)
print("dtype %s: %d threads x %d rounds x 3 entries in %.1f s" % (dtype, nthreads, rounds, dt))
print("OK" if ok else "BAD", "mismatches", bad, "rounds done", done)
sys.exit(0 if ok else 1)
