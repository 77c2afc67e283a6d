"""Soak test of grouped forwards in flight (diagnostics): G groups of one model (a net + clones each; 4 members = two concurrent
lanes), each driven by its own host thread on its own stream, alternating between two tuples of member shapes (plan cache, graphs,
lane-stream choice) for many rounds; every result must equal, bit for bit, what the same group produced for the same inputs in
its first, sequential round.  usage: stress_groups.py [f16|f32] [rounds] [groups] [new shapes per thread]"""
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

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ngroups = int(sys.argv[3]) if len(sys.argv) > 3 else 3
caffe.set_mode_gpu()
caffe.set_device(0)
dev = torch.device("cuda", 0)
SETS = [[(2, 40, 56), (2, 64, 64), (2, 72, 104), (2, 104, 136)], [(2, 104, 136), (2, 40, 56), (2, 64, 64), (2, 72, 104)]]
layers = synth_weights(152, seed=0)
n0, h0, w0 = SETS[0][0]
net = caffe.Net(deepercut_prototxt(152, h0, w0, n0), caffe.TEST, from_text=True, hipgraph=1, dtype=dtype)
for name, _t, blobs in layers:
    for pb, b in zip(net.params[name], blobs):
        pb.data[...] = b
groups = [caffe.NetGroup.for_shapes(net if g == 0 else net.clone(), SETS[0]) for g in range(ngroups)]
rs = np.random.RandomState(7)
inputs = [[torch.from_numpy((rs.randn(n, 3, h, w) * 50).astype(np.float32)).to(dev) for (n, h, w) in s] for s in SETS]


def outs_for(shapes):
    return [[torch.zeros(n, c, h // 8, w // 8, device=dev) for c in (14, 28, 364)] for (n, h, w) in shapes]


def run(grp, si, outs, stream):
    grp.forward_device([x.data_ptr() for x in inputs[si]], SETS[si], [o[0].data_ptr() for o in outs], [o[1].data_ptr() for o in outs],
                       [o[2].data_ptr() for o in outs], stream=stream.cuda_stream)


# round 0: sequentially (tuning, graphs, lane streams), the results every later round must reproduce
streams = [torch.cuda.Stream(dev) for _ in groups]
ref, tiles = [], []
for g, grp in enumerate(groups):
    per_set, per_set_tiles = [], []
    for si in range(len(SETS)):
        o = outs_for(SETS[si])
        run(grp, si, o, streams[g])
        streams[g].synchronize()
        per_set.append([[t.clone() for t in m] for m in o])
        per_set_tiles.append([(r["signature"], r["tile"]) for r in grp.tune_report()] +
                             [(("member %d " % c) + r["signature"], r["tile"]) for c in range(4) for r in grp.nets[c].tune_report()])
    ref.append(per_set)
    tiles.append(per_set_tiles)
for g in range(1, ngroups):  # the groups share the model's tile choices: same launches, same bits
    for si in range(len(SETS)):
        for c, (a, b) in enumerate(zip(ref[0][si], ref[g][si])):
            for ta, tb in zip(a, b):
                if not torch.equal(ta, tb):
                    print("sequential round: group %d differs from group 0 on shape set %d, member %d %s: max |diff| %.3e of max %.3e"
                          % (g, si, c, tuple(ta.shape), float((ta - tb).abs().max()), float(ta.abs().max())))
        for (ka, va), (kb, vb) in zip(tiles[0][si], tiles[g][si]):
            if (ka, va) != (kb, vb):
                print("  set %d tiles differ: %s %s | %s %s" % (si, ka[:70], va, kb[:70] if kb != ka else "=", vb))
ngrow = int(sys.argv[4]) if len(sys.argv) > 4 else 4
big = torch.from_numpy((rs.randn(2, 3, 256, 256) * 50).astype(np.float32)).to(dev)  # read as [2,3,h,w] for any h*w <= 256*256
bad = [0] * ngroups
done = [0] * ngroups


def worker(g):
    caffe.set_mode_gpu()  # mode and device are per thread (common.cpp:13-20 of the reference)
    caffe.set_device(0)
    grp = groups[g]
    o = [outs_for(s) for s in SETS]
    probe = torch.arange(64, device=dev, dtype=torch.float32)
    grow = [(2, 112 + 16 * k + 8 * g, 144 + 16 * k) for k in range(ngrow)]
    torch.cuda.synchronize(dev)
    for r in range(rounds):
        si = (r + g) % len(SETS) if r % 5 else r % len(SETS)
        if r % 3 == 0:  # what a host application does beside us: synchronous work on the legacy stream (while another thread may be capturing)
            assert float(probe.sum().cpu()) == 2016.0
        with torch.cuda.stream(streams[g]):
            for m in o[si]:
                for t in m:
                    t.zero_()
            run(grp, si, o[si], streams[g])
            if r % 7 == 3:  # a member alone between grouped forwards (its own plan at another shape, then back) ...
                grp.nets[r % 4].forward_device(inputs[si][0].data_ptr(), *SETS[si][0], stream=streams[g].cuda_stream)
            if r % 11 == 2 + g and grow:  # ... and at a shape nobody has met, larger every time: lowering, tile timing, buffer growth
                n, h, w = grow.pop(0)     #     (allocation, fill, release) and a graph capture, while the other threads run and capture
                grp.nets[(r + g) % 4].forward_device(big.data_ptr(), n, h, w, stream=streams[g].cuda_stream)
            streams[g].synchronize()
            for m, mr in zip(o[si], ref[g][si]):
                for t, tr in zip(m, mr):
                    if not torch.equal(t, tr):
                        bad[g] += 1
        done[g] += 1


t0 = time.time()
threads = [threading.Thread(target=worker, args=(g,)) for g in range(ngroups)]
for t in threads:
    t.start()
for t in threads:
    t.join()
dt = time.time() - t0
print("dtype %s: %d groups x %d rounds in %.1f s (%.1f grouped forwards/s), stats of group 0: %r" % (dtype, ngroups, rounds, dt, sum(done) / dt, groups[0].stats()))
ok = sum(bad) == 0 and all(d == rounds for d in done)
print("OK" if ok else "BAD", "mismatches", bad, "rounds done", done)
sys.exit(0 if ok else 1)
