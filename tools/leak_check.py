"""Does anything stay behind when executors go?  (diagnostics)  Cycles of: a net from the weights file, clones, a group, forwards
through the host / device / image / grouped entries at two shape sets, everything dropped — and the device's free memory
(hipMemGetInfo through torch), the process's resident set and its open file descriptors after every cycle.  The first cycle pays
for what is process-wide by design (the kernels' code objects, the lanes' candidate streams, the utility stream); from the
second cycle on nothing may grow.  usage: leak_check.py [f16|f32] [cycles]"""
import gc
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import caffe  # noqa: E402
from deepcut_tools import deepercut_prototxt, synth_weights, write_caffemodel  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 8
caffe.set_mode_gpu()
caffe.set_device(0)
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "synth.caffemodel")
write_caffemodel(path, "ResNet-152", synth_weights(152, seed=0))
SETS = [[(1, 64, 80), (2, 40, 56), (1, 104, 136), (2, 72, 104)], [(2, 72, 104), (1, 64, 80), (1, 104, 136), (2, 40, 56)]]
rs = np.random.RandomState(0)
stream = torch.cuda.Stream(dev)


def rss_mb():
    with open("/proc/self/status") as f:
        for ln in f:
            if ln.startswith("VmRSS"):
                return int(ln.split()[1]) / 1024.0
    return 0.0


def one_cycle():
    net = caffe.Net(deepercut_prototxt(152, 64, 80, 1), path, caffe.TEST, from_text=True, hipgraph=1, dtype=dtype)
    grp = caffe.NetGroup.for_shapes(net, SETS[0])
    for shapes in SETS:
        xs = [(rs.randn(n, 3, h, w) * 50).astype(np.float32) for (n, h, w) in shapes]
        grp.forward_batch(xs)
        dx = [torch.from_numpy(x).to(dev) for x in xs]
        outs = [[torch.empty(n, c, h // 8, w // 8, device=dev) for c in (14, 28, 364)] for (n, h, w) in shapes]
        grp.forward_device([x.data_ptr() for x in dx], shapes, [o[0].data_ptr() for o in outs], [o[1].data_ptr() for o in outs],
                           [o[2].data_ptr() for o in outs], stream=stream.cuda_stream)
        stream.synchronize()
        n, h, w = shapes[0]
        grp.nets[1].forward_batch(xs[0])
        grp.nets[2].forward_images(rs.randint(0, 256, size=(n, h, w, 3), dtype=np.uint8), 1.0)
        net.blobs["data"].reshape(n, 3, h, w)
        net.blobs["data"].data[...] = xs[0]
        net.forward()
        del dx, outs
    alive = torch.cuda.mem_get_info(dev)[0] / 2**20  # (shows that the figure below moves at all)
    del grp, net
    gc.collect()
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    return alive


rows = []
for c in range(cycles):
    alive = one_cycle()
    free, total = torch.cuda.mem_get_info(dev)
    rows.append((free / 2**20, rss_mb(), len(os.listdir("/proc/self/fd"))))
    print("cycle %2d: device free %.1f MiB (%.1f with the executors alive), host RSS %.1f MiB, %d file descriptors" % (c, rows[-1][0], alive, *rows[-1][1:]))
# from the second cycle on: no growth (2 MiB of device slack for allocator granularity; 5 % / 64 MiB of RSS for the host allocator's pools)
dev_drift = rows[1][0] - rows[-1][0]
rss_drift = rows[-1][1] - rows[1][1]
fd_drift = rows[-1][2] - rows[1][2]
ok = dev_drift <= 2.0 and rss_drift <= max(64.0, 0.05 * rows[1][1]) and fd_drift <= 0
print("%s: over cycles 1..%d device memory %+.1f MiB, RSS %+.1f MiB, file descriptors %+d" % ("OK" if ok else "LEAK", cycles - 1, dev_drift, rss_drift, fd_drift))
sys.exit(0 if ok else 1)
