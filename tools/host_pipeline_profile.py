"""ON THE GPU BOX: the host-in / host-out pipeline (deepcut_tools.Pipeline.submit_host, 4 executors, pinned buffers) as a plain closed
loop, for rocprofv3 --kernel-trace --memory-copy-trace: are the copies on the DMA engines beside the other executors' kernels?
    rocprofv3 --kernel-trace --memory-copy-trace -d <dir> -o h -- python tools/host_pipeline_profile.py [requests]
    python tools/host_pipeline_profile.py --summarise <dir>/.../h_results.db"""
import os
import sqlite3
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)


def summarise(path):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    ker = sorted(c.execute("select start, end from kernels").fetchall())
    ctab = [t for t in tables if t in ("memory_copies", "memory_copy")]
    if not ctab:
        print("no memory-copy table in", path, tables[:20])
        return
    cols = [d[1] for d in c.execute("pragma table_info('%s')" % ctab[0])]
    size_col = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
    cop = sorted(c.execute("select start, end%s from %s" % (", " + size_col if size_col else "", ctab[0])).fetchall())
    # steady state: the last 60 % of the run
    t_lo = ker[0][0] + 0.4 * (ker[-1][1] - ker[0][0])
    ker = [(s, e) for s, e in ker if s >= t_lo]
    cop = [r for r in cop if r[0] >= t_lo and r[1] - r[0] > 20000]  # the 4.8 MB uploads and the map downloads (> 20 us), not the small ones
    wall = max(ker[-1][1], cop[-1][1]) - min(ker[0][0], cop[0][0])

    def union(iv):
        tot, cur_s, cur_e = 0, None, None
        for s, e in sorted(iv):
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    tot += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        return tot + (cur_e - cur_s if cur_e is not None else 0)

    kbusy, cbusy = union(ker), union([(r[0], r[1]) for r in cop])
    both = kbusy + cbusy - union(ker + [(r[0], r[1]) for r in cop])
    print("# steady state of the host-in / host-out pipeline (rocprofv3 --kernel-trace --memory-copy-trace), %s" % os.path.basename(path))
    print("wall %.1f ms: some kernel resident %.1f %%, some large copy in flight %.1f %%, BOTH at once %.1f %% of the wall time (%.0f %% of the copy time)"
          % (wall / 1e6, 100.0 * kbusy / wall, 100.0 * cbusy / wall, 100.0 * both / wall, 100.0 * both / max(cbusy, 1)))
    if size_col:
        nbytes = sum(r[2] for r in cop)
        print("large copies: %d, %.1f MB, %.2f GB/s while one is in flight, %.2f GB/s over the wall time" % (len(cop), nbytes / 1e6, nbytes / max(cbusy, 1), nbytes / wall))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
        sys.exit(0)
    import numpy as np

    import caffe
    from deepcut_tools import Pipeline, deepercut_prototxt, synth_weights

    nreq = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    caffe.set_mode_gpu()
    caffe.set_device(0)
    H, W = 544, 736
    net = caffe.Net(deepercut_prototxt(152, H, W, 1), caffe.TEST, from_text=True, hipgraph=1)
    for name, _t, blobs in synth_weights(152, seed=0):
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    pipe = Pipeline(net, depth=4, coalesce=1)
    slots = []
    for i in range(8):
        x = caffe.pinned_empty((1, 3, H, W))
        x[...] = (np.random.RandomState(i).randn(1, 3, H, W) * 50).astype(np.float32)
        slots.append((x, [caffe.pinned_empty(tuple(net.blobs[k].shape)) for k in ("prob", "loc_pred", "next_pred")]))

    def loop(count):
        sent = done = 0
        while done < count:
            while sent < count and sent - done < len(slots):
                x, o = slots[sent % len(slots)]
                pipe.submit_host(x, o[0], o[1], o[2], tag=sent)
                sent += 1
            pipe.wait_one()
            done += 1

    loop(32)
    t0 = time.perf_counter()
    loop(nreq)
    dt = time.perf_counter() - t0
    print("%d host-in / host-out requests in %.3f s = %.1f images/s (stream choice: %s)" % (nreq, dt, nreq / dt, pipe.stream_choice))
