#!/usr/bin/env python
"""Per-launch hipEvent tables of the float16 4-scale pyramid (BASELINE configs[2]): the GROUPED plan (one multi-problem launch per
layer over the four scales) next to the four scale-by-scale plans.  Writes <out>/group_per_launch.txt and
<out>/scale<k>_per_launch.txt (tools/breakdown.py aggregates them) and prints whole-pyramid timings of both forms."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/group_profile")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-members", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="plain launches instead of a hipGraph (counter collection)")
    ap.add_argument("--pmc-run", type=int, default=0, help="only this many grouped forwards, nothing else (the command rocprofv3 --pmc wraps; "
                                                           "seed DC_TUNE_CACHE from a plain run so that no timing launches are profiled)")
    ap.add_argument("--scales", default="0,1,2,3", help="which of the four pyramid scales (indices, smallest first) take part")
    ap.add_argument("--lanes", type=int, default=0, help="also measure ONE pyramid batch as this many sub-groups on their own streams "
                                                         "(members dealt largest+smallest, ...), one batch at a time and two in flight")
    ap.add_argument("--inflight", type=int, default=1, help="also measure this many groups in flight, each on its own stream")
    ap.add_argument("--pyramids", type=int, default=1, help="pyramid batches coalesced into ONE group (members = 4 x this)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import torch
    import __graft_entry__ as ge

    ge.build()
    import caffe
    from deepcut_tools import deepercut_prototxt, synth_weights

    caffe.set_mode_gpu()
    caffe.set_device(0)
    B = args.batch
    shapes = [[(272, 368), (408, 552), (544, 736), (680, 920)][int(i)] for i in args.scales.split(",")] * args.pyramids
    layers = synth_weights(152, seed=0)
    net = caffe.Net(deepercut_prototxt(152, 544, 736, B), caffe.TEST, from_text=True, hipgraph=0 if args.no_graph else 1, dtype=args.dtype)
    for name, _t, blobs in layers:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(10)
    xs = [(torch.randn(B, 3, h, w, generator=g) * 50).to(dev) for h, w in shapes]
    grp = caffe.NetGroup.for_shapes(net, [(B, h, w) for h, w in shapes][::-1])  # largest first on the original net
    grp.nets.reverse()
    grp = caffe.NetGroup(grp.nets)
    gshapes = [(B, h, w) for h, w in shapes]

    def grouped():
        grp.forward_device([x.data_ptr() for x in xs], gshapes)

    def by_scale():
        for m, x, s in zip(grp.nets, xs, gshapes):
            m.forward_device(x.data_ptr(), *s)

    if args.pmc_run:
        for _ in range(args.pmc_run):
            grouped()
        torch.cuda.synchronize()
        return
    for name, fn in (("grouped", grouped), ("scale by scale (same executors)", by_scale)):
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
        print("%-36s %.3f ms per pyramid batch (min %.3f max %.3f) = %.1f image-pyramids/s" % (name, sorted(ts)[2], min(ts), max(ts), B * args.pyramids / sorted(ts)[2] * 1e3), flush=True)
    if args.lanes > 1:
        # one pyramid batch = `lanes` sub-groups running concurrently on their own streams: the tails of one fill with the other
        order = sorted(range(len(gshapes)), key=lambda i: -gshapes[i][1] * gshapes[i][2])
        lanes = [[] for _ in range(args.lanes)]
        for j, i in enumerate(order):  # snake: largest + smallest together
            r, c = divmod(j, args.lanes)
            lanes[c if r % 2 == 0 else args.lanes - 1 - c].append(i)

        def make(copy):
            out = []
            for ln in lanes:
                nets_l = [grp.nets[i].clone() if copy else grp.nets[i] for i in ln]
                for m, i in zip(nets_l, ln):
                    m.reserve(*gshapes[i])
                out.append((caffe.NetGroup(nets_l, lanes=1), ln))
            return out

        sets = [make(False), make(True)]
        lstreams = [[torch.cuda.Stream(dev) for _ in lanes] for _ in sets]

        def lane_step(k):
            for (g2, ln), st in zip(sets[k % 2], lstreams[k % 2]):
                g2.forward_device([xs[i].data_ptr() for i in ln], [gshapes[i] for i in ln], stream=st.cuda_stream)

        for mode, sync_each in (("%d lanes, one pyramid batch at a time" % args.lanes, True), ("%d lanes, two pyramid batches in flight" % args.lanes, False)):
            for k in range(4):
                lane_step(k)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                for k in range(10):
                    lane_step(k if not sync_each else 0)
                    if sync_each:
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 10 * 1e3)
            print("%-44s %.3f ms per pyramid batch (min %.3f max %.3f) = %.1f image-pyramids/s" % (mode, sorted(ts)[2], min(ts), max(ts), B * args.pyramids / sorted(ts)[2] * 1e3), flush=True)
    if args.inflight > 1:
        grps = [grp] + [caffe.NetGroup.for_shapes(net.clone(), gshapes[::-1]) for _ in range(args.inflight - 1)]
        for g2 in grps[1:]:
            g2.nets.reverse()
        grps = [grp] + [caffe.NetGroup(g2.nets) for g2 in grps[1:]]
        streams = [torch.cuda.Stream(dev) for _ in grps]

        def inflight(k):
            g2, st = grps[k % len(grps)], streams[k % len(grps)]
            g2.forward_device([x.data_ptr() for x in xs], gshapes, stream=st.cuda_stream)

        for k in range(2 * len(grps)):
            inflight(k)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for k in range(10):
                inflight(k)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 10 * 1e3)
        print("%-36s %.3f ms per pyramid batch (min %.3f max %.3f) = %.1f image-pyramids/s" % ("%d groups in flight" % len(grps), sorted(ts)[2], min(ts), max(ts),
                                                                                                   B * args.pyramids / sorted(ts)[2] * 1e3), flush=True)
    grouped()
    torch.cuda.synchronize()
    with open(os.path.join(args.out, "group_per_launch.txt"), "w") as f:
        f.write(grp.plan_text())
        f.write(grp.profile_text(args.iters))
    if not args.no_members:
        for k, (m, x, s) in enumerate(zip(grp.nets, xs, gshapes)):
            m.forward_device(x.data_ptr(), *s)
            torch.cuda.synchronize()
            with open(os.path.join(args.out, "scale%d_per_launch.txt" % k), "w") as f:
                f.write(m.plan_text())
                f.write(m.profile_text(args.iters))
    print(grp.stats())


if __name__ == "__main__":
    main()
