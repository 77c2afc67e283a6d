#!/usr/bin/env python
"""Where does dc_net_forward_batch(host buffers) spend its time?  (review r3, weak 2: the float16 batch-8 host entry
took 39 ms against 4 ms device-resident.)  Times, per dtype: the device-resident forward, the host entry with no /
one / all maps wanted, with fresh and with reused destination arrays, and prints the net's counters around each."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=6)
    args = ap.parse_args()
    import __graft_entry__ as ge

    ge.build()
    import caffe
    from caffe import pycaffe as P
    from deepcut_tools import deepercut_prototxt, synth_weights

    caffe.set_mode_gpu()
    caffe.set_device(0)
    B, H, W = args.batch, 544, 736
    layers = synth_weights(152, seed=0)
    net = caffe.Net(deepercut_prototxt(152, H, W, B), caffe.TEST, from_text=True, hipgraph=1, dtype=args.dtype)
    for name, _t, blobs in layers:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    dev = torch.device("cuda", 0)
    x = (torch.randn(B, 3, H, W) * 50)
    xd = x.to(dev)
    xh = x.numpy()
    shp = {k: net.blobs[k].shape for k in ("prob", "loc_pred", "next_pred")}
    od = {k: torch.empty(shp[k], device=dev) for k in shp}

    def t(label, fn, reps=args.reps):
        fn()
        torch.cuda.synchronize()
        s0 = net.stats()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        s1 = net.stats()
        d = {k: s1[k] - s0[k] for k in s1 if s1[k] != s0[k]}
        print("%-58s min %7.2f  med %7.2f  max %7.2f ms  counters moved: %s" % (label, min(ts), sorted(ts)[len(ts) // 2], max(ts), d or "-"), flush=True)

    t("device-resident forward_device (maps to device)", lambda: net.forward_device(xd.data_ptr(), B, H, W, od["prob"].data_ptr(), od["loc_pred"].data_ptr(), od["next_pred"].data_ptr()))
    t("forward_batch(host) want=() [H2D only]", lambda: net.forward_batch(xh, want=()))
    t("forward_batch(host) want=(prob,)", lambda: net.forward_batch(xh, want=("prob",)))
    t("forward_batch(host) want=(prob,loc_pred)", lambda: net.forward_batch(xh, want=("prob", "loc_pred")))
    t("forward_batch(host) all three maps, fresh np.empty per call", lambda: net.forward_batch(xh))
    # reused destinations through the raw C entry (no numpy allocation inside the timed call)
    lib = P._lib
    outs = {k: np.zeros(shp[k], np.float32) for k in shp}

    def raw():
        P._check(lib.dc_net_forward_batch(net._h, xh.ctypes.data_as(C.c_void_p), B, H, W, 0, outs["prob"].ctypes.data_as(C.c_void_p),
                                          outs["loc_pred"].ctypes.data_as(C.c_void_p), outs["next_pred"].ctypes.data_as(C.c_void_p), None))

    t("dc_net_forward_batch(host) all maps, REUSED (touched) arrays", raw)
    pin = {k: torch.empty(shp[k], pin_memory=True) for k in shp}
    xp = x.pin_memory()

    def rawpin():
        P._check(lib.dc_net_forward_batch(net._h, C.c_void_p(xp.data_ptr()), B, H, W, 0, C.c_void_p(pin["prob"].data_ptr()),
                                          C.c_void_p(pin["loc_pred"].data_ptr()), C.c_void_p(pin["next_pred"].data_ptr()), None))

    t("dc_net_forward_batch(host) all maps, PINNED torch buffers", rawpin)

    def pyc():
        net.blobs["data"].data[...] = xh
        net.forward()
        return [net.blobs[k].data for k in shp]

    t("pycaffe: data[...]=x; forward(); read 3 blobs", pyc)
    # plain copies of the same sizes for scale
    big = np.empty(shp["next_pred"], np.float32)
    dn = od["next_pred"]

    def d2h_fresh():
        a = np.empty(shp["next_pred"], np.float32)
        torch.from_numpy(a).copy_(dn)

    def d2h_reuse():
        torch.from_numpy(big).copy_(dn)

    t("torch D2H of next_pred into fresh np.empty (%.0f MB)" % (big.nbytes / 1e6), d2h_fresh)
    t("torch D2H of next_pred into a reused array", d2h_reuse)


if __name__ == "__main__":
    main()
