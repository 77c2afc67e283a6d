"""Quick look at the fp16 path: error against the fp32 CPU oracle and speed (diagnostics)."""
import os, sys, time
sys.path.insert(0, 'deepcut-cnn_amd/python'); sys.path.insert(0, '.')
import numpy as np
import caffe
from deepcut_tools import deepercut_prototxt, synth_weights
from oracle import oracle as O
caffe.set_mode_gpu()
layers = synth_weights(152, 0)
for (h, w) in [(64, 64), (104, 136), (240, 320)]:
    proto = deepercut_prototxt(152, h, w)
    img = (np.random.RandomState(1).randn(1, 3, h, w) * 50).astype(np.float32)
    O.set_threads(16)
    ref = O.OracleNet(proto, layers).forward(data=img)
    for dt in ('f32', 'f16'):
        net = caffe.Net(proto, caffe.TEST, from_text=True, dtype=dt)
        for name, _t, blobs in layers:
            for p, b in zip(net.params[name], blobs):
                p.data[...] = b
        net.blobs['data'].data[...] = img
        out = net.forward()
        print(h, w, dt, {k: '%.2e (range %.2f)' % (float(np.abs(out[k] - ref[k]).max()), float(np.abs(ref[k]).max())) for k in out}, flush=True)
