"""-m gpu: the whole DeeperCut ResNet-152 forward on MI355X (through the C-ABI / pycaffe shim) against
the CPU oracle on the same seeded input and the same synthetic weights.  Bound: 1e-3 max-abs on every
output map (BASELINE.json north_star), fp32."""
import os

import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _oracle(proto, layers, img):
    from oracle import oracle as O

    O.set_threads(min(16, os.cpu_count() or 1))
    return O.OracleNet(proto, layers).forward(data=img)


@pytest.mark.parametrize("hw", [(64, 64), (104, 136), (72, 200)])
def test_unfused_every_blob_matches_oracle(gpu_caffe, synth152, hw):
    """DC_OPT_FUSE 0 materialises every Caffe-visible blob: compare ALL of them (in-place chains hold
    their final value on both sides)."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, fuse=0)
    img = rand_image(1, h, w)
    net.blobs["data"].data[...] = img
    net.forward()
    ref = _oracle(proto, layers, img)
    worst = ("", 0.0)
    for name, r in ref.items():
        got = net.blobs[name].data
        assert got.shape == r.shape, name
        scale = max(1.0, float(np.abs(r).max()))
        err = float(np.abs(got - r).max()) / scale
        if err > worst[1]:
            worst = (name, err)
        assert err <= TOL, "%s: rel-to-range err %g" % (name, err)
    for name in ("prob", "loc_pred", "next_pred"):
        assert float(np.abs(net.blobs[name].data - ref[name]).max()) <= TOL
    print("worst blob", worst)


@pytest.mark.parametrize("hw", [(64, 64), (104, 136), (240, 320)])
def test_fused_outputs_match_oracle(gpu_caffe, synth152, hw):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True)
    img = rand_image(2, h, w)
    net.blobs["data"].data[...] = img
    out = net.forward()
    ref = _oracle(proto, layers, img)
    assert sorted(out) == ["loc_pred", "next_pred", "prob"]
    for k in out:
        assert out[k].shape == ref[k].shape
        err = float(np.abs(out[k] - ref[k]).max())
        print(k, out[k].shape, "max abs err", err, "range", float(np.abs(ref[k]).max()))
        assert err <= TOL, k
