"""-m gpu: the float16 Winograd F(2x2,3x3) form of the stride-1 3x3 convolutions (csrc/wino_f16.hip, tile name wino_h23: fp16
operands, fp32 accumulate, fp32 epilogue), forced with DC_WINOGRAD=1; by default it is used only where the per-shape timing
finds it faster.  Reference arithmetic: conv_layer.cpp:25-40, base_conv_layer.cpp:257-280 (the CPU oracle restates them).

Tolerances are the float16 path's own (tests/test_gpu_fp16.py): single layers <= 2e-3 x output range, prob <= 2.5e-3,
loc_pred / next_pred <= 4e-3 x range.  Winograd changes the rounding, not the mathematics: the transformed patch B^T d B is
formed by packed float16 adds (two roundings per value) from pixels pre-multiplied by 1/4, the products accumulate in float32."""
import os

import numpy as np
import pytest

from conftest import rand_image
from oracle import oracle as O
from test_gpu_fp16 import _check_maps, _large_activation_weights
from test_gpu_winograd import _conv_net

pytestmark = pytest.mark.gpu
LABEL = "wino_h23<"


@pytest.fixture(autouse=True)
def _force(monkeypatch):
    monkeypatch.setenv("DC_WINOGRAD", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")


CASES = [  # n, cin, cout, h, w, dilation, relu
    (8, 256, 256, 34, 46, 1, True),    # the res4 shape of configs[2]'s 544x736 member: two 128-channel blocks
    (1, 64, 64, 31, 45, 1, True),      # odd sizes (ragged last tile row / column), one 64-channel block (NF = 1)
    (2, 128, 128, 17, 9, 1, False),    # batch 2, narrower than one tile block, no ReLU
    (1, 64, 64, 5, 3, 1, True),        # smaller than a tile block in both directions, two staged steps (the least a float16 layer has)
    (1, 192, 192, 12, 20, 1, True),    # six staged steps (the ring wraps twice), 192 = 3 x 64 channels
    (2, 512, 512, 34, 46, 2, True),    # res5: dilation 2 = four interleaved phase images
    (3, 64, 128, 13, 21, 2, False),    # dilation 2, odd sizes, batch 3
    (1, 128, 64, 11, 50, 3, False),    # dilation 3
]


def _weights(rs, cin, cout):
    return [("c", "Convolution", [(rs.randn(cout, cin, 3, 3) / np.sqrt(9.0 * cin)).astype(np.float32)]),
            ("bn", "BatchNorm", [rs.randn(cout).astype(np.float32) * 0.1, rs.uniform(0.5, 1.5, cout).astype(np.float32),
                                 np.array([1.0], np.float32)]),
            ("sc", "Scale", [rs.uniform(0.5, 1.5, cout).astype(np.float32), rs.randn(cout).astype(np.float32) * 0.1])]


@pytest.mark.parametrize("case", CASES)
def test_single_layers_match_oracle(gpu_caffe, case):
    n, cin, cout, h, w, dil, relu = case
    proto, out = _conv_net(n, cin, cout, h, w, dil, relu, False)
    rs = np.random.RandomState(cin + h)
    weights = _weights(rs, cin, cout)
    net = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16")
    for name, _t, blobs in weights:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    x = rs.randn(n, cin, h, w).astype(np.float32)
    net.blobs["data"].data[...] = x
    net.forward()
    assert LABEL in net.plan_text(), "the layer was not lowered to the float16 Winograd kernel"
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto, weights).forward(data=x)[out]
    got = net.blobs[out].data
    assert got.shape == ref.shape and np.isfinite(got).all()
    err = float(np.abs(got - ref).max())
    print("wino_h23 %s: max|hip - oracle| = %.3e (range %.2f)" % (case, err, float(np.abs(ref).max())))
    assert err <= 2e-3 * max(1.0, float(np.abs(ref).max())), err
    assert err > 1e-6, "suspiciously exact: is the float16 kernel really running?"


def test_a_shortcut_operand_keeps_the_direct_kernel(gpu_caffe):
    """the kernel has no shortcut operand (no 3x3 layer of the path has one): such a launch stays a gather-GEMM"""
    proto, out = _conv_net(1, 64, 64, 20, 28, 1, True, True)
    net = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16")
    assert "wino" not in net.plan_text()


@pytest.mark.parametrize("hw,n", [((104, 136), 2), ((240, 320), 1)])
def test_full_net_with_every_eligible_layer_in_winograd_form(gpu_caffe, synth152, hw, n):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w, n)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16")
    img = rand_image(3, h, w, n=n)
    out = net.forward_batch(img)
    assert sum(LABEL in ln for ln in net.plan_text().splitlines()) == 50  # 47 plain + 3 dilated 3x3 layers
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto, layers).forward(data=img)
    _check_maps(out, ref)


@pytest.mark.parametrize("gain", [32.0, 1024.0])
def test_trained_weight_like_magnitudes(gpu_caffe, synth152, tmp_path, gain):
    """As tests/test_gpu_fp16.py::test_fp16_on_trained_weight_like_magnitudes, every 3x3 layer in the Winograd form: with the trunk at
    up to 70 % of float16's largest finite value the transformed patches (sums of four pixels) must not overflow — the kernel
    stages the pixels pre-multiplied by 1/4 — and the transformed filters keep their precision through their own row scale."""
    from deepcut_tools import deepercut_prototxt, write_caffemodel

    _, layers = synth152
    big = _large_activation_weights(layers, gain)
    path = str(tmp_path / "big.caffemodel")
    write_caffemodel(path, "ResNet-152", big)
    h, w = 104, 136
    proto = deepercut_prototxt(152, h, w, 1)
    img = rand_image(33, h, w)
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto, big).forward(data=img)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16", fuse=0)
    net.blobs["data"].data[...] = img
    out = net.forward()
    assert sum(LABEL in ln for ln in net.plan_text().splitlines()) == 50
    for k in ("prob", "loc_pred", "next_pred"):
        assert np.isfinite(out[k]).all(), k
    _check_maps(out, ref)
    for name in ("res4b35", "res5c"):
        r = ref[name]
        got = net.blobs[name].data
        assert np.isfinite(got).all(), name
        assert float(np.abs(got - r).max()) <= 1e-2 * float(np.abs(r).max()), name


def test_autotuner_times_the_form_and_set_tile_takes_it(gpu_caffe, monkeypatch):
    """unset DC_WINOGRAD: the form competes with the direct tiles per shape (the tune report lists its timing), and
    set_tile / the tune-cache name `wino_h23` select it"""
    monkeypatch.delenv("DC_WINOGRAD")
    monkeypatch.delenv("DC_AUTOTUNE")
    proto, out = _conv_net(8, 256, 256, 34, 46, 1, True, False)
    net = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16")
    rs = np.random.RandomState(1)
    for name, _t, blobs in _weights(rs, 256, 256):
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    x = rs.randn(8, 256, 34, 46).astype(np.float32)
    net.blobs["data"].data[...] = x
    net.forward()
    a = net.blobs[out].data.copy()
    (ent,) = [e for e in net.tune_report() if "+w" in e["signature"]]
    assert any(t[0] == "wino_h23" for t in ent["timed"]), ent
    other = "wino_h23" if ent["tile"] != "wino_h23" else [t[0] for t in ent["timed"] if t[0] != "wino_h23"][0]
    net.set_tile(ent["signature"], other)
    net.blobs["data"].data[...] = x
    net.forward()
    b = net.blobs[out].data
    assert ("wino_h23<" in net.plan_text()) == (other == "wino_h23")
    assert float(np.abs(a - b).max()) <= 4e-3 * max(1.0, float(np.abs(a).max()))
