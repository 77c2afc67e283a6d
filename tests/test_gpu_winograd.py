"""-m gpu: the Winograd F(2x2,3x3) form of the stride-1 3x3 convolutions (kernels.hip wino_f23_kernel), forced with
DC_WINOGRAD=1 (8 waves per workgroup) and DC_WINOGRAD=2 (the 16-wave form, tile name wino_f23_w16); by default either is used
only where the per-shape timing finds it faster.  Against the CPU oracle.
Winograd changes the rounding (transforms in fp32), not the mathematics: the bound stays the path's 1e-3, measured ~1e-5."""
import os

import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["1", "2"], ids=["w8", "w16"])
def _force(monkeypatch, request):
    monkeypatch.setenv("DC_WINOGRAD", request.param)
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    return "wino_f23<4x8x16_w16>" if request.param == "2" else "wino_f23<4x8x16>"


def _oracle(proto, layers, img):
    from oracle import oracle as O

    O.set_threads(min(16, os.cpu_count() or 1))
    return O.OracleNet(proto, layers).forward(data=img)


def _conv_net(n, cin, cout, h, w, dil, relu, resid):
    L = ['name: "w"', 'input: "data"'] + ["input_dim: %d" % d for d in (n, cin, h, w)]
    L.append('layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: %d kernel_size: 3 '
             'pad: %d dilation: %d bias_term: false } }' % (cout, dil, dil))
    L.append('layer { name: "bn" type: "BatchNorm" bottom: "c" top: "c" batch_norm_param { use_global_stats: true } }')
    L.append('layer { name: "sc" type: "Scale" bottom: "c" top: "c" scale_param { bias_term: true } }')
    out = "c"
    if resid:
        L.append('layer { name: "sum" type: "Eltwise" bottom: "data" bottom: "c" top: "sum" }')
        out = "sum"
    if relu:
        L.append('layer { name: "relu" type: "ReLU" bottom: "%s" top: "%s" }' % (out, out))
    return "\n".join(L) + "\n", out


CASES = [  # n, cin, cout, h, w, dilation, relu, residual
    (1, 256, 256, 34, 46, 1, True, False),   # the res4 shape of the benchmark
    (1, 64, 64, 31, 45, 1, True, False),     # odd sizes: ragged last tile row / column
    (2, 128, 128, 17, 9, 1, False, False),   # batch 2, narrower than one tile block
    (1, 32, 16, 5, 3, 1, True, False),       # smaller than a tile block in both directions, minimum channel counts
    (1, 512, 512, 34, 46, 2, True, False),   # res5: dilation 2 = four interleaved phase images
    (3, 64, 96, 13, 21, 2, False, False),    # dilation 2, odd sizes, batch 3
    (1, 64, 64, 20, 28, 1, True, True),      # fused shortcut + ReLU epilogue
    (1, 96, 48, 11, 50, 3, False, False),    # dilation 3
]


@pytest.mark.parametrize("case", CASES)
def test_single_layers_match_oracle(gpu_caffe, case, _force):
    n, cin, cout, h, w, dil, relu, resid = case
    if resid and cin != cout:
        pytest.skip("residual needs equal channel counts")
    proto, out = _conv_net(n, cin, cout, h, w, dil, relu, resid)
    rs = np.random.RandomState(cin + h)
    weights = [("c", "Convolution", [(rs.randn(cout, cin, 3, 3) / np.sqrt(9.0 * cin)).astype(np.float32)]),
               ("bn", "BatchNorm", [rs.randn(cout).astype(np.float32) * 0.1, rs.uniform(0.5, 1.5, cout).astype(np.float32),
                                    np.array([1.0], np.float32)]),
               ("sc", "Scale", [rs.uniform(0.5, 1.5, cout).astype(np.float32), rs.randn(cout).astype(np.float32) * 0.1])]
    net = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True)
    for name, _t, blobs in weights:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    x = rs.randn(n, cin, h, w).astype(np.float32)
    net.blobs["data"].data[...] = x
    net.forward()
    assert _force in net.plan_text(), "the layer was not lowered to the Winograd kernel"
    ref = _oracle(proto, weights, x)[out]
    got = net.blobs[out].data
    assert got.shape == ref.shape
    err = float(np.abs(got - ref).max())
    assert err <= 1e-4 * max(1.0, float(np.abs(ref).max())), err


@pytest.mark.parametrize("hw", [(104, 136), (240, 320)])
def test_full_net_with_every_eligible_layer_in_winograd_form(gpu_caffe, synth152, hw, _force):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True)
    img = rand_image(3, h, w)
    net.blobs["data"].data[...] = img
    net.forward()
    assert sum(_force in ln for ln in net.plan_text().splitlines()) == 50  # 47 plain + 3 dilated 3x3 layers
    ref = _oracle(proto, layers, img)
    for k in ("prob", "loc_pred", "next_pred"):
        assert float(np.abs(net.blobs[k].data - ref[k]).max()) <= 1e-3, k


def test_off_switch_and_the_float16_form(gpu_caffe, synth152, monkeypatch):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    monkeypatch.setenv("DC_WINOGRAD", "0")
    net = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True)
    assert "wino" not in net.plan_text()
    monkeypatch.setenv("DC_WINOGRAD", "2")
    half = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True, dtype="f16")
    assert "wino_f23" not in half.plan_text() and "wino_h23<" in half.plan_text()  # float16 nets have a form of their own (wino_f16.hip)
    monkeypatch.setenv("DC_WINOGRAD", "0")
    half = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True, dtype="f16")
    assert "wino" not in half.plan_text()


def test_the_two_forms_agree_and_are_deterministic(gpu_caffe, monkeypatch):
    """The 16-wave form splits the channel sum of every staged step between two wave groups and adds the groups' partial sums in a
    fixed order: run twice it gives the same bits, and it differs from the 8-wave form by float32 rounding only."""
    proto, out = _conv_net(2, 256, 256, 34, 46, 1, True, False)
    rs = np.random.RandomState(5)
    wts = (rs.randn(256, 256, 3, 3) / np.sqrt(9.0 * 256)).astype(np.float32)
    x = rs.randn(2, 256, 34, 46).astype(np.float32)
    res = {}
    for mode in ("1", "2"):
        monkeypatch.setenv("DC_WINOGRAD", mode)
        net = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True)
        net.params["c"][0].data[...] = wts
        net.params["bn"][0].data[...] = 0.0
        net.params["bn"][1].data[...] = 1.0
        net.params["bn"][2].data[...] = 1.0
        net.params["sc"][0].data[...] = 1.0
        net.params["sc"][1].data[...] = 0.0
        net.blobs["data"].data[...] = x
        net.forward()
        a = net.blobs[out].data.copy()
        net.blobs["data"].data[...] = x
        net.forward()
        assert np.array_equal(a, net.blobs[out].data), mode
        res[mode] = a
    assert float(np.abs(res["1"] - res["2"]).max()) <= 1e-5 * max(1.0, float(np.abs(res["1"]).max()))
