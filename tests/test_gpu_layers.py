"""-m gpu: every layer type of the path stand-alone (Layer::Forward_gpu surface = a one-layer prototxt,
DC_OPT_FUSE 0, weights injected through net.params) and every one of the 26 distinct convolution /
deconvolution configurations of the DeeperCut net at reduced spatial size, against the
CPU oracle.  fp32; bound 1e-4 relative to the output range (the reference's own conv-test bound)."""
import zlib

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _net(caffe, text, fuse=0):
    return caffe.Net(text, caffe.TEST, from_text=True, fuse=fuse)


def _inp(name, shape):
    return 'input: "%s" input_dim: %d input_dim: %d input_dim: %d input_dim: %d\n' % ((name,) + tuple(shape))


def _check(got, ref, tol=1e-4):
    assert got.shape == ref.shape
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max()) / scale
    assert err <= tol, "max err %g (range %g)" % (err, scale)


# (kind, k, s, p, d, bias, cin, cout, h, w)  — the 26 T2 rows at small spatial sizes, odd sizes included
T2 = [
    ("conv", 7, 2, 3, 1, False, 3, 64, 41, 54),
    ("conv", 1, 1, 0, 1, False, 64, 256, 13, 17),
    ("conv", 1, 1, 0, 1, False, 64, 64, 13, 17),
    ("conv", 3, 1, 1, 1, False, 64, 64, 13, 17),
    ("conv", 1, 1, 0, 1, False, 256, 64, 13, 17),
    ("conv", 1, 2, 0, 1, False, 256, 512, 13, 17),
    ("conv", 1, 2, 0, 1, False, 256, 128, 14, 18),
    ("conv", 3, 1, 1, 1, False, 128, 128, 9, 11),
    ("conv", 1, 1, 0, 1, False, 128, 512, 9, 11),
    ("conv", 1, 1, 0, 1, False, 512, 128, 9, 11),
    ("conv", 1, 2, 0, 1, False, 512, 1024, 9, 11),
    ("conv", 1, 2, 0, 1, False, 512, 256, 9, 11),
    ("conv", 3, 1, 1, 1, False, 256, 256, 7, 9),
    ("conv", 1, 1, 0, 1, False, 256, 1024, 7, 9),
    ("conv", 1, 1, 0, 1, False, 1024, 256, 7, 9),
    ("conv", 1, 1, 0, 1, False, 1024, 2048, 5, 6),
    ("conv", 1, 1, 0, 1, False, 1024, 512, 5, 6),
    ("conv", 3, 1, 2, 2, False, 512, 512, 7, 9),
    ("conv", 1, 1, 0, 1, False, 512, 2048, 5, 6),
    ("conv", 1, 1, 0, 1, False, 2048, 512, 5, 6),
    ("deconv", 3, 2, 0, 1, True, 2048, 14, 4, 5),
    ("conv", 1, 1, 0, 1, True, 512, 14, 9, 11),
    ("deconv", 3, 2, 0, 1, True, 2048, 28, 4, 5),
    ("conv", 1, 1, 0, 1, True, 512, 28, 9, 11),
    ("deconv", 3, 2, 0, 1, True, 2048, 364, 4, 5),
    ("conv", 1, 1, 0, 1, True, 512, 364, 9, 11),
]


@pytest.mark.parametrize("cfg", T2, ids=lambda c: "%s_k%ds%dp%dd%d_%dto%d" % (c[0], c[1], c[2], c[3], c[4], c[6], c[7]))
@pytest.mark.parametrize("batch", [1, 2])
def test_conv_deconv_configs(gpu_caffe, cfg, batch):
    kind, k, s, p, d, bias, cin, cout, h, w = cfg
    rs = np.random.RandomState(zlib.crc32(repr(cfg).encode()) & 0x7fffffff)  # (hash() of a tuple holding a str moves with PYTHONHASHSEED)
    typ = "Convolution" if kind == "conv" else "Deconvolution"
    text = _inp("x", (batch, cin, h, w)) + (
        'layer { name: "l" type: "%s" bottom: "x" top: "y" convolution_param { num_output: %d kernel_size: %d '
        "stride: %d pad: %d dilation: %d bias_term: %s } }" % (typ, cout, k, s, p, d, "true" if bias else "false"))
    net = _net(gpu_caffe, text)
    x = rs.randn(batch, cin, h, w).astype(np.float32)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    wt = (rs.randn(*wshape) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32) if bias else None
    net.params["l"][0].data[...] = wt
    if bias:
        net.params["l"][1].data[...] = b
    net.blobs["x"].data[...] = x
    got = net.forward()["y"]
    ref = (O.conv_forward if kind == "conv" else O.deconv_forward)(x, wt, b, s, p, d)
    _check(got, ref)


def test_deconv_closed_form_on_gpu(gpu_caffe):
    # src/caffe/test/test_deconvolution_layer.cpp:91-137 through the HIP path (Cin padded case: 3 channels
    # is outside the supported multiple-of-32 input, so the closed form is evaluated with 32 channels:
    # every count scales by 32/3)
    text = _inp("x", (2, 32, 6, 4)) + ('layer { name: "l" type: "Deconvolution" bottom: "x" top: "y" '
                                       "convolution_param { num_output: 4 kernel_size: 3 stride: 2 } }")
    net = _net(gpu_caffe, text)
    net.params["l"][0].data[...] = 1.0
    net.params["l"][1].data[...] = 0.1
    net.blobs["x"].data[...] = 1.0
    y = net.forward()["y"]
    assert y.shape == (2, 4, 13, 9)
    H, W = 13, 9
    for h in range(H):
        for w in range(W):
            n = 1
            ho = h % 2 == 0 and 0 < h < H - 1
            wo = w % 2 == 0 and 0 < w < W - 1
            n = 4 if (ho and wo) else 2 if (ho or wo) else 1
            assert np.allclose(y[:, :, h, w], 32.0 * n + 0.1, atol=1e-4), (h, w)


@pytest.mark.parametrize("hw", [(3, 5), (16, 16), (15, 20), (33, 47)])
def test_maxpool(gpu_caffe, hw):
    text = _inp("x", (2, 64) + hw) + 'layer { name: "p" type: "Pooling" bottom: "x" top: "y" pooling_param { pool: MAX kernel_size: 3 stride: 2 } }'
    net = _net(gpu_caffe, text)
    x = np.random.RandomState(1).randn(2, 64, *hw).astype(np.float32)
    net.blobs["x"].data[...] = x
    assert np.array_equal(net.forward()["y"], O.maxpool_forward(x, 3, 2))


def test_maxpool_reference_literals(gpu_caffe):
    # test_pooling_layer.cpp:49-119 (k2 s1) and :478-521 (k3 s2 pad 2), on 4 channels
    img = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)
    net = _net(gpu_caffe, _inp("x", (1, 4, 3, 5)) + 'layer { name: "p" type: "Pooling" bottom: "x" top: "y" pooling_param { kernel_size: 2 } }')
    net.blobs["x"].data[...] = img
    assert np.array_equal(net.forward()["y"][0, 2], np.array([[9, 5, 5, 8], [9, 5, 5, 8]], np.float32))
    img2 = np.array([[1, 2, 4], [2, 3, 2], [4, 2, 1]], np.float32)
    net = _net(gpu_caffe, _inp("x", (1, 4, 3, 3)) + 'layer { name: "p" type: "Pooling" bottom: "x" top: "y" pooling_param { kernel_size: 3 stride: 2 pad: 2 } }')
    net.blobs["x"].data[...] = img2
    assert np.array_equal(net.forward()["y"][0, 1], np.array([[1, 4, 4], [4, 4, 4], [4, 4, 1]], np.float32))


@pytest.mark.parametrize("sf", [0.0, 1.0, 999.98236])
def test_batchnorm_scale_relu_standalone(gpu_caffe, sf):
    C = 64
    text = _inp("x", (2, C, 5, 7)) + (
        'layer { name: "bn" type: "BatchNorm" bottom: "x" top: "y" batch_norm_param { use_global_stats: true } }'
        'layer { name: "sc" type: "Scale" bottom: "y" top: "z" scale_param { bias_term: true } }'
        'layer { name: "re" type: "ReLU" bottom: "z" top: "r" }')
    net = _net(gpu_caffe, text)
    rs = np.random.RandomState(2)
    x = rs.randn(2, C, 5, 7).astype(np.float32)
    m = (rs.randn(C) * max(sf, 1)).astype(np.float32)
    v = ((1 + rs.rand(C)) * max(sf, 1)).astype(np.float32)
    g = rs.randn(C).astype(np.float32)
    b = rs.randn(C).astype(np.float32)
    for p, val in zip(net.params["bn"], (m, v, np.array([sf], np.float32))):
        p.data[...] = val
    net.params["sc"][0].data[...] = g
    net.params["sc"][1].data[...] = b
    net.blobs["x"].data[...] = x
    net.forward()
    y = O.batchnorm_forward(x, m, v, np.float32(sf))
    z = O.scale_forward(y, g, b)
    tol = 2e-3 if sf == 0 else 1e-4
    _check(net.blobs["y"].data, y, tol)
    _check(net.blobs["z"].data, z, tol)
    _check(net.blobs["r"].data, O.relu_forward(z), tol)


def test_eltwise_crop_sigmoid_standalone(gpu_caffe):
    text = (_inp("a", (2, 14, 9, 11)) + _inp("b", (2, 14, 8, 10)) +
            'layer { name: "c" type: "Crop" bottom: "a" bottom: "b" top: "ac" }'
            'layer { name: "e" type: "Eltwise" bottom: "b" bottom: "ac" top: "s" }'
            'layer { name: "g" type: "Sigmoid" bottom: "s" top: "p" }')
    net = _net(gpu_caffe, text)
    rs = np.random.RandomState(3)
    a = rs.randn(2, 14, 9, 11).astype(np.float32)
    b = rs.randn(2, 14, 8, 10).astype(np.float32)
    net.blobs["a"].data[...] = a
    net.blobs["b"].data[...] = b
    net.forward()
    ac = O.crop_forward(a, b)
    assert np.array_equal(net.blobs["ac"].data, ac)
    assert np.array_equal(net.blobs["s"].data, O.eltwise_sum(b, ac))
    _check(net.blobs["p"].data, O.sigmoid_forward(O.eltwise_sum(b, ac)), 1e-6)
    assert net.outputs == ["p"]


@pytest.mark.parametrize("stage", ["res2_first", "res3_first_stride2", "res4_identity", "res5_dilated"])
@pytest.mark.parametrize("fuse", [0, 2])
def test_bottleneck_blocks(gpu_caffe, stage, fuse):
    """One full bottleneck block per stage type (G6): projection shortcut, stride-2 entry, identity
    shortcut, dilated conv5 — fused and unfused."""
    cin, width, stride, dil, proj, h, w = {
        "res2_first": (64, 64, 1, 1, True, 13, 17),
        "res3_first_stride2": (256, 128, 2, 1, True, 13, 17),
        "res4_identity": (1024, 256, 1, 1, False, 7, 9),
        "res5_dilated": (2048, 512, 1, 2, False, 7, 9),
    }[stage]
    L = []

    def conv(n, bot, top, co, k, p, s, d=1):
        L.append('layer { name: "%s" type: "Convolution" bottom: "%s" top: "%s" convolution_param { num_output: %d '
                 "kernel_size: %d pad: %d stride: %d dilation: %d bias_term: false } }" % (n, bot, top, co, k, p, s, d))
        L.append('layer { name: "bn_%s" type: "BatchNorm" bottom: "%s" top: "%s" batch_norm_param { use_global_stats: true } }' % (n, top, top))
        L.append('layer { name: "sc_%s" type: "Scale" bottom: "%s" top: "%s" scale_param { bias_term: true } }' % (n, top, top))

    short = "x"
    if proj:
        conv("b1", "x", "b1", width * 4, 1, 0, stride)
        short = "b1"
    conv("b2a", "x", "b2a", width, 1, 0, stride)
    L.append('layer { name: "r2a" type: "ReLU" bottom: "b2a" top: "b2a" }')
    conv("b2b", "b2a", "b2b", width, 3, dil, 1, dil)
    L.append('layer { name: "r2b" type: "ReLU" bottom: "b2b" top: "b2b" }')
    conv("b2c", "b2b", "b2c", width * 4, 1, 0, 1)
    L.append('layer { name: "sum" type: "Eltwise" bottom: "%s" bottom: "b2c" top: "out" }' % short)
    L.append('layer { name: "rout" type: "ReLU" bottom: "out" top: "out" }')
    text = _inp("x", (2, cin, h, w)) + "\n".join(L)
    net = _net(gpu_caffe, text, fuse)
    rs = np.random.RandomState(len(stage))
    weights = []
    for name in net.params:
        blobs = []
        for i, p in enumerate(net.params[name]):
            if name.startswith("bn_"):
                val = [rs.randn(*p.shape) * 0.1, 1 + rs.rand(*p.shape), np.ones(1)][i]
            elif name.startswith("sc_"):
                val = [1 + 0.1 * rs.randn(*p.shape), 0.1 * rs.randn(*p.shape)][i]
            else:
                val = rs.randn(*p.shape) / np.sqrt(np.prod(p.shape[1:]))
            val = np.asarray(val, np.float32)
            p.data[...] = val
            blobs.append(val)
        weights.append((name, "", blobs))
    x = rs.randn(2, cin, h, w).astype(np.float32)
    net.blobs["x"].data[...] = x
    got = net.forward()["out"]
    ref = O.OracleNet(text, weights).forward(x=x)
    _check(got, ref["out"])
    if fuse == 0:
        for k in ("b2a", "b2b", "b2c"):
            _check(net.blobs[k].data, ref[k])


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_nan_and_inf_propagate_through_a_relu_less_layer(gpu_caffe, dtype, monkeypatch):
    """the swapped-operand vector epilogue implemented "no ReLU" as max(x, -inf), which turns a NaN accumulator
    into -inf (and, after the next fused add + ReLU, into 0).  A ReLU-less convolution has to hand a NaN / inf on, as the
    reference's SGEMM does, on every tile family (forced: the LDS-DMA tiles carry that epilogue)."""
    names = [n for n, _ in gpu_caffe.conv_variants()]
    for tile in (["e64x64x32_w221_s4", "64x64x32_w221_p3"] if dtype == "f32" else ["d64x64x64_w221_s2", "h64x64x64_w221_p3"]):
        monkeypatch.setenv("DC_CONV_VARIANT", str(names.index(tile)))
        text = _inp("x", (1, 64, 9, 11)) + ('layer { name: "l" type: "Convolution" bottom: "x" top: "y" convolution_param '
                                            "{ num_output: 64 kernel_size: 1 bias_term: false } }")
        net = gpu_caffe.Net(text, gpu_caffe.TEST, from_text=True, fuse=0, dtype=dtype)
        x = np.random.RandomState(3).randn(1, 64, 9, 11).astype(np.float32)
        x[0, 5, 2, 3] = np.nan
        x[0, 7, 4, 4] = np.inf
        net.params["l"][0].data[...] = np.eye(64, dtype=np.float32).reshape(64, 64, 1, 1)
        net.blobs["x"].data[...] = x
        y = net.forward()["y"].copy()
        assert tile in net.plan_text(), net.plan_text()
        assert np.isnan(y[0, :, 2, 3]).all(), tile  # NaN * 0 = NaN: the whole pixel
        assert np.isnan(y[0, 7, 4, 4]) or np.isinf(y[0, 7, 4, 4]), tile
        ok = np.isfinite(x).all(axis=1)[0]
        assert np.isfinite(y[0][:, ok]).all(), tile
