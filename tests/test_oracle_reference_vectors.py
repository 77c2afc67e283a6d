"""CPU: pin the oracle (oracle/caffe_cpu.c) against the known-answer vectors the reference's OWN unit
tests hold for this path, and against independent float64 implementations.  Each test
cites the reference test it restates.  The reference tests do NOT pin BatchNorm(use_global_stats),
the fork's Crop, or the DeeperCut graph — those are checked here against first-principles numpy only
("parity unpinned" for them)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O


def test_deconv_closed_form_known_answer():
    # src/caffe/test/test_deconvolution_layer.cpp:91-137 — ones input 2x3x6x4, k3 s2, 4 outputs,
    # weight 1, bias 0.1: 3.1 everywhere, +3 where one axis overlaps, +9 where both do.
    x = np.ones((2, 3, 6, 4), np.float32)
    w = np.ones((3, 4, 3, 3), np.float32)
    b = np.full(4, 0.1, np.float32)
    y = O.deconv_forward(x, w, b, stride=2)
    assert y.shape == (2, 4, 13, 9)  # :55-89 TestSetup: 6x4 -> 13x9
    H, W = y.shape[2:]
    for h in range(H):
        for ww in range(W):
            exp = 3.1
            ho = h % 2 == 0 and 0 < h < H - 1
            wo = ww % 2 == 0 and 0 < ww < W - 1
            if ho and wo:
                exp += 9
            elif ho or wo:
                exp += 3
            assert np.allclose(y[:, :, h, ww], exp, atol=1e-4), (h, ww)


def test_maxpool_square_literal():
    # src/caffe/test/test_pooling_layer.cpp:49-119 — k2 s1 on a 3x5 literal
    img = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)
    x = np.tile(img, (2, 2, 1, 1))
    y = O.maxpool_forward(x, 2, 1)
    assert y.shape == (2, 2, 2, 4)
    assert np.array_equal(y[1, 1], np.array([[9, 5, 5, 8], [9, 5, 5, 8]], np.float32))


def test_maxpool_padded_literal():
    # src/caffe/test/test_pooling_layer.cpp:478-521 — k3 s2 pad2 on a 3x3 literal
    x = np.array([[1, 2, 4], [2, 3, 2], [4, 2, 1]], np.float32).reshape(1, 1, 3, 3)
    y = O.maxpool_forward(x, 3, 2, 2)
    assert np.array_equal(y[0, 0], np.array([[1, 4, 4], [4, 4, 4], [4, 4, 1]], np.float32))


@pytest.mark.parametrize("hw", [(120, 160), (272, 368), (15, 20), (16, 16), (13, 17), (3, 3)])
def test_maxpool_ceil_mode_shapes_and_values(hw):
    # pooling_layer.cpp:90-93 ceil-mode output; the net's pool1 is k3 s2 p0
    x = np.random.RandomState(0).randn(1, 3, *hw).astype(np.float32)
    y = O.maxpool_forward(x, 3, 2)
    yt = F.max_pool2d(torch.from_numpy(x), 3, 2, ceil_mode=True).numpy()
    assert y.shape == yt.shape
    assert np.array_equal(y, yt)


def _loop_conv(x, w, b, s, p, d):
    """independent direct-loop convolution in float64 (the role of caffe_conv,
    src/caffe/test/test_convolution_layer.cpp:22-139)"""
    n, c, h, wd = x.shape
    co, _, kh, kw = w.shape
    oh = (h + 2 * p - (d * (kh - 1) + 1)) // s + 1
    ow = (wd + 2 * p - (d * (kw - 1) + 1)) // s + 1
    xp = np.zeros((n, c, h + 2 * p, wd + 2 * p))
    xp[:, :, p:p + h, p:p + wd] = x
    y = np.zeros((n, co, oh, ow))
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, :, ky * d: ky * d + s * (oh - 1) + 1: s, kx * d: kx * d + s * (ow - 1) + 1: s]
            y += np.einsum("nchw,oc->nohw", patch, w[:, :, ky, kx].astype(np.float64))
    if b is not None:
        y += b.reshape(1, -1, 1, 1)
    return y


@pytest.mark.parametrize("k,s,p,d", [(3, 2, 0, 1),  # TestSimpleConvolution :231-265
                                      (3, 1, 0, 2),  # TestDilatedConvolution :267-309
                                      (1, 1, 0, 1),  # Test1x1Convolution :443-468
                                      (3, 1, 1, 1), (3, 1, 2, 2), (1, 2, 0, 1), (7, 2, 3, 1)])
def test_conv_against_loop_reference(k, s, p, d):
    rs = np.random.RandomState(k * 100 + s * 10 + d)
    x = rs.randn(2, 3, 17, 14).astype(np.float32)
    w = rs.randn(4, 3, k, k).astype(np.float32)
    b = rs.randn(4).astype(np.float32)
    y = O.conv_forward(x, w, b, s, p, d)
    ref = _loop_conv(x, w, b, s, p, d)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 1e-4  # the reference's own tolerance


def test_sobel_separable_identity():
    # src/caffe/test/test_convolution_layer.cpp:498-589: Sobel G_x as one 3x3 stride-2 filter equals the
    # [1 2 1]^T (3x1, stride_h 2) column filter followed by the [-1 0 1] (1x3, stride_w 2) row filter.
    rs = np.random.RandomState(7)
    x = rs.randn(2, 3, 6, 4).astype(np.float32)
    gx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float32)
    w = np.tile(gx, (1, 3, 1, 1))
    full = O.conv_forward(x, w, None, 2, 0, 1)
    w1 = np.tile(np.array([1, 2, 1], np.float32).reshape(3, 1), (1, 3, 1, 1))
    mid = O.conv_forward(x, w1, None, (2, 1), 0, 1)
    w2 = np.array([-1, 0, 1], np.float32).reshape(1, 1, 1, 3)
    sep = O.conv_forward(mid, w2, None, (1, 2), 0, 1)
    assert full.shape == sep.shape
    assert np.abs(full - sep).max() <= 1e-4


def test_deconv_matches_float64_transposed_conv():
    rs = np.random.RandomState(3)
    x = rs.randn(2, 8, 5, 7).astype(np.float32)
    w = rs.randn(8, 6, 3, 3).astype(np.float32)
    b = rs.randn(6).astype(np.float32)
    y = O.deconv_forward(x, w, b, 2)
    yt = F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), 2).numpy()
    assert y.shape == (2, 6, 11, 15)
    assert np.abs(y - yt).max() <= 1e-4


def test_scale_bias_broadcast_over_channels():
    # src/caffe/test/test_scale_layer.cpp:317-342 (axis 1 broadcast with bias)
    rs = np.random.RandomState(5)
    x = rs.randn(2, 3, 4, 5).astype(np.float32)
    g = rs.randn(3).astype(np.float32)
    b = rs.randn(3).astype(np.float32)
    y = O.scale_forward(x, g, b)
    assert np.allclose(y, x * g.reshape(1, 3, 1, 1) + b.reshape(1, 3, 1, 1), atol=1e-5)
    assert np.allclose(O.scale_forward(x, g), x * g.reshape(1, 3, 1, 1), atol=1e-6)


def test_eltwise_sum_relu_sigmoid():
    rs = np.random.RandomState(6)
    a = rs.randn(2, 3, 4, 5).astype(np.float32)
    b = rs.randn(2, 3, 4, 5).astype(np.float32)
    assert np.array_equal(O.eltwise_sum(a, b), a + b)  # test_eltwise_layer.cpp:87-104
    r = O.relu_forward(a)  # test_neuron_layer.cpp:208-221
    assert (r >= 0).all() and np.array_equal(r[a > 0], a[a > 0]) and (r[a <= 0] == 0).all()
    s = O.sigmoid_forward(a)  # test_neuron_layer.cpp:321-336: 1e-4 bound and range
    assert np.abs(s - 1.0 / (1.0 + np.exp(-a.astype(np.float64)))).max() <= 1e-4
    assert (s >= 0).all() and (s <= 1).all()


@pytest.mark.parametrize("sf", [0.0, 1.0, 999.98236])
def test_batchnorm_global_stats(sf):
    # batch_norm_layer.cpp:86-93,138-149 (unpinned by the reference's tests, which only cover batch
    # statistics): y = (x - b0*f) / sqrt(b1*f + eps), f = (b2 == 0 ? 0 : 1/b2)
    rs = np.random.RandomState(8)
    x = rs.randn(2, 5, 3, 4).astype(np.float32)
    m = (rs.randn(5) * max(sf, 1)).astype(np.float32)
    v = ((1 + rs.rand(5)) * max(sf, 1)).astype(np.float32)
    y = O.batchnorm_forward(x, m, v, np.float32(sf), 1e-5)
    f = 0.0 if sf == 0 else 1.0 / np.float32(sf)
    ref = (x.astype(np.float64) - (m * f).reshape(1, 5, 1, 1)) / np.sqrt((v * f).reshape(1, 5, 1, 1).astype(np.float64) + 1e-5)
    assert np.abs(y - ref).max() <= (2e-3 if sf == 0 else 1e-5)  # sf==0 divides by sqrt(1e-5): large values


def test_crop_top_left_and_strictness():
    # crop_layer.cpp:25-50 (fork-specific; no reference test)
    x = np.arange(2 * 3 * 5 * 7, dtype=np.float32).reshape(2, 3, 5, 7)
    ref = np.zeros((2, 3, 4, 6), np.float32)
    assert np.array_equal(O.crop_forward(x, ref), x[:, :, :4, :6])
    assert np.array_equal(O.crop_forward(x, np.zeros((2, 3, 3, 4)), 1, 2), x[:, :, 1:4, 2:6])
    with pytest.raises(AssertionError):
        O.crop_forward(x, np.zeros((2, 3, 5, 6)))  # CHECK_GT: equal height is rejected


def test_sgemm_tiny_literal():
    # src/caffe/test/test_util_blas.cpp:20-89: A(2x3) B(3x4) literal
    A = np.arange(1, 7, dtype=np.float32).reshape(2, 3)
    B = np.arange(1, 13, dtype=np.float32).reshape(3, 4)
    C = np.zeros((2, 4), np.float32)
    import ctypes as ct
    fp = ct.POINTER(ct.c_float)
    O.lib().oracle_sgemm(0, 2, 4, 3, A.ctypes.data_as(fp), B.ctypes.data_as(fp), 0.0, C.ctypes.data_as(fp))
    assert np.array_equal(C, np.array([[38, 44, 50, 56], [83, 98, 113, 128]], np.float32))
    At = np.ascontiguousarray(A.T)
    C2 = np.zeros((2, 4), np.float32)
    O.lib().oracle_sgemm(1, 2, 4, 3, At.ctypes.data_as(fp), B.ctypes.data_as(fp), 0.0, C2.ctypes.data_as(fp))
    assert np.array_equal(C2, C)


def test_float64_accumulation_mode_bounds_float32_noise():
    rs = np.random.RandomState(9)
    x = rs.randn(1, 256, 9, 11).astype(np.float32)
    w = (rs.randn(64, 256, 3, 3) * 0.05).astype(np.float32)
    y32 = O.conv_forward(x, w, None, 1, 1, 1)
    O.set_double_acc(True)
    try:
        y64 = O.conv_forward(x, w, None, 1, 1, 1)
    finally:
        O.set_double_acc(False)
    assert 0 < np.abs(y32 - y64).max() < 1e-4


_MAGIC6 = np.array([[35, 1, 6, 26, 19, 24], [3, 32, 7, 21, 23, 25], [31, 9, 2, 22, 27, 20],
                    [8, 28, 33, 17, 10, 15], [30, 5, 34, 12, 14, 16], [4, 36, 29, 13, 18, 11]], np.float32)


def test_maxpool_rectangular_literals():
    # src/caffe/test/test_pooling_layer.cpp:121-245 (TestForwardRectHigh: kernel_h 3, kernel_w 2, stride 1 on magic(6)) and
    # :246-372 (TestForwardRectWide: kernel_h 2, kernel_w 3) — the literals the reference holds for its pooling loop
    x = np.tile(_MAGIC6, (2, 2, 1, 1))
    high = O.maxpool_forward(x, (3, 2), 1)
    assert high.shape == (2, 2, 4, 5)
    want_high = np.array([[35, 32, 26, 27, 27], [32, 33, 33, 27, 27], [31, 34, 34, 27, 27], [36, 36, 34, 18, 18]], np.float32)
    wide = O.maxpool_forward(x, (2, 3), 1)
    assert wide.shape == (2, 2, 5, 4)
    want_wide = np.array([[35, 32, 26, 26], [32, 32, 27, 27], [33, 33, 33, 27], [34, 34, 34, 17], [36, 36, 34, 18]], np.float32)
    for n in range(2):
        for c in range(2):
            assert np.array_equal(high[n, c], want_high)
            assert np.array_equal(wide[n, c], want_wide)


def test_bias_broadcast_middle_and_channel():
    # src/caffe/test/test_bias_layer.cpp:199-220 (TestForwardBroadcastMiddle): bottom 2x3x4x5 and a 3x4 bias at axis 1, both
    # uniform in [1, 10]: top(n,c,h,w) = bottom(n,c,h,w) + bias(c,h) within 1e-5.  BiasLayer flattens that to
    # outer 2 x bias_dim 12 x inner 5 (bias_layer.cpp:30-46,72-87) — the same loop the net's Scale(bias_term) layers run
    # with bias_dim = C, inner = H*W, which is the second half of this test.
    rs = np.random.RandomState(1701)
    x = rs.uniform(1, 10, (2, 3, 4, 5)).astype(np.float32)
    bias = rs.uniform(1, 10, (3, 4)).astype(np.float32)
    y = O.scale_forward(x.reshape(2, 12, 1, 5), np.ones(12, np.float32), bias.reshape(12)).reshape(2, 3, 4, 5)
    for n in range(2):
        for c in range(3):
            for h in range(4):
                for w in range(5):
                    assert abs(float(y[n, c, h, w]) - (float(x[n, c, h, w]) + float(bias[c, h]))) <= 1e-5
    bc = rs.uniform(1, 10, 3).astype(np.float32)
    y = O.scale_forward(x, np.ones(3, np.float32), bc)
    assert np.abs(y - (x + bc.reshape(1, 3, 1, 1))).max() <= 1e-5
