"""CPU: the host half of the float16 Winograd form (csrc/wino_f16.hip, tile wino_h23) — the filter image the lowering packs.
Against first principles in NumPy float64: U = G g G^T per (output channel, input channel), the per-channel power-of-two row scale,
the MFMA fragment order the kernel reads it in, and the identity the kernel's arithmetic rests on: A^T [U . (B^T d B)] A is the 3x3
cross-correlation of the reference's Convolution (im2col + SGEMM, base_conv_layer.cpp:257-280, im2col.cpp:19-55) on a 4x4 patch."""
import numpy as np
import pytest

import caffe

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def _unpack(img, cout, cin):
    """[cout/32][4 i][cin/16][4 j][64 lanes][8] -> U[co, ci, i, j]: lane = 32 * ((ci % 16) / 8) + co % 32, element = ci % 8"""
    u = np.zeros((cout, cin, 4, 4))
    for co in range(cout):
        for ci in range(cin):
            u[co, ci] = img[co // 32, :, ci // 16, :, 32 * ((ci % 16) // 8) + co % 32, ci % 8]
    return u


@pytest.mark.parametrize("cout,cin", [(64, 64), (128, 32), (32, 48)])
def test_image_is_G_g_Gt_in_fragment_order_with_a_power_of_two_row_scale(cout, cin):
    rs = np.random.RandomState(cout + cin)
    g = (rs.randn(cout, cin, 3, 3) / np.sqrt(9.0 * cin)).astype(np.float32)
    g[3] *= 1e-5   # a head-like channel: tiny filters must not end up in float16's subnormals
    g[5] = 0.0     # an all-zero channel keeps scale 1
    img, scale = caffe.wino_half_pack(g, rowscale=True)
    plain, ones = caffe.wino_half_pack(g, rowscale=False)
    assert np.array_equal(ones, np.ones(cout, np.float32))
    want = np.einsum("ia,ocab,jb->ocij", G, g.astype(np.float64), G)
    assert np.allclose(_unpack(plain, cout, cin), want, rtol=0, atol=1e-7 * np.abs(want).max())
    u = _unpack(img, cout, cin)
    for co in range(cout):
        m, e = np.frexp(float(scale[co]))
        assert m == 0.5, "row scales are exact powers of two"
        assert np.array_equal(u[co] * float(scale[co]), _unpack(plain, cout, cin)[co])  # scaling by a power of two is exact in float32
        peak = np.abs(u[co]).max()
        assert (peak == 0 and scale[co] == 1) or 2.0 ** 13 <= peak < 2.0 ** 14, (co, peak)
    assert np.isfinite(img.astype(np.float16)).all() and (np.abs(img.astype(np.float16)[img != 0]) >= 6.2e-5).mean() > 0.99  # normal float16s


def test_the_transform_identity_the_kernel_computes():
    """Y = A^T [ sum_ci U[co, ci] . (B^T d[ci] B) ] A equals the reference's 3x3 convolution (pad-free, stride 1) on the 4x4 patch;
    with the kernel's extras: patch pre-multiplied by 1/4, row-scaled image, epilogue scale = 4 x row scale"""
    rs = np.random.RandomState(7)
    cout, cin = 32, 16
    g = (rs.randn(cout, cin, 3, 3) / 12.0).astype(np.float32)
    d = rs.randn(cin, 4, 4)
    img, scale = caffe.wino_half_pack(g, rowscale=True)
    u = _unpack(img, cout, cin)
    v = np.einsum("ia,cab,jb->cij", BT, 0.25 * d, BT)           # B^T (d / 4) B per input channel
    m = np.einsum("ocij,cij->oij", u, v)                         # the 16 GEMMs over the input channels
    y = np.einsum("ai,oij,bj->oab", AT, m, AT) * (4.0 * scale.astype(np.float64))[:, None, None]
    ref = np.zeros((cout, 2, 2))
    for a in range(2):
        for b in range(2):
            ref[:, a, b] = np.einsum("ocij,cij->o", g.astype(np.float64), d[:, a:a + 3, b:b + 3])  # cross-correlation, as im2col + GEMM
    assert np.allclose(y, ref, rtol=0, atol=1e-6 * np.abs(ref).max())


def test_bad_sizes_are_refused():
    with pytest.raises(caffe.DeepcutError):
        caffe.wino_half_pack(np.zeros((48, 32, 3, 3), np.float32))
    with pytest.raises(caffe.DeepcutError):
        caffe.wino_half_pack(np.zeros((32, 24, 3, 3), np.float32))


def test_the_kernels_float16_arithmetic_emulated_in_numpy_stays_inside_the_layer_bound():
    """No GPU: the arithmetic of wino_h23 restated in NumPy with float16 where the kernel has float16 — activations rounded to half,
    pre-multiplied by 1/4 (exact), B^T d B by packed float16 adds (two roundings), the row-scaled filter image rounded to half,
    products accumulated wider, inverse transform and the x 4 x row-scale epilogue in float32 — against the CPU oracle's float32
    convolution (oracle/caffe_cpu.c: im2col + SGEMM, base_conv_layer.cpp:257-280).  Bound: the float16 path's single-layer 2e-3 x range
    (tests/test_gpu_fp16.py); what the GPU kernel measures is 4-7e-4 x range."""
    from oracle import oracle as O

    rs = np.random.RandomState(3)
    cin, cout, h, w = 64, 64, 12, 10
    x = rs.randn(1, cin, h, w).astype(np.float32)
    g = (rs.randn(cout, cin, 3, 3) / np.sqrt(9.0 * cin)).astype(np.float32)
    ref = O.conv_forward(x, g, None, 1, 1, 1)[0]
    img, scale = caffe.wino_half_pack(g, rowscale=True)
    u16 = _unpack(img, cout, cin).astype(np.float16)                      # [co, ci, i, j], as uploaded
    xp = np.zeros((cin, h + 2 + 2, w + 2 + 2), np.float16)               # zero padding 1 (+ slack for the last tile)
    xp[:, 1:h + 1, 1:w + 1] = x[0].astype(np.float16) * np.float16(0.25)  # staged pixels: rounded to half, x 1/4 (exact)
    out = np.zeros((cout, h, w), np.float32)
    for ty in range((h + 1) // 2):
        for tx in range((w + 1) // 2):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]              # [ci, 4, 4] float16
            t = np.stack([d[:, 0] - d[:, 2], d[:, 1] + d[:, 2], d[:, 2] - d[:, 1], d[:, 1] - d[:, 3]], 1)      # B^T rows: float16 adds
            v = np.stack([t[:, :, 0] - t[:, :, 2], t[:, :, 1] + t[:, :, 2], t[:, :, 2] - t[:, :, 1], t[:, :, 1] - t[:, :, 3]], 2)
            assert t.dtype == np.float16 and v.dtype == np.float16
            m = np.einsum("ocij,cij->oij", u16.astype(np.float64), v.astype(np.float64)).astype(np.float32)   # MFMA: exact products, wide sums
            y = np.einsum("ai,oij,bj->oab", AT.astype(np.float32), m, AT.astype(np.float32)) * (np.float32(4.0) * scale)[:, None, None]
            out[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = y[:, :min(2, h - 2 * ty), :min(2, w - 2 * tx)]
    out = out.astype(np.float16).astype(np.float32)                       # the stored activation
    err, rng = float(np.abs(out - ref).max()), float(np.abs(ref).max())
    assert 1e-5 * rng < err <= 2e-3 * max(1.0, rng), (err, rng)
    # and the 1/4 is what keeps a trunk at 70 % of float16's range finite: without it the transformed patch overflows
    big = np.full((1, 4, 4), 4.6e4, np.float16)
    big[0, 1::2, :] *= np.float16(-1)  # alternating rows: B^T rows 1..3 subtract, so magnitudes add
    with np.errstate(over="ignore"):
        assert np.isinf((big[:, 2] - big[:, 1]).astype(np.float16)).any()
    q = big * np.float16(0.25)
    tq = q[:, 2] - q[:, 1]
    assert np.isfinite(tq).all() and np.isfinite(tq[:, 2:3] - tq[:, 1:2]).all()
