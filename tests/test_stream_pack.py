"""CPU: the host half of the streaming kernels of round 6 (the two float16 ones, and the float32 form of the 1x1 stream at the end) — the filter images the lowering packs for the streaming 1x1 form
(csrc/stream1x1.hip) and the stem kernel (csrc/stem_f16.hip) — against first principles in NumPy: the operand layout of
v_mfma_f32_32x32x16_f16 (row operand: lane l supplies row l % 32, K elements 8 (l / 32) .. + 7; column operand alike) emulated on the
images reproduces the reference's convolutions (1x1: one GEMM per image, base_conv_layer.cpp:326-341; the 7x7 / 2 stem: im2col + GEMM,
im2col.cpp:19-55 with pad 3)."""
import numpy as np
import pytest

import caffe


def _mfma(a_lanes, b_lanes):
    """D[row, col] += sum_k A[row, k] B[k, col] for one 32x32x16 step: a_lanes / b_lanes [64, 8] = what every lane holds"""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for lane in range(64):
        A[lane % 32, 8 * (lane // 32):8 * (lane // 32) + 8] = a_lanes[lane]
        B[8 * (lane // 32):8 * (lane // 32) + 8, lane % 32] = b_lanes[lane]
    return A @ B


@pytest.mark.parametrize("cout,k", [(64, 64), (256, 128), (96, 48)])
def test_stream1x1_image_is_the_row_operand_of_the_matrix_instruction(cout, k):
    rs = np.random.RandomState(cout + k)
    g = rs.randn(cout, k).astype(np.float32)
    img = caffe.stream1x1_pack(g)
    assert img.shape == (cout // 32, k // 16, 64, 8)
    assert sorted(img.ravel().tolist()) == sorted(g.ravel().tolist())  # a permutation of the filters
    x = rs.randn(32, k).astype(np.float32)  # 32 pixels
    want = g.astype(np.float64) @ x.astype(np.float64).T  # [cout, pixel]
    for f in range(cout // 32):
        d = np.zeros((32, 32))
        for kk in range(k // 16):
            b_lanes = np.stack([x[lane % 32, kk * 16 + 8 * (lane // 32):kk * 16 + 8 * (lane // 32) + 8] for lane in range(64)])
            d += _mfma(img[f, kk], b_lanes)
        assert np.allclose(d, want[f * 32:(f + 1) * 32], rtol=0, atol=1e-5 * np.abs(want).max())


def _mfma16(a_lanes, b_lanes):
    """one v_mfma_f32_16x16x4_f32 step: lane 16 q + r holds A[r, q] (row operand) and B[q, r] (column operand); D = A B [16, 16]"""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for lane in range(64):
        A[lane % 16, lane // 16] = a_lanes[lane]
        B[lane // 16, lane % 16] = b_lanes[lane]
    return A @ B


@pytest.mark.parametrize("cout,k", [(64, 256), (32, 512), (64, 128), (48, 64)])
def test_stream1x1f_image_is_the_row_operand_of_the_float32_matrix_instruction(cout, k):
    """The float32 form (csrc/stream1x1_f32.hip): the K range in four runs, lane 16 q + c on run q; a lane's 16-byte read of its pixel's row
    at run q, vector j feeds the four matrix steps of filter vector j."""
    rs = np.random.RandomState(cout + k)
    g = rs.randn(cout, k).astype(np.float32)
    img = caffe.stream1x1f_pack(g)
    assert img.shape == (cout // 16, k // 16, 64, 4)
    assert sorted(img.ravel().tolist()) == sorted(g.ravel().tolist())  # a permutation of the filters
    x = rs.randn(16, k).astype(np.float32)  # 16 pixels
    want = g.astype(np.float64) @ x.astype(np.float64).T  # [cout, pixel]
    for f in range(cout // 16):
        d = np.zeros((16, 16))
        for j in range(k // 16):
            xv = np.stack([x[lane % 16, (lane // 16) * (k // 4) + 4 * j:(lane // 16) * (k // 4) + 4 * j + 4] for lane in range(64)])  # the LDS read
            for e in range(4):
                d += _mfma16(img[f, j, :, e], xv[:, e])
        assert np.allclose(d, want[f * 16:(f + 1) * 16], rtol=0, atol=1e-5 * np.abs(want).max())
    with pytest.raises(caffe.DeepcutError):
        caffe.stream1x1f_pack(np.zeros((24, 64), np.float32))


@pytest.mark.parametrize("c", [3, 4, 1])
def test_stem_image_reproduces_the_7x7_stride_2_convolution(c):
    rs = np.random.RandomState(c)
    g = rs.randn(64, c, 7, 7).astype(np.float32)
    img = caffe.stem7x7_pack(g)
    assert img.shape == (2, 7, 2, 64, 8)
    assert np.count_nonzero(img) == np.count_nonzero(g)  # every filter once, zeros in the padding channel(s) and the eighth tap
    h, w = 9, 70
    x = rs.randn(c, h, w).astype(np.float32)
    oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    xp = np.zeros((4, h + 6, w + 8))  # pad 3 (+ the pixel behind the eighth tap), 4 channels per pixel
    xp[:c, 3:3 + h, 3:3 + w] = x
    want = np.zeros((64, oh, ow))
    for ky in range(7):
        for kx in range(7):
            want += np.einsum("oc,chw->ohw", g[:, :, ky, kx].astype(np.float64), xp[:c, ky:ky + 2 * oh:2, kx:kx + 2 * ow:2])
    orow = 2
    cols = min(32, ow)
    for f in range(2):
        d = np.zeros((32, 32))
        for ky in range(7):
            for s in range(2):
                # lane (pixel p, K half hh): the two adjacent 4-channel pixels 2 p + 4 s + 2 hh, + 1 of padded image row 2 orow + ky
                b_lanes = np.zeros((64, 8))
                for lane in range(64):
                    p, hh = lane % 32, lane // 32
                    if p >= cols:
                        continue
                    px = 2 * p + 4 * s + 2 * hh
                    b_lanes[lane] = xp[:, 2 * orow + ky, px:px + 2].T.ravel()
                d += _mfma(img[f, ky, s], b_lanes)
        assert np.allclose(d[:, :cols], want[f * 32:(f + 1) * 32, orow, :cols], rtol=0, atol=1e-5 * np.abs(want).max())


def test_pack_entries_refuse_what_the_kernels_cannot_take():
    with pytest.raises(Exception):
        caffe.stream1x1_pack(np.zeros((48, 64), np.float32))   # not whole 32-channel fragments
    with pytest.raises(Exception):
        caffe.stream1x1_pack(np.zeros((64, 24), np.float32))   # not whole K steps
    with pytest.raises(Exception):
        caffe.stem7x7_pack(np.zeros((64, 5, 7, 7), np.float32))
