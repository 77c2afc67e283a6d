"""Image pre-processing (reference python/pose/estimate_pose.py:83-103).

The arithmetic that matters is Pillow's 8-bit bilinear resample (scipy.misc.imresize).  Pinned three ways:
the oracle restatement (oracle/preprocess.py) against tests/golden/preprocess_golden.npz (real Pillow, made in the
build container), against Pillow live on random sizes, and the package's host-side `preprocess` against the oracle.
The canvas geometry exported through the C ABI is checked here too (no GPU needed)."""
import os

import numpy as np
import pytest

from oracle import preprocess as OP

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "preprocess_golden.npz"))
N = int(G["n"])


@pytest.mark.parametrize("i", range(N))
def test_oracle_matches_pillow_goldens(i):
    got = OP.preprocess(G["image_%d" % i], float(G["scale_%d" % i]))
    want = G["canvas_%d" % i]
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got, want)


def test_oracle_resample_is_bit_exact_against_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(5)
    for t in range(120):
        h, w = int(rs.randint(1, 70)), int(rs.randint(1, 70))
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        if t % 4 == 0:
            img = (rs.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)
        oh, ow = int(rs.randint(1, 150)), int(rs.randint(1, 150))
        if t % 5 == 0:
            oh = h
        if t % 7 == 0:
            ow = w
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(OP.resize_bilinear_u8(img, (ow, oh)), want), (h, w, oh, ow)


def test_coefficients_are_normalised_fixed_point():
    for n_in, n_out in [(124, 62), (124, 155), (300, 17), (5, 64), (64, 64)]:
        bounds, coeffs = OP.bilinear_coeffs(n_in, n_out)
        assert coeffs.shape[0] == n_out and (coeffs >= 0).all()
        assert np.abs(coeffs.sum(axis=1) - (1 << 22)).max() <= coeffs.shape[1]  # each weight rounds by < 1 ulp
        assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= n_in).all()
        for xx in range(n_out):
            assert (coeffs[xx, bounds[xx, 1]:] == 0).all()


@pytest.mark.parametrize("i", range(N))
def test_package_host_preprocess_matches_goldens(i):
    pytest.importorskip("PIL.Image")
    from pose.estimate_pose import preprocess

    got = preprocess(G["image_%d" % i], float(G["scale_%d" % i]))
    assert np.array_equal(got, G["canvas_%d" % i])


@pytest.mark.parametrize("hw,scale", [((240, 320), 1.0), ((544, 736), 0.5), ((544, 736), 0.75), ((544, 736), 1.25),
                                      ((336, 256), 0.5), ((61, 83), 0.33), ((30, 200), 0.05), ((17, 9), 2.5)])
def test_c_abi_canvas_size_is_the_reference_rule(hw, scale):
    import caffe

    ch, cw = caffe.canvas_size(hw[0], hw[1], scale)
    assert ch == int(np.ceil(float(hw[0]) * scale / 8) * 8) and cw == int(np.ceil(float(hw[1]) * scale / 8) * 8)
    assert (ch, cw) == OP.preprocess(np.zeros(hw + (3,), np.uint8), scale).shape[:2]


def test_c_abi_canvas_size_refuses_bad_arguments():
    import caffe

    for args in [(0, 10, 1.0), (10, -1, 1.0), (10, 10, 0.0), (10, 10, -2.0)]:
        with pytest.raises(caffe.DeepcutError):
            caffe.canvas_size(*args)
