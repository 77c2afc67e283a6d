"""CPU: the pose pre/post-processing of this package against golden vectors produced by the reference's
own python/pose/estimate_pose.py (tests/golden/make_pose_golden.py), plus first-principles checks of the
pre-processing, whose resize step the reference delegates to scipy.misc.imresize (absent from every
current SciPy; identity at scale 1 — other scales are pinned to PIL here, "parity unpinned")."""
import os

import numpy as np
import pytest

from pose import estimate_pose as ep

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_golden.npz"))


@pytest.mark.parametrize("i", range(5))
def test_pose_from_maps_matches_reference(i):
    pose = ep.pose_from_maps(G["prob_%d" % i], G["loc_%d" % i], float(G["scale_%d" % i]))
    ref = G["pose_%d" % i]
    assert pose.shape == ref.shape == (5, 14)
    assert np.allclose(pose, ref, rtol=0, atol=1e-9)


def test_constants_and_tile_rule_match_reference():
    assert np.array_equal(ep.MEAN_BGR, G["mean"])
    assert ep.LOCREF_SCALE == float(G["locref_scale"]) and ep.STRIDE == float(G["stride"])
    got = [ep.num_tiles(int(l)) for l in G["tile_lengths"]]
    assert got == list(G["tile_counts"])


def test_preprocess_scale1_is_mean_subtraction_on_a_stride_canvas():
    img = np.random.RandomState(0).randint(0, 256, (240, 320, 3)).astype(np.uint8)  # BASELINE configs[0] input
    x = ep.preprocess(img, 1.0)
    assert x.shape == (240, 320, 3) and x.dtype == np.float32
    assert np.array_equal(x, img.astype(np.float32) - ep.MEAN_BGR.astype(np.float32))
    img2 = img[:235, :317]
    x2 = ep.preprocess(img2, 1.0)
    assert x2.shape == (240, 320, 3)
    # the 64-px replicate pad fills the canvas up to the stride: last row / column repeated, not zeros
    assert np.array_equal(x2[:235, :317], img2.astype(np.float32) - ep.MEAN_BGR.astype(np.float32))
    assert np.array_equal(x2[236, :317], x2[234, :317]) and np.array_equal(x2[:235, 319], x2[:235, 316])


@pytest.mark.parametrize("scale,shape", [(0.5, (272, 368)), (0.75, (408, 552)), (1.25, (680, 920))])
def test_preprocess_scaled_shapes_follow_the_stride_rule(scale, shape):
    # 736x544 at scales 0.5/0.75/1.25 -> 272x368, 408x552, 680x920
    img = np.random.RandomState(1).randint(0, 256, (544, 736, 3)).astype(np.uint8)
    assert ep.preprocess(img, scale).shape == shape + (3,)


def test_select_best_uses_strict_min_confidence():
    a = np.zeros((5, 14)); a[2] = 0.3
    b = np.zeros((5, 14)); b[2] = 0.9; b[2, 3] = 0.2
    c = np.zeros((5, 14)); c[2] = 0.3
    assert ep.select_best([a, b, c]) is a        # b has the higher mean but the lower minimum; c ties -> first kept
    assert ep.select_best([np.zeros((5, 14))]) is None
