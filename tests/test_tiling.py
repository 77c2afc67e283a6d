"""Tiling of > 700-px inputs (reference python/pose/estimate_pose.py:146-221,245-259).

* "reference" mode against tests/golden/tiling_golden.npz, which is the output of the reference's own
  `_process_image_tiled` driven with a closed-form stand-in network (tests/golden/make_tiling_golden.py);
* "exact" mode: the kept cell ranges partition the map, and with a network whose receptive field fits in the
  224-px margin the stitched maps equal the un-tiled maps."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_tiling_golden import CASES, canvas, fake_maps  # noqa: E402  (generator helpers; no reference import)

from pose import estimate_pose as ep  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "tiling_golden.npz"))


def _fake_forward(calls):
    def fwd(_net, tile_hwc):
        calls.append(tile_hwc.shape[:2])
        return fake_maps(np.ascontiguousarray(tile_hwc.transpose(2, 0, 1)))

    return fwd


@pytest.mark.parametrize("i", range(len(CASES)))
def test_reference_mode_reproduces_the_reference_stitching(i):
    h, w = CASES[i]
    calls = []
    prob, loc = ep.forward_maps_tiled(None, canvas(i, h, w), mode="reference", forward=_fake_forward(calls))
    score, off = G["score_%d" % i], G["off_%d" % i]  # (H', W', 14), (H', W', 14, 2)
    assert np.array_equal(np.array(calls), G["tiles_%d" % i])
    assert prob.shape == (14,) + score.shape[:2]
    assert np.array_equal(prob.transpose(1, 2, 0), score)
    hh, ww = score.shape[:2]
    assert np.array_equal(loc.reshape(14, 2, hh, ww).transpose(2, 3, 0, 1), off)


@pytest.mark.parametrize("length", [8, 696, 704, 928, 936, 1168, 1176, 1304, 2000, 4000])
def test_exact_spans_partition_the_map(length):
    spans = ep.tile_spans(length)
    covered = []
    for start, end, lo, hi in spans:
        assert start % 16 == 0 and 0 < end - start <= 700 and end <= length and lo < hi
        covered += list(range(start // 8 + lo, start // 8 + hi))
    assert covered == list(range(length // 8))
    assert spans[-1][1] == length
    if length <= 700:
        assert len(spans) == 1
    for (s, e, lo, hi), nxt in zip(spans, spans[1:]):
        assert e - s == 688 and nxt[0] - s == 240 and hi == 86 - 28 and nxt[2] == 28


def test_exact_spans_refuse_bad_geometry():
    with pytest.raises(ValueError):
        ep.tile_spans(1001)
    with pytest.raises(ValueError):
        ep.tile_spans(2000, rf=100)
    with pytest.raises(ValueError):
        ep.tile_spans(2000, max_size=440)


@pytest.mark.parametrize("hw", [(120, 160), (704, 1000), (1000, 360), (1304, 1416)])
def test_exact_mode_equals_the_untiled_maps_for_a_local_network(hw):
    # fake_maps looks only at the 8x8 block under a cell: any correct stitcher must reproduce the whole map.
    x = canvas(7, *hw)
    calls = []
    prob, loc = ep.forward_maps_tiled(None, x, mode="exact", forward=_fake_forward(calls))
    ref_prob, ref_loc = fake_maps(x.transpose(2, 0, 1))
    assert np.array_equal(prob, ref_prob) and np.array_equal(loc, ref_loc)
    assert len(calls) == len(ep.tile_spans(hw[0])) * len(ep.tile_spans(hw[1]))


def test_unknown_mode_is_refused():
    with pytest.raises(ValueError):
        ep.forward_maps_tiled(None, canvas(0, 16, 16), mode="matlab", forward=_fake_forward([]))
