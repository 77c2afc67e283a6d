"""Generate tests/golden/tiling_golden.npz by running the reference's own `_process_image_tiled`
(python/pose/estimate_pose.py:159-221) in the build container (/root/reference is not on the GPU box).

Two obstacles, and how they are handled without touching the reference:

* the module imports `caffe` / `scipy.misc` at load time -> an empty `caffe` module is put in sys.modules
  (as make_pose_golden.py does); the network is replaced by `FakeNet` below, an object with the four pycaffe
  members the function touches (`blobs[...]`, `.reshape`, `.data`, `.forward()`), whose "maps" are a
  deterministic closed-form function of the input tile, so the fixture needs no weights;
* the function was written for Python 2: `cut_off = rf / stride` is an int there and a float (TypeError when
  slicing) on Python 3.  `stride` is an argument, so it is passed as `Py2Int(8)`, an int subclass whose
  true division is Python 2's integer division.  What is recorded is therefore what the reference did under
  the interpreter it was written for.

Stored: the canvas seed/shape, the reference's stitched maps, and per-case tile counts.

    python tests/golden/make_tiling_golden.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/python/pose"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiling_golden.npz")


class Py2Int(int):
    def __rtruediv__(self, other):
        return int(other) // int(self)

    def __truediv__(self, other):
        return int(self) // int(other)


def fake_maps(tile_chw):
    """A stand-in network with the real one's geometry (maps = ceil(side / 8) cells): cell (i, j) of
    channel c is a fixed function of the 8x8 pixel block under it.  Shared with tests/test_tiling.py."""
    _, h, w = tile_chw.shape
    mh, mw = -(-h // 8), -(-w // 8)
    pad = np.zeros((3, mh * 8, mw * 8), np.float64)
    pad[:, :h, :w] = tile_chw
    blocks = pad.reshape(3, mh, 8, mw, 8)
    mean = blocks.mean(axis=(2, 4))  # [3, mh, mw]
    corner = blocks[:, :, 0, :, 0]
    prob = np.empty((14, mh, mw), np.float32)
    loc = np.empty((28, mh, mw), np.float32)
    for c in range(14):
        prob[c] = 1.0 / (1.0 + np.exp(-(mean[c % 3] * (0.01 + 0.002 * c) + 0.1 * c - 0.5)))
    for c in range(28):
        loc[c] = corner[c % 3] * 0.01 + mean[(c + 1) % 3] * 0.003 * (c - 13)
    return prob, loc


class _Blob(object):
    def __init__(self):
        self.data = np.zeros((1, 3, 8, 8), np.float32)

    def reshape(self, *shape):
        self.data = np.zeros(shape, np.float32)


class FakeNet(object):
    def __init__(self):
        self.blobs = {"data": _Blob(), "prob": _Blob(), "loc_pred": _Blob()}
        self.calls = []

    def forward(self):
        x = self.blobs["data"].data[0]
        self.calls.append(x.shape[1:])
        prob, loc = fake_maps(x)
        self.blobs["prob"].data = prob[None]
        self.blobs["loc_pred"].data = loc[None]


CASES = [(120, 160), (704, 1000), (1000, 360), (760, 1304)]  # canvas H x W (multiples of 8)


def canvas(i, h, w):
    return (np.random.RandomState(500 + i).randn(h, w, 3) * 40).astype(np.float32)


def main():
    sys.modules.setdefault("caffe", types.ModuleType("caffe"))
    sys.path.insert(0, REF)
    import estimate_pose as ref

    data = {"cases": np.array(CASES)}
    for i, (h, w) in enumerate(CASES):
        net = FakeNet()
        score, off = ref._process_image_tiled(net, canvas(i, h, w), Py2Int(8))
        data["score_%d" % i] = score.astype(np.float32)  # (H', W', 14)
        data["off_%d" % i] = off.astype(np.float32)      # (H', W', 14, 2)
        data["tiles_%d" % i] = np.array(net.calls)
        print(i, (h, w), "->", score.shape, off.shape, "tiles", len(net.calls))
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
