"""-m gpu: seeded random shapes through the forward — heights and widths that are NOT multiples of 8 (or of anything), batches
of 1-3, alone and as members of groups with 1 / 2 / automatic lanes, with measured tiles and with the cost model's — against the
CPU oracle.  The reference takes any input size (`Layer::Forward` re-derives every shape, layer.hpp:451-456; floor / ceil output
sizes conv_layer.cpp:8-22, pooling_layer.cpp:79-123, deconv_layer.cpp:8-22; Crop wants the deconvolved map strictly larger,
crop_layer.cpp:25-50); the tile edges, the tap masks and the XCD maps of every launch move with every pixel of the shape."""
import os

import numpy as np
import pytest

from conftest import rand_image
from oracle import oracle as O

pytestmark = pytest.mark.gpu


OFFSET = int(os.environ.get("DC_FUZZ_OFFSET", "0"))  # other shape draws (a longer hunt: for o in 1 2 3; do DC_FUZZ_OFFSET=$o pytest ...)


def _shapes(seed, count):
    rs = np.random.RandomState(seed + 10007 * OFFSET)
    out = []
    for _ in range(count):
        n = int(rs.randint(1, 4))
        h, w = int(rs.randint(16, 150)), int(rs.randint(16, 150))
        out.append((n, h, w))
    return out


def _oracle(layers, n, h, w, img):
    from deepcut_tools import deepercut_prototxt

    O.set_threads(min(16, os.cpu_count() or 1))
    return O.OracleNet(deepercut_prototxt(152, h, w, n), layers).forward(data=img)


def _check(out, ref, tol, what):
    for k in ("prob", "loc_pred", "next_pred"):
        assert out[k].shape == ref[k].shape, (what, k, out[k].shape, ref[k].shape)
        lim = tol * max(1.0, float(np.abs(ref[k]).max())) if k != "prob" else tol
        err = float(np.abs(out[k] - ref[k]).max())
        assert err <= lim, (what, k, err)


@pytest.mark.parametrize("autotune", ["1", "0"])
def test_random_shapes_one_executor(gpu_caffe, synth152, monkeypatch, autotune):
    """One executor walks through ten random shapes (plans, buffers, graphs of all of them stay cached), then back through
    the first three."""
    from deepcut_tools import deepercut_prototxt

    monkeypatch.setenv("DC_AUTOTUNE", autotune)
    path, layers = synth152
    shapes = _shapes(100 + int(autotune), 10)
    n, h, w = shapes[0]
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w, n), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    refs = []
    for i, (n, h, w) in enumerate(shapes):
        img = rand_image(500 + i, h, w, n=n)
        ref = _oracle(layers, n, h, w, img)
        refs.append((img, ref))
        _check(net.forward_batch(img), ref, 1e-3, shapes[i])
    for i in range(3):
        _check(net.forward_batch(refs[i][0]), refs[i][1], 1e-3, ("again", shapes[i]))


@pytest.mark.parametrize("dtype,lanes,members", [("f32", 1, 3), ("f32", None, 4), ("f32", 2, 5), ("f16", None, 4), ("f16", 1, 2)])
def test_random_shapes_grouped(gpu_caffe, synth152, dtype, lanes, members):
    """Groups of random-shape members (merged multi-problem launches over tensors whose tile counts, row lengths and tap masks all
    differ), two different shape tuples per group."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    seed = 7 * members + (0 if lanes is None else lanes) + (50 if dtype == "f16" else 0)
    tol = 1e-3 if dtype == "f32" else 4e-3
    first = _shapes(seed, members)
    n, h, w = first[0]
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w, n), path, gpu_caffe.TEST, from_text=True, hipgraph=1, dtype=dtype)
    grp = gpu_caffe.NetGroup.for_shapes(net, first, lanes=lanes)
    for rnd, shapes in enumerate((first, _shapes(seed + 1000, members))):
        imgs = [rand_image(900 + 10 * rnd + i, h, w, n=n) for i, (n, h, w) in enumerate(shapes)]
        outs = grp.forward_batch(imgs)
        for (n, h, w), img, out in zip(shapes, imgs, outs):
            _check(out, _oracle(layers, n, h, w, img), tol, (rnd, n, h, w))
    st = grp.stats()
    assert st["merges"] == 2 and (st["multi_launches"] > 100) == (lanes == 1 or members >= 3), st
