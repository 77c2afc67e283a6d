"""Fixtures written by the INDEPENDENT torch float64 graph (tests/golden/make_torch_golden.py; neither the oracle nor the library
produced them): CPU — the oracle reproduces them; GPU — the HIP path does, to the north-star's 1e-3, without running the oracle.
Includes the ResNet-101 table and a 60x76 input (not a multiple of 8: ceil-mode pooling and the Crop decide the map size)."""
import glob
import os

import numpy as np
import pytest

FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "torch64_*.npz")))


def _case(g):
    n, _, h, w = [int(v) for v in g["shape"]]
    x = (np.random.RandomState(int(g["input_seed"])).randn(n, 3, h, w) * float(g["input_scale"])).astype(np.float32)
    return int(g["depth"]), n, h, w, x


def test_fixtures_exist():
    assert len(FIX) == 4
    for f in FIX:
        g = np.load(f)
        assert 0.0 < float(g["prob"].min()) and float(g["prob"].max()) < 1.0 and float(g["res5c_absmax"]) < 500


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(f) for f in FIX])
def test_oracle_reproduces_the_independent_graph(path):
    from deepcut_tools import deepercut_prototxt, synth_weights
    from oracle import oracle as O

    g = np.load(path)
    depth, n, h, w, x = _case(g)
    O.set_threads(min(8, os.cpu_count() or 1))
    out = O.OracleNet(deepercut_prototxt(depth, h, w, n), synth_weights(depth, int(g["weight_seed"]))).forward(data=x)
    for k in ("prob", "loc_pred", "next_pred"):
        assert out[k].shape == g[k].shape
        assert float(np.abs(out[k] - g[k]).max()) <= 2e-5 * max(1.0, float(np.abs(g[k]).max())), k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(f) for f in FIX])
def test_hip_path_matches_the_independent_graph(gpu_caffe, tmp_path, path, dtype):
    from deepcut_tools import deepercut_prototxt, synth_weights, write_caffemodel

    g = np.load(path)
    depth, n, h, w, x = _case(g)
    wpath = str(tmp_path / "w.caffemodel")
    write_caffemodel(wpath, "ResNet-%d" % depth, synth_weights(depth, int(g["weight_seed"])))
    net = gpu_caffe.Net(deepercut_prototxt(depth, h, w, n), wpath, gpu_caffe.TEST, from_text=True, dtype=dtype)
    out = net.forward_batch(x)
    for k in ("prob", "loc_pred", "next_pred"):
        assert out[k].shape == g[k].shape
        err = float(np.abs(out[k] - g[k]).max())
        if dtype == "f32":
            assert err <= 1e-3, (k, err)
        else:  # the fp16 bounds of tests/test_gpu_fp16.py
            assert err <= (2.5e-3 if k == "prob" else 4e-3 * max(1.0, float(np.abs(g[k]).max()))), (k, err)
