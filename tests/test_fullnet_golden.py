"""Golden full-net fixtures (tests/golden/fullnet_*.npz, written by make_fullnet_golden.py from the CPU
oracle): CPU — the oracle and the seeded weight generator still reproduce them; GPU — the HIP path matches
them to 1e-3 without running the oracle."""
import glob
import os

import numpy as np
import pytest

FIX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fullnet_*.npz")))


def _input(g):
    return (np.random.RandomState(int(g["input_seed"])).randn(*g["shape"]) * float(g["input_scale"])).astype(np.float32)


def test_fixtures_exist_and_are_conditioned():
    assert len(FIX) == 3
    for f in FIX:
        g = np.load(f)
        assert 0.5 < float(np.abs(g["loc_pred"]).max()) < 50 and float(g["res5c_absmax"]) < 500
        assert 0.0 < float(g["prob"].min()) and float(g["prob"].max()) < 1.0


def test_oracle_reproduces_smallest_fixture():
    from deepcut_tools import deepercut_prototxt, synth_weights
    from oracle import oracle as O

    g = np.load(FIX[-1] if "64x64" in FIX[-1] else [f for f in FIX if "64x64" in f][0])
    n, _, h, w = g["shape"]
    O.set_threads(os.cpu_count() or 1)
    out = O.OracleNet(deepercut_prototxt(152, int(h), int(w), int(n)), synth_weights(152, int(g["weight_seed"]))).forward(data=_input(g))
    for k in ("prob", "loc_pred", "next_pred"):
        assert np.abs(out[k] - g[k]).max() <= 2e-5, k  # thread-count dependent GEMM blocking only


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(f) for f in FIX])
def test_hip_path_matches_golden(gpu_caffe, synth152, path):
    from deepcut_tools import deepercut_prototxt

    g = np.load(path)
    n, _, h, w = [int(v) for v in g["shape"]]
    wpath, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w, n), wpath, gpu_caffe.TEST, from_text=True)
    net.blobs["data"].data[...] = _input(g)
    out = net.forward()
    for k in ("prob", "loc_pred", "next_pred"):
        assert out[k].shape == g[k].shape
        assert float(np.abs(out[k] - g[k]).max()) <= 1e-3, k
