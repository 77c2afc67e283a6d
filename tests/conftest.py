import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deepcut-cnn_amd")
for p in (ROOT, PKG, os.path.join(PKG, "python")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # test modules import `caffe` (the ctypes shim) at collection time: the native library has to exist before that.
    # A no-op when the in-tree build is up to date (e.g. on the GPU box, where the prebuilt .so travels with the snapshot).
    import __graft_entry__ as g

    g.build()


@pytest.fixture(scope="session", autouse=True)
def built():
    """Everything native is built once per session (no-op when up to date)."""
    import __graft_entry__ as g

    g.build()


@pytest.fixture(scope="session")
def synth152(tmp_path_factory):
    """(path of a synthetic ResNet-152 .caffemodel, its layer list) — seed 0."""
    from deepcut_tools import synth_weights, write_caffemodel

    d = tmp_path_factory.mktemp("weights")
    path = str(d / "synth152_seed0.caffemodel")
    layers = synth_weights(152, seed=0)
    write_caffemodel(path, "ResNet-152", layers)
    return path, layers


@pytest.fixture(scope="session")
def gpu_caffe():
    import caffe

    if caffe.device_count() < 1:
        pytest.fail("a -m gpu test ran without a HIP device; the forward path has no CPU fallback")
    caffe.set_mode_gpu()
    caffe.set_device(0)
    return caffe


def rand_image(seed, h, w, n=1):
    return (np.random.RandomState(seed).randn(n, 3, h, w) * 50).astype(np.float32)
