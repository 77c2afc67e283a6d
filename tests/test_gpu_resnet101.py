"""-m gpu: the ResNet-101 variant of the DeeperCut net — the depth BASELINE.json's metric and configs name — on the HIP path
against the CPU oracle.  The reference ships only models/deepercut/ResNet-152.prototxt; `deepercut_prototxt(101, ...)`
is the same generator with the (3, 4, 23, 3) block counts (names res3b1..b3 / res4b1..b22), and the product is data driven, so the
same lowering, kernels and tolerances apply: fp32 1e-3 max-abs on all three maps, fp16 at the bounds of tests/test_gpu_fp16.py."""
import os

import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def synth101(tmp_path_factory):
    from deepcut_tools import synth_weights, write_caffemodel

    d = tmp_path_factory.mktemp("weights101")
    path = str(d / "synth101_seed0.caffemodel")
    layers = synth_weights(101, seed=0)
    write_caffemodel(path, "ResNet-101", layers)
    return path, layers


def _oracle(proto, layers, img):
    from oracle import oracle as O

    O.set_threads(min(16, os.cpu_count() or 1))
    return O.OracleNet(proto, layers).forward(data=img)


@pytest.mark.parametrize("hw", [(64, 64), (240, 320)])
def test_resnet101_fp32_matches_oracle(gpu_caffe, synth101, hw):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth101
    h, w = hw
    proto = deepercut_prototxt(101, h, w)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True)
    # 101 layers deep: 1 + 3*(3+4+23+3) + 1 projection per stage ... = 104 convolutions + 6 head layers lowered to
    # 104 + 2 launches (+ the max-pool); what matters here: every residual block of the 101 table is in the plan
    names = list(net.blobs)
    assert "res4b22" in names and "res4b23" not in names and "res3b3" in names and "res3b4" not in names
    assert net.plan_text().count("\n") - 1 == 104 + 2 + 1  # 104 trunk convolutions, the merged skip + deconvolution heads, the max-pool
    img = rand_image(7, h, w)
    net.blobs["data"].data[...] = img
    out = net.forward()
    ref = _oracle(proto, layers, img)
    assert sorted(out) == ["loc_pred", "next_pred", "prob"]
    for k in out:
        assert out[k].shape == ref[k].shape
        err = float(np.abs(out[k] - ref[k]).max())
        print(k, out[k].shape, "max abs err", err)
        assert err <= TOL, k
    full = gpu_caffe.Net(deepercut_prototxt(152, h, w), gpu_caffe.TEST, from_text=True).flops()  # host only: shapes
    assert 0.6 * full < net.flops() < 0.8 * full  # 101 is ~0.7 of 152's arithmetic (33.0 of 46.2 GFLOP at 240x320)


@pytest.mark.parametrize("fuse", [0, 2])
def test_resnet101_unfused_blobs_and_fused_outputs(gpu_caffe, synth101, fuse):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth101
    h, w = 72, 104
    proto = deepercut_prototxt(101, h, w)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, fuse=fuse)
    img = rand_image(8, h, w)
    net.blobs["data"].data[...] = img
    out = net.forward()
    ref = _oracle(proto, layers, img)
    for k in out:
        assert float(np.abs(out[k] - ref[k]).max()) <= TOL, k
    if fuse == 0:
        for name, r in ref.items():
            got = net.blobs[name].data
            assert got.shape == r.shape, name
            assert float(np.abs(got - r).max()) <= TOL * max(1.0, float(np.abs(r).max())), name


@pytest.mark.parametrize("hw,batch", [((64, 64), 2), ((240, 320), 8)])
def test_resnet101_fp16_matches_oracle(gpu_caffe, synth101, hw, batch):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth101
    h, w = hw
    proto = deepercut_prototxt(101, h, w, batch)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16")
    assert "dtype=f16" in net.plan_text()
    img = rand_image(9, h, w, n=batch)
    out = net.forward_batch(img)
    ref = _oracle(proto, layers, img)
    assert float(np.abs(out["prob"] - ref["prob"]).max()) <= 2.5e-3
    for k in ("loc_pred", "next_pred"):
        rng = max(1.0, float(np.abs(ref[k]).max()))
        err = float(np.abs(out[k] - ref[k]).max())
        assert err <= 4e-3 * rng, (k, err, rng)
        assert err > 1e-5, "suspiciously exact: is the fp16 path really running?"
