"""-m gpu: the float32 stem on the streaming skeleton (csrc/stream1x1_f32.hip, tile name "ws7x7f": conv1 of ResNet-152.prototxt — 7x7,
stride 2, pad 3, 3 -> 64 channels, + BatchNorm / Scale / ReLU; the reference: im2col + SGEMM, base_conv_layer.cpp:257-280), forced with
DC_STEM=1; by default it is used where the per-shape timing finds it faster.  Against the CPU oracle at 2e-5 x range per layer (the float32
path's whole-net bound is 1e-3) and against the row-tap gather-GEMM launch of the same layer (same operands, another summation grouping).
Odd and tiny images (the zero padding on every side, rows of requests that hang over the image), batches, with and without the affine / the
ReLU, the geometries the form does not take, and conv1 inside the full net."""
import numpy as np
import pytest

from conftest import rand_image
from test_gpu_stem import CASES, _net_text, _oracle

pytestmark = pytest.mark.gpu


def _run(caffe, proto, weights, x, mode, monkeypatch):
    monkeypatch.setenv("DC_STEM", mode)
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    net = caffe.Net(proto, caffe.TEST, from_text=True)
    for name, _t, blobs in weights:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    net.blobs["data"].data[...] = x
    net.forward()
    return net.blobs["conv1"].data.copy(), net.plan_text()


@pytest.mark.parametrize("case", CASES + [(2, 300, 17, True, True), (5, 16, 16, True, True)])
def test_stem_matches_the_oracle_and_the_row_tap_launch(gpu_caffe, case, monkeypatch):
    n, h, w, relu, affine = case
    proto = _net_text(n, h, w, relu, affine)
    rs = np.random.RandomState(h + w)
    weights = [("conv1", "Convolution", [(rs.randn(64, 3, 7, 7) / np.sqrt(147.0)).astype(np.float32)])]
    if affine:
        weights.append(("bn", "BatchNorm", [rs.randn(64).astype(np.float32) * 0.1, rs.uniform(0.5, 1.5, 64).astype(np.float32), np.array([1.0], np.float32)]))
        weights.append(("scale", "Scale", [rs.uniform(0.5, 1.5, 64).astype(np.float32), rs.randn(64).astype(np.float32) * 0.1]))
    x = (rs.randn(n, 3, h, w) * 50.0).astype(np.float32)  # mean-subtracted pixel values
    got, plan = _run(gpu_caffe, proto, weights, x, "1", monkeypatch)
    assert "ws7x7f" in plan, plan
    direct, plan0 = _run(gpu_caffe, proto, weights, x, "0", monkeypatch)
    assert "ws7x7f" not in plan0
    ref = _oracle(proto, weights, data=x)["conv1"]
    assert got.shape == ref.shape
    rng = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(got - ref).max()) <= 2e-5 * rng
    assert float(np.abs(got - direct).max()) <= 2e-5 * rng


def test_only_the_stem_geometry_takes_it(gpu_caffe, monkeypatch):
    monkeypatch.setenv("DC_STEM", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    base = ['name: "s"', 'input: "data"'] + ["input_dim: %d" % d for d in (1, 3, 32, 32)]
    for conv in ("num_output: 64 kernel_size: 7 pad: 3 stride: 1", "num_output: 64 kernel_size: 5 pad: 2 stride: 2", "num_output: 32 kernel_size: 7 pad: 3 stride: 2",
                 "num_output: 64 kernel_size: 7 pad: 2 stride: 2"):
        proto = "\n".join(base + ['layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { %s bias_term: false } }' % conv]) + "\n"
        assert "ws7x7f" not in gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True).plan_text(), conv
    proto = "\n".join(base + ['layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: 64 kernel_size: 7 pad: 3 stride: 2 bias_term: false } }']) + "\n"
    assert "ws7x7f" in gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True).plan_text()
    assert "ws7x7f" not in gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16").plan_text()  # (a float16 net has its own stem kernel)


def test_conv1_inside_the_full_net_and_the_autotuner(gpu_caffe, synth152, monkeypatch):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w, n = 104, 136, 2
    proto = deepercut_prototxt(152, h, w, n)
    monkeypatch.setenv("DC_STEM", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True)
    img = rand_image(13, h, w, n=n)
    net.blobs["data"].data[...] = img
    net.forward()
    assert sum("ws7x7f" in ln for ln in net.plan_text().splitlines()) == 1
    ref = _oracle(proto, layers, data=img)
    assert float(np.abs(net.blobs["prob"].data - ref["prob"]).max()) <= 1e-3
    for k in ("loc_pred", "next_pred"):
        assert float(np.abs(net.blobs[k].data - ref[k]).max()) <= 1e-3 * max(1.0, float(np.abs(ref[k]).max())), k
    monkeypatch.delenv("DC_STEM", raising=False)
    monkeypatch.setenv("DC_AUTOTUNE", "1")
    tuned = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True)
    tuned.blobs["data"].data[...] = img
    tuned.forward()
    assert any(t[0] == "ws7x7f" for e in tuned.tune_report() for t in e["timed"])
