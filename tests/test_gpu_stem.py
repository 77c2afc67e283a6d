"""-m gpu: the float16 stem kernel (csrc/stem_f16.hip, tile name "stem7x7": conv1 of ResNet-152.prototxt — 7x7, stride 2, pad 3,
3 -> 64 channels, + BatchNorm / Scale / ReLU), forced with DC_STEM=1; by default it is used where the per-shape timing finds it faster.
Against the CPU oracle at the float16 path's single-layer bound (2e-3 x range) and against the row-tap gather-GEMM launch of the same
layer (same operands, another summation grouping: float16 rounding apart).  Odd and tiny images (the zero padding on every side, tiles
that hang over the right / bottom edge, fewer pixels than one tile), batches, with and without the affine / the ReLU."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(proto, layers, **inputs):
    from oracle import oracle as O

    O.set_threads(min(16, os.cpu_count() or 1))
    return O.OracleNet(proto, layers).forward(**inputs)


def _net_text(n, h, w, relu, affine):
    L = ['name: "s"', 'input: "data"'] + ["input_dim: %d" % d for d in (n, 3, h, w)]
    L.append('layer { name: "conv1" type: "Convolution" bottom: "data" top: "conv1" convolution_param { num_output: 64 kernel_size: 7 pad: 3 stride: 2 bias_term: false } }')
    if affine:
        L.append('layer { name: "bn" type: "BatchNorm" bottom: "conv1" top: "conv1" batch_norm_param { use_global_stats: true } }')
        L.append('layer { name: "scale" type: "Scale" bottom: "conv1" top: "conv1" scale_param { bias_term: true } }')
    if relu:
        L.append('layer { name: "relu" type: "ReLU" bottom: "conv1" top: "conv1" }')
    return "\n".join(L) + "\n"


CASES = [  # n, h, w, relu, affine
    (1, 544, 736, True, True),   # the benchmark's image
    (2, 131, 77, True, True),    # odd sizes: tiles hang over both edges, batch 2
    (3, 9, 11, False, True),     # 5 x 6 outputs: less than one tile
    (1, 1, 1, True, False),      # a single pixel: one output, every tap but the centre in the padding; no affine
    (1, 64, 200, True, False),
    (2, 33, 129, False, False),  # 65 output columns: one column into the second tile
]


def _run(caffe, proto, weights, x, mode, monkeypatch):
    monkeypatch.setenv("DC_STEM", mode)
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    net = caffe.Net(proto, caffe.TEST, from_text=True, dtype="f16")
    for name, _t, blobs in weights:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    net.blobs["data"].data[...] = x
    net.forward()
    return net.blobs["conv1"].data.copy(), net.plan_text()


@pytest.mark.parametrize("case", CASES)
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
    assert "stem7x7" in plan, plan
    direct, plan0 = _run(gpu_caffe, proto, weights, x, "0", monkeypatch)
    assert "stem7x7" not in plan0
    ref = _oracle(proto, weights, data=x)["conv1"]
    assert got.shape == ref.shape
    rng = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(got - ref).max()) <= 2e-3 * rng
    assert float(np.abs(got - direct).max()) <= 1e-3 * rng


def test_only_the_stem_geometry_takes_it(gpu_caffe, monkeypatch):
    monkeypatch.setenv("DC_STEM", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    base = ['name: "s"', 'input: "data"'] + ["input_dim: %d" % d for d in (1, 3, 32, 32)]
    for conv in ("num_output: 64 kernel_size: 7 pad: 3 stride: 1", "num_output: 64 kernel_size: 5 pad: 2 stride: 2", "num_output: 32 kernel_size: 7 pad: 3 stride: 2",
                 "num_output: 64 kernel_size: 7 pad: 2 stride: 2"):
        proto = "\n".join(base + ['layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { %s bias_term: false } }' % conv]) + "\n"
        assert "stem7x7" not in gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16").plan_text(), conv
    proto = "\n".join(base + ['layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: 64 kernel_size: 7 pad: 3 stride: 2 bias_term: false } }']) + "\n"
    assert "stem7x7" not in gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True).plan_text()  # float32
    assert "stem7x7" in gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16").plan_text()


def test_full_net_on_the_stem_kernel(gpu_caffe, synth152, monkeypatch):
    from conftest import rand_image
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w, n = 104, 136, 2
    proto = deepercut_prototxt(152, h, w, n)
    monkeypatch.setenv("DC_STEM", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16")
    img = rand_image(12, h, w, n=n)
    net.blobs["data"].data[...] = img
    net.forward()
    assert sum("stem7x7" in ln for ln in net.plan_text().splitlines()) == 1
    ref = _oracle(proto, layers, data=img)
    assert float(np.abs(net.blobs["prob"].data - ref["prob"]).max()) <= 2.5e-3
    for k in ("loc_pred", "next_pred"):
        assert float(np.abs(net.blobs[k].data - ref[k]).max()) <= 4e-3 * max(1.0, float(np.abs(ref[k]).max())), k
