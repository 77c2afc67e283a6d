"""-m gpu: the streaming form of the float16 dense 1x1 layers (csrc/stream1x1.hip, tile name "ws1x1"), forced with DC_STREAM1X1=1 —
by default it is used only where the per-shape timing finds it faster.  Against the CPU oracle at the float16 path's stated bound
(single layers <= 2e-3 x range) AND bit for bit against a gather-GEMM tile without split-K on the same layer (the epilogue is the same instruction
sequence on the same operands; the matrix products accumulate the same K order): what the kernel changes is how the bytes travel.
Covers every K the kernel takes (64, 128, 256, 512), ragged pixel counts (M % 32 != 0, M < 32, a single pixel), batches, the
shortcut + ReLU and plain epilogues, layers without BatchNorm / Scale (no scale, no shift), multi-tensor launches of a NetGroup, and the
reference's own 1x1 expansion layers inside the full net (ResNet-152.prototxt res2a_branch2c ... res5c_branch2c, res2a_branch1)."""
import os

import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu


def _oracle(proto, layers, **inputs):
    from oracle import oracle as O

    O.set_threads(min(16, os.cpu_count() or 1))
    return O.OracleNet(proto, layers).forward(**inputs)


def _net_text(n, cin, cout, h, w, shortcut, relu, affine):
    L = ['name: "s"', 'input: "data"'] + ["input_dim: %d" % d for d in (n, cin, h, w)]
    if shortcut:
        L += ['input: "sc"'] + ["input_dim: %d" % d for d in (n, cout, h, w)]
    L.append('layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: %d kernel_size: 1 bias_term: false } }' % cout)
    if affine:
        L.append('layer { name: "bn" type: "BatchNorm" bottom: "c" top: "c" batch_norm_param { use_global_stats: true } }')
        L.append('layer { name: "scale" type: "Scale" bottom: "c" top: "c" scale_param { bias_term: true } }')
    out = "c"
    if shortcut:
        L.append('layer { name: "sum" type: "Eltwise" bottom: "sc" bottom: "c" top: "sum" }')
        out = "sum"
    if relu:
        L.append('layer { name: "relu" type: "ReLU" bottom: "%s" top: "%s" }' % (out, out))
    return "\n".join(L) + "\n", out


CASES = [  # n, cin, cout, h, w, shortcut, relu, affine
    (8, 256, 1024, 17, 23, True, True, True),    # res4x_branch2c's channels; 8*17*23 = 3128 pixels = 97.75 steps
    (2, 128, 512, 33, 19, True, True, True),     # res3x_branch2c
    (2, 64, 256, 40, 56, True, True, True),      # res2x_branch2c
    (2, 512, 2048, 9, 13, True, True, True),     # res5x_branch2c
    (2, 64, 256, 40, 56, False, False, True),    # res2a_branch1: no shortcut, no ReLU
    (1, 256, 512, 5, 5, True, False, True),      # 25 pixels: less than one step; shortcut without ReLU
    (1, 64, 256, 1, 1, False, True, False),      # a single pixel; no BatchNorm / Scale: no scale, no shift vector
    (3, 128, 256, 7, 9, True, True, False),      # 189 pixels, no affine, batch 3
    (1, 256, 256, 34, 46, False, True, True),    # one 256-channel slice: a grid of pixel chunks only
    (1, 512, 256, 16, 16, True, True, True),     # K = 512 with a single slice
    # walks of 1-2, 2-3, 4-5 and 6-7 steps per workgroup (64 pixel ranges at N = 1024): the peeled first steps, the steady loop, the tail
    (1, 256, 1024, 45, 46, True, True, True),    # 2070 pixels = 65 steps
    (1, 256, 1024, 65, 71, True, True, True),    # 4615 pixels = 145 steps
    (2, 256, 1024, 67, 70, True, False, True),   # 9380 pixels = 294 steps
    (3, 256, 1024, 61, 70, False, True, True),   # 12810 pixels = 401 steps
    (2, 64, 256, 100, 101, True, True, True),    # the 4-wave form (512 ranges): 20200 pixels = 632 steps, 1-2 per workgroup
    (1, 128, 512, 130, 131, True, True, True),   # K = 128: 17030 pixels = 533 steps over 128 ranges: 4-5 steps
]


def _weights(rs, cin, cout, affine):
    w = [("c", "Convolution", [(rs.randn(cout, cin, 1, 1) / np.sqrt(cin)).astype(np.float32)])]
    if affine:
        w.append(("bn", "BatchNorm", [rs.randn(cout).astype(np.float32) * 0.1, rs.uniform(0.5, 1.5, cout).astype(np.float32), np.array([1.0], np.float32)]))
        w.append(("scale", "Scale", [rs.uniform(0.5, 1.5, cout).astype(np.float32), rs.randn(cout).astype(np.float32) * 0.1]))
    return w


DIRECT_TILE = ("40", "d128x64x64_w221_s2")  # DC_CONV_VARIANT index / name of a tile WITHOUT in-workgroup split-K: the same summation order


def _run(caffe, proto, weights, inputs, out, mode, monkeypatch):
    monkeypatch.setenv("DC_STREAM1X1", mode)
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    if mode == "0":
        monkeypatch.setenv("DC_CONV_VARIANT", DIRECT_TILE[0])
    else:
        monkeypatch.delenv("DC_CONV_VARIANT", raising=False)
    net = caffe.Net(proto, caffe.TEST, from_text=True, dtype="f16")
    for name, _t, blobs in weights:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    for k, v in inputs.items():
        net.blobs[k].data[...] = v
    net.forward()
    return net.blobs[out].data.copy(), net.plan_text()


@pytest.mark.parametrize("case", CASES)
def test_single_layers_match_the_oracle_and_the_tiles_bit_for_bit(gpu_caffe, case, monkeypatch):
    n, cin, cout, h, w, shortcut, relu, affine = case
    proto, out = _net_text(n, cin, cout, h, w, shortcut, relu, affine)
    rs = np.random.RandomState(cin + cout + h)
    weights = _weights(rs, cin, cout, affine)
    inputs = {"data": rs.randn(n, cin, h, w).astype(np.float32)}
    if shortcut:
        inputs["sc"] = rs.randn(n, cout, h, w).astype(np.float32)
    got, plan = _run(gpu_caffe, proto, weights, inputs, out, "1", monkeypatch)
    assert "ws1x1" in plan, plan
    direct, plan0 = _run(gpu_caffe, proto, weights, inputs, out, "0", monkeypatch)
    assert "ws1x1" not in plan0 and DIRECT_TILE[1] in plan0, plan0
    assert np.array_equal(got, direct), float(np.abs(got - direct).max())
    ref = _oracle(proto, weights, **inputs)[out]
    assert got.shape == ref.shape
    assert float(np.abs(got - ref).max()) <= 2e-3 * max(1.0, float(np.abs(ref).max()))


def test_layers_the_form_does_not_take_keep_their_tiles(gpu_caffe, monkeypatch):
    """Stride 2, 3x3, a channel count that is not a whole slice, float32: lowered as before even when the form is forced."""
    monkeypatch.setenv("DC_STREAM1X1", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    base = ['name: "s"', 'input: "data"'] + ["input_dim: %d" % d for d in (1, 64, 16, 16)]
    for conv in ("num_output: 256 kernel_size: 1 stride: 2", "num_output: 256 kernel_size: 3 pad: 1", "num_output: 192 kernel_size: 1"):
        proto = "\n".join(base + ['layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { %s bias_term: false } }' % conv]) + "\n"
        assert "ws1x1" not in gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16").plan_text(), conv
    proto = "\n".join(base + ['layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { num_output: 256 kernel_size: 1 bias_term: false } }']) + "\n"
    f32_plan = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True).plan_text()  # float32: never this kernel (its own form is ws1x1f, round 6)
    assert "ws1x1<" not in f32_plan and "ws1x1f<" in f32_plan
    assert "ws1x1" in gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16").plan_text()


@pytest.mark.parametrize("hw,n", [((104, 136), 2), ((240, 320), 1)])
def test_full_net_with_every_eligible_layer_on_the_form(gpu_caffe, synth152, hw, n, monkeypatch):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w, n)
    monkeypatch.setenv("DC_STREAM1X1", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16")
    img = rand_image(11, h, w, n=n)
    net.blobs["data"].data[...] = img
    net.forward()
    # 3 + 8 + 36 + 3 branch2c expansions and res2a_branch1 (the other projections are stride 2)
    assert sum("ws1x1" in ln for ln in net.plan_text().splitlines()) == 51
    ref = _oracle(proto, layers, data=img)
    assert float(np.abs(net.blobs["prob"].data - ref["prob"]).max()) <= 2.5e-3
    for k in ("loc_pred", "next_pred"):
        assert float(np.abs(net.blobs[k].data - ref[k]).max()) <= 4e-3 * max(1.0, float(np.abs(ref[k]).max())), k
    monkeypatch.setenv("DC_STREAM1X1", "0")
    off = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16")
    off.blobs["data"].data[...] = img
    off.forward()
    assert "ws1x1" not in off.plan_text()
    # (the tiles the cost model picks for these layers may split K inside the workgroup: another summation order, float16 rounding apart)
    assert float(np.abs(net.blobs["prob"].data - off.blobs["prob"].data).max()) <= 1e-3
    for k in ("loc_pred", "next_pred"):
        r = off.blobs[k].data
        assert float(np.abs(net.blobs[k].data - r).max()) <= 2e-3 * max(1.0, float(np.abs(r).max())), k


def test_group_launches_walk_the_members_tensors_in_one_launch(gpu_caffe, synth152, monkeypatch):
    """NetGroup: the four 'scales' of a small pyramid, every eligible layer ONE ws1x1 launch over the four tensors (different pixel counts,
    none a multiple of 32 steps) — against the members' own forwards with the form off (the other layers of a group may run on other tiles
    than a member alone: float16 rounding of a different summation order, not bit equality)."""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    shapes = [(2, 40, 56), (2, 64, 64), (2, 72, 104), (2, 104, 136)]
    imgs = [rand_image(60 + i, h, w, n=n) for i, (n, h, w) in enumerate(shapes)]
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    monkeypatch.setenv("DC_WINOGRAD", "0")
    monkeypatch.setenv("DC_STREAM1X1", "1")
    n, h, w = shapes[0]
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w, n), path, gpu_caffe.TEST, from_text=True, dtype="f16", hipgraph=1)
    grp = gpu_caffe.NetGroup.for_shapes(net, shapes, lanes=1)
    outs = grp.forward_batch(imgs)
    text = grp.plan_text()
    assert sum("conv_gemm_mp<ws1x1>" in ln for ln in text.splitlines()) == 51, text[:600]
    monkeypatch.setenv("DC_STREAM1X1", "0")
    for (n, h, w), img, o in zip(shapes, imgs, outs):
        ref = gpu_caffe.Net(deepercut_prototxt(152, h, w, n), path, gpu_caffe.TEST, from_text=True, dtype="f16")
        ref.blobs["data"].data[...] = img
        ref.forward()
        assert "ws1x1" not in ref.plan_text()
        assert float(np.abs(o["prob"] - ref.blobs["prob"].data).max()) <= 1e-3
        for k in ("loc_pred", "next_pred"):
            r = ref.blobs[k].data
            assert float(np.abs(o[k] - r).max()) <= 2e-3 * max(1.0, float(np.abs(r).max())), (k, (n, h, w))
