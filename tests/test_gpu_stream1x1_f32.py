"""-m gpu: the float32 form of the streaming dense 1x1 layers (csrc/stream1x1_f32.hip, tile name "ws1x1f"), forced with DC_STREAM1X1=1 —
by default it is used only where the per-shape timing finds it faster.  Against the CPU oracle (the reference's arithmetic for these
layers: one SGEMM per image, base_conv_layer.cpp:326-341, + BatchNorm / Scale / Eltwise / ReLU) at 2e-5 x range — the float32 path's
stated bound for whole nets is 1e-3; a single layer of K <= 512 products differs by summation order only — and against a gather-GEMM tile on
the same layer (not bit for bit: four interleaved K runs and two accumulators are another grouping of the same sums).
Covers every K the kernel takes (64, 128, 256, 512), ragged pixel counts (M % 16 != 0, M < 16, a single pixel), walks of 0 / 1 / 2 / 3 / many steps per
workgroup (the peeled first steps, the steady loop), batches, shortcut + ReLU and plain epilogues, layers without BatchNorm / Scale, the
shapes the form does not take, and the reference's own expansion layers inside the full net (ResNet-152.prototxt res4*_branch2c, res5*_branch2c)."""
import os

import numpy as np
import pytest

from conftest import rand_image
from test_gpu_stream1x1 import _net_text, _weights

pytestmark = pytest.mark.gpu


def _oracle(proto, layers, **inputs):
    from oracle import oracle as O

    O.set_threads(min(16, os.cpu_count() or 1))
    return O.OracleNet(proto, layers).forward(**inputs)


CASES = [  # n, cin, cout, h, w, shortcut, relu, affine
    (1, 256, 1024, 34, 46, True, True, True),    # res4x_branch2c at 544x736 (BASELINE configs[1]): 1564 pixels = 97.75 steps over 16 ranges
    (1, 512, 2048, 34, 46, True, True, True),    # res5x_branch2c: 32 channel slices x 8 ranges, 12-13 steps each
    (2, 256, 1024, 17, 23, True, True, True),    # batch 2
    (1, 256, 512, 5, 5, True, False, True),      # 25 pixels: 2 steps, most ranges empty; shortcut without ReLU
    (1, 256, 256, 1, 1, False, True, False),     # a single pixel; no BatchNorm / Scale: no scale, no shift vector
    (3, 512, 256, 7, 9, True, True, False),      # 189 pixels, no affine, batch 3
    (1, 256, 64, 34, 46, False, True, True),     # one 64-channel slice: 98 ranges of one step
    (1, 512, 64, 16, 16, False, False, True),    # K = 512, one slice, 16 ranges of one step, plain epilogue
    (1, 256, 1024, 9, 13, True, True, True),     # 117 pixels = 8 steps over 8 ranges: one step each
    (1, 256, 1024, 14, 16, True, True, True),    # 224 pixels = 14 steps over 8 ranges: 1-2
    (1, 256, 1024, 20, 20, True, True, True),    # 400 pixels = 25 steps over 16 ranges: 1-2
    (1, 256, 1024, 25, 31, True, True, True),    # 775 pixels = 49 steps: 3-4 (all peeled steps, D = 4)
    (2, 256, 1024, 67, 70, True, False, True),   # 9380 pixels = 587 steps: 36-37 per range, the steady loop
    (2, 512, 1024, 40, 33, True, True, True),    # K = 512 (D = 3) in the steady loop: 165 steps over 16 ranges
    # K = 128: two pixel rows per 1 KiB request (res3x_branch2c)
    (1, 128, 512, 68, 92, True, True, True),     # res3x_branch2c at 544x736: 6256 pixels = 391 steps over 32 ranges, 12-13 each
    (2, 128, 512, 33, 19, True, True, True),     # 1254 pixels (ragged: 78.4 steps), batch 2
    (1, 128, 64, 5, 7, False, False, True),      # 35 pixels: 3 steps (the last one 3 rows: half a request), one slice, plain epilogue
    (1, 128, 256, 1, 1, True, True, False),      # a single pixel, no affine
    (3, 128, 128, 9, 11, False, True, True),     # 297 pixels = 19 steps over 19 ranges, batch 3
    # K = 64: four pixel rows per request, four product groups per step (res2x_branch2c, res2a_branch1, res2a_branch2a)
    (1, 64, 256, 136, 184, True, True, True),    # res2x_branch2c at 544x736: 25024 pixels = 1564 steps over 128 ranges
    (2, 64, 256, 40, 56, False, False, True),    # res2a_branch1: no shortcut, no ReLU
    (1, 64, 64, 7, 9, False, True, True),        # res2a_branch2a's channels; 63 pixels: the last step 15 rows
    (1, 64, 128, 3, 2, True, True, False),       # 6 pixels: one step of two requests' worth, no affine
    (3, 64, 256, 21, 23, True, False, True),     # 1449 pixels = 90.6 steps, batch 3, shortcut without ReLU
]


def _run(caffe, proto, weights, inputs, out, mode, monkeypatch):
    monkeypatch.setenv("DC_STREAM1X1", mode)
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    net = caffe.Net(proto, caffe.TEST, from_text=True)
    for name, _t, blobs in weights:
        for p, b in zip(net.params[name], blobs):
            p.data[...] = b
    for k, v in inputs.items():
        net.blobs[k].data[...] = v
    net.forward()
    return net.blobs[out].data.copy(), net.plan_text()


@pytest.mark.parametrize("case", CASES)
def test_single_layers_match_the_oracle_and_the_tiles(gpu_caffe, case, monkeypatch):
    n, cin, cout, h, w, shortcut, relu, affine = case
    proto, out = _net_text(n, cin, cout, h, w, shortcut, relu, affine)
    rs = np.random.RandomState(cin + cout + h)
    weights = _weights(rs, cin, cout, affine)
    inputs = {"data": rs.randn(n, cin, h, w).astype(np.float32)}
    if shortcut:
        inputs["sc"] = rs.randn(n, cout, h, w).astype(np.float32)
    got, plan = _run(gpu_caffe, proto, weights, inputs, out, "1", monkeypatch)
    assert "ws1x1f" in plan, plan
    direct, plan0 = _run(gpu_caffe, proto, weights, inputs, out, "0", monkeypatch)
    assert "ws1x1f" not in plan0, plan0
    ref = _oracle(proto, weights, **inputs)[out]
    assert got.shape == ref.shape
    rng = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(got - ref).max()) <= 2e-5 * rng
    assert float(np.abs(got - direct).max()) <= 2e-5 * rng


def test_layers_the_form_does_not_take_keep_their_tiles(gpu_caffe, monkeypatch):
    """K other than 64 / 128 / 256 / 512, stride 2, 3x3, a channel count that is not a whole 64-slice: lowered as before even when the form is forced."""
    monkeypatch.setenv("DC_STREAM1X1", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")

    def plan(cin, conv):
        base = ['name: "s"', 'input: "data"'] + ["input_dim: %d" % d for d in (1, cin, 16, 16)]
        proto = "\n".join(base + ['layer { name: "c" type: "Convolution" bottom: "data" top: "c" convolution_param { %s bias_term: false } }' % conv]) + "\n"
        return gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True).plan_text()

    for cin, conv in ((256, "num_output: 256 kernel_size: 1 stride: 2"), (256, "num_output: 256 kernel_size: 3 pad: 1"), (256, "num_output: 96 kernel_size: 1"),
                      (32, "num_output: 256 kernel_size: 1"), (192, "num_output: 256 kernel_size: 1"), (1024, "num_output: 256 kernel_size: 1")):
        assert "ws1x1f" not in plan(cin, conv), (cin, conv)
    assert "ws1x1f" in plan(256, "num_output: 256 kernel_size: 1")
    assert "ws1x1f" in plan(512, "num_output: 128 kernel_size: 1")
    assert "ws1x1f" in plan(128, "num_output: 512 kernel_size: 1")
    assert "ws1x1f" in plan(64, "num_output: 256 kernel_size: 1")


@pytest.mark.parametrize("hw,n", [((104, 136), 2), ((240, 320), 1)])
def test_full_net_with_every_eligible_layer_on_the_form(gpu_caffe, synth152, hw, n, monkeypatch):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w, n)
    monkeypatch.setenv("DC_STREAM1X1", "1")
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True)
    img = rand_image(11, h, w, n=n)
    net.blobs["data"].data[...] = img
    net.forward()
    # the 3 + 8 + 36 + 3 branch2c expansions (K = 64 / 128 / 256 / 512), res2a_branch1 / _branch2a and the K = 256 / 512 reductions at stride 1
    # (res5b / res5c_branch2a are K = 2048, conv4_x's K = 1024, the first reduction of a later stage has stride 2: not taken)
    took = [ln for ln in net.plan_text().splitlines() if "ws1x1f" in ln]
    assert len(took) >= 52, len(took)
    ref = _oracle(proto, layers, data=img)
    assert float(np.abs(net.blobs["prob"].data - ref["prob"]).max()) <= 1e-3
    for k in ("loc_pred", "next_pred"):
        assert float(np.abs(net.blobs[k].data - ref[k]).max()) <= 1e-3 * max(1.0, float(np.abs(ref[k]).max())), k


def test_the_autotuner_times_the_form_and_set_tile_takes_its_name(gpu_caffe, monkeypatch):
    monkeypatch.delenv("DC_STREAM1X1", raising=False)
    monkeypatch.setenv("DC_AUTOTUNE", "1")
    proto, out = _net_text(1, 256, 1024, 34, 46, True, True, True)
    net = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True)
    net.forward()
    rep = net.tune_report()
    assert any(t[0] == "ws1x1f" for e in rep for t in e["timed"]), rep
    sig = rep[0]["signature"]
    net.set_tile(sig, "ws1x1f")
    assert "ws1x1f" in net.plan_text()


def test_a_group_merges_members_on_the_form_as_the_direct_layer(gpu_caffe, synth152, monkeypatch):
    """NetGroup (float32): members whose own plans run the expansions on ws1x1f still merge them into multi-problem gather-GEMM launches
    (the form has no multi-problem kernel) — the same number of merged launches as with the form off, results within the float32 bound
    of the members' own forwards."""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    shapes = [(1, 72, 104), (1, 104, 136)]
    imgs = [rand_image(70 + i, h, w, n=n) for i, (n, h, w) in enumerate(shapes)]
    monkeypatch.setenv("DC_AUTOTUNE", "0")

    def run(mode):
        monkeypatch.setenv("DC_STREAM1X1", mode)
        n, h, w = shapes[0]
        net = gpu_caffe.Net(deepercut_prototxt(152, h, w, n), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
        grp = gpu_caffe.NetGroup.for_shapes(net, shapes, lanes=1)
        outs = grp.forward_batch(imgs)
        return outs, sum("conv_gemm_mp<" in ln for ln in grp.plan_text().splitlines()), net.plan_text()

    on, merged_on, member_plan = run("1")
    assert "ws1x1f" in member_plan
    off, merged_off, member_plan_off = run("0")
    assert "ws1x1f" not in member_plan_off
    assert merged_on == merged_off and merged_on > 100, (merged_on, merged_off)
    for a, b in zip(on, off):
        for k in ("prob", "loc_pred", "next_pred"):
            assert float(np.abs(a[k] - b[k]).max()) <= 1e-3 * max(1.0, float(np.abs(b[k]).max())), k
