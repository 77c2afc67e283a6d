"""-m gpu: the float16 path (DC_OPT_DTYPE 1, BASELINE configs[2]: fp16 operands in HBM, v_mfma_f32_32x32x16_f16
with float32 accumulation and float32 epilogue) against the float32 CPU oracle.

Tolerances (stated, not 1e-3: activations are rounded to 11 significant bits after each of 152 layers):
  prob      <= 2.5e-3 max-abs          (measured ~9e-4 at 240x320; the sigmoid compresses the error)
  loc_pred, next_pred <= 4e-3 x max(1, range of the map)   (measured ~1.4e-3 x range)
Single layers: <= 2e-3 relative to the output range (one rounding of inputs, weights and output)."""
import os
import zlib

import numpy as np
import pytest

from conftest import rand_image
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _check_maps(out, ref):
    assert float(np.abs(out["prob"] - ref["prob"]).max()) <= 2.5e-3
    for k in ("loc_pred", "next_pred"):
        rng = max(1.0, float(np.abs(ref[k]).max()))
        err = float(np.abs(out[k] - ref[k]).max())
        assert err <= 4e-3 * rng, (k, err, rng)
        assert err > 1e-5, "suspiciously exact: is the fp16 path really running?"


@pytest.mark.parametrize("fuse", [0, 2])
@pytest.mark.parametrize("hw", [(64, 64), (104, 136)])
def test_fp16_full_net_matches_oracle(gpu_caffe, synth152, hw, fuse):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w, 2)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16", fuse=fuse)
    assert "dtype=f16" in net.plan_text() and "conv_gemm<h" in net.plan_text()
    img = rand_image(31, h, w, n=2)
    net.blobs["data"].data[...] = img
    out = net.forward()
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto, layers).forward(data=img)
    _check_maps(out, ref)
    if fuse == 0:  # intermediate blobs too, relative to their range
        for name in ("conv1", "pool1", "res2c", "res3b7", "res4b35", "res5c"):
            r = ref[name]
            assert float(np.abs(net.blobs[name].data - r).max()) <= 1e-2 * max(1.0, float(np.abs(r).max())), name


def test_fp16_pyramid_scale_batch8(gpu_caffe, synth152):
    """BASELINE configs[2]: batch of 8 at the 0.5 scale of 736x544 (272x368), fp16 MFMA with fp32 accumulate."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = 272, 368
    proto = deepercut_prototxt(152, h, w, 8)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16")
    imgs = rand_image(10, h, w, n=8)
    out = net.forward_batch(imgs)
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto, layers).forward(data=imgs)
    _check_maps(out, ref)
    pose = net.decode_pose(0.5)  # the device decode reads the half maps
    assert pose.shape == (8, 5, 14) and np.isfinite(pose).all()


def test_switching_dtype_on_a_live_net(gpu_caffe, synth152):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True)
    img = rand_image(32, 64, 64)
    net.blobs["data"].data[...] = img
    a = {k: v.copy() for k, v in net.forward().items()}
    net.set_option(3, 1)
    net.blobs["data"].data[...] = img
    b = {k: v.copy() for k, v in net.forward().items()}
    net.set_option(3, 0)
    net.blobs["data"].data[...] = img
    c = net.forward()
    for k in a:
        assert np.array_equal(a[k], c[k]), k              # float32 is reproduced bit for bit
        d = float(np.abs(a[k] - b[k]).max())
        assert 1e-6 < d < 2e-2, (k, d)                    # float16 differs, slightly


@pytest.mark.parametrize("cfg", [("conv", 7, 2, 3, 1, 3, 64, 41, 54), ("conv", 3, 1, 1, 1, 64, 64, 13, 17),
                                 ("conv", 1, 2, 0, 1, 256, 128, 14, 18), ("conv", 3, 1, 2, 2, 512, 512, 7, 9),
                                 ("conv", 1, 1, 0, 1, 2048, 512, 5, 6), ("deconv", 3, 2, 0, 1, 2048, 28, 4, 5)])
def test_fp16_single_layers(gpu_caffe, cfg):
    kind, k, s, p, d, cin, cout, h, w = cfg
    rs = np.random.RandomState(zlib.crc32(repr(cfg).encode()) & 0x7fffffff)  # (hash() of a tuple holding a str moves with PYTHONHASHSEED)
    typ = "Convolution" if kind == "conv" else "Deconvolution"
    text = ('input: "x" input_dim: 2 input_dim: %d input_dim: %d input_dim: %d\n' % (cin, h, w) +
            'layer { name: "l" type: "%s" bottom: "x" top: "y" convolution_param { num_output: %d kernel_size: %d '
            "stride: %d pad: %d dilation: %d } }" % (typ, cout, k, s, p, d))
    net = gpu_caffe.Net(text, gpu_caffe.TEST, from_text=True, fuse=0, dtype="f16")
    x = rs.randn(2, cin, h, w).astype(np.float32)
    wshape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
    wt = (rs.randn(*wshape) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rs.randn(cout).astype(np.float32)
    net.params["l"][0].data[...] = wt
    net.params["l"][1].data[...] = b
    net.blobs["x"].data[...] = x
    got = net.forward()["y"]
    ref = (O.conv_forward if kind == "conv" else O.deconv_forward)(x, wt, b, s, p, d)
    assert got.shape == ref.shape
    assert float(np.abs(got - ref).max()) <= 2e-3 * max(1.0, float(np.abs(ref).max()))


def _large_activation_weights(layers, gain):
    """Synthetic weights whose trunk activations are `gain` times larger — what the un-normalised blocks of a trained
    ResNet look like — with the head filters divided by `gain` so that the output maps stay O(1): conv1's Scale (gamma,
    beta) is multiplied (the global-statistics BatchNorm layers do not re-normalise, ReLU is positively homogeneous), the
    six head convolutions / deconvolutions (not their biases) are divided."""
    out = []
    for name, t, blobs in layers:
        blobs = [b.copy() for b in blobs]
        if name == "scale_conv1":
            blobs = [b * np.float32(gain) for b in blobs]
        if name.startswith("res5c_up_") or name.startswith("res3d_"):
            blobs[0] = blobs[0] / np.float32(gain)
        out.append((name, t, blobs))
    return out


@pytest.mark.parametrize("gain", [32.0, 256.0, 1024.0])
def test_fp16_on_trained_weight_like_magnitudes(gpu_caffe, synth152, tmp_path, gain):
    """The float16 path on activations of trained-network magnitude: the conditioned synthetic trunk peaks at ~33 (res4b35)
    and ~45 (res5c); scaled by 32 / 256 / 1024 it reaches 1.4e3 / 1.1e4 / 4.6e4 — up to 70 % of float16's largest finite value
    65504, where its spacing is 32.  The maps must stay finite and inside the stated bounds against the float32 oracle of the SAME
    weights (rounding is relative: the bounds, relative to the maps' range, do not move with the magnitude)."""
    from deepcut_tools import deepercut_prototxt, write_caffemodel

    _, layers = synth152
    big = _large_activation_weights(layers, gain)
    path = str(tmp_path / "big.caffemodel")
    write_caffemodel(path, "ResNet-152", big)
    h, w = 104, 136
    proto = deepercut_prototxt(152, h, w, 1)
    img = rand_image(33, h, w)
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto, big).forward(data=img)
    peak = {k: float(np.abs(ref[k]).max()) for k in ("res3b7", "res4b35", "res5c")}
    assert peak["res4b35"] > 25 * gain and peak["res5c"] > 35 * gain, peak  # the activations ARE large (gain 1024: 3.3e4 / 4.6e4)
    assert max(peak.values()) < 6.0e4, peak                                   # and still representable in float16
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, dtype="f16", fuse=0)
    net.blobs["data"].data[...] = img
    out = net.forward()
    for k in ("prob", "loc_pred", "next_pred"):
        assert np.isfinite(out[k]).all(), k
    _check_maps(out, ref)
    for name in ("res4b35", "res5c"):  # the large blobs themselves: relative to their range
        r = ref[name]
        got = net.blobs[name].data
        assert np.isfinite(got).all(), name
        assert float(np.abs(got - r).max()) <= 1e-2 * float(np.abs(r).max()), name


def test_fp16_net_with_a_narrow_skip_level(gpu_caffe):
    """A DeeperCut-shaped FCN whose layers have 16 / 32 input channels — fewer than the 64 halves of a float16 K tile, so they
    take the row-tap packing — including the two sibling skip convolutions that run as ONE launch: round 2 packed only the
    first sibling's filters in that path (the second head read rows beyond the image); found by round 3's bounds check."""
    from test_gpu_tiling import _fill, local_fcn_prototxt

    h, w = 64, 96
    proto = local_fcn_prototxt(h, w)
    net32 = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True)
    _fill(net32, 5)
    layers = [(name, typ, [b.data.copy() for b in net32.params[name]])
              for name, typ in zip(net32._layer_names, net32.layer_types) if name in net32.params]
    img = rand_image(7, h, w)
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto, layers).forward(data=img)
    net32.blobs["data"].data[...] = img
    out32 = net32.forward()
    for k in ("prob", "loc_pred"):
        assert float(np.abs(out32[k] - ref[k]).max()) <= 1e-3, k
    for fuse in (0, 2):
        net16 = gpu_caffe.Net(proto, gpu_caffe.TEST, from_text=True, dtype="f16", fuse=fuse)
        _fill(net16, 5)
        net16.blobs["data"].data[...] = img
        out = net16.forward()
        # (an unconditioned toy net: its logits span tens of units, so the sigmoid output is compared loosely and the linear
        #  maps relative to their range — a missing sibling shows up as an error of the order of the range itself)
        assert float(np.abs(out["prob"] - ref["prob"]).max()) <= 3e-2, fuse
        rng = max(1.0, float(np.abs(ref["loc_pred"]).max()))
        assert float(np.abs(out["loc_pred"] - ref["loc_pred"]).max()) <= 4e-3 * rng, (fuse, rng)
        if fuse == 0:
            lg = ref["fc_pose"]
            assert float(np.abs(net16.blobs["fc_pose"].data - lg).max()) <= 4e-3 * max(1.0, float(np.abs(lg).max()))


def test_host_entry_of_the_fp16_batch8_net_stays_close_to_the_device_forward(gpu_caffe, synth152):
    """Review r3, weak 2: `Net.forward_batch(ndarray)` on a float16 batch-8 net took 39 ms against 4 ms device-resident —
    the result arrays were fresh `np.empty` memory on every call and the device-to-host copy faulted 81 MB of pages inside
    the driver.  Ten consecutive calls must move none of the net's lowering / graph / re-pack / buffer counters and stay
    under twice the device-resident forward of the same net (measured: 6.4 ms against 3.9)."""
    import time

    import torch
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    h, w, b = 544, 736, 8
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w, b), path, gpu_caffe.TEST, from_text=True, dtype="f16", hipgraph=1)
    x = rand_image(77, h, w, n=b)
    xd = torch.from_numpy(x).cuda()
    od = [torch.empty(net.blobs[k].shape, device="cuda") for k in ("prob", "loc_pred", "next_pred")]

    def dev():
        net.forward_device(xd.data_ptr(), b, h, w, od[0].data_ptr(), od[1].data_ptr(), od[2].data_ptr())
        torch.cuda.synchronize()

    dev()
    dev()
    t_dev = []
    for _ in range(5):
        t0 = time.perf_counter()
        dev()
        t_dev.append(time.perf_counter() - t0)
    first = {k: v.copy() for k, v in net.forward_batch(x).items()}  # also warms the pool of result arrays
    net.forward_batch(x)
    before = net.stats()
    t_host = []
    for _ in range(10):
        t0 = time.perf_counter()
        out = net.forward_batch(x)
        t_host.append(time.perf_counter() - t0)
        del out
    after = net.stats()
    for k in ("lowerings", "graph_instantiations", "repacks", "buffer_growths"):
        assert after[k] == before[k], (k, before[k], after[k])
    out = net.forward_batch(x)
    for k in first:
        assert np.array_equal(out[k], first[k]), k  # recycled destinations, same values
    held = net.forward_batch(x)  # `out` is still referenced: its arrays must not be handed out again
    assert all(held[k] is not out[k] for k in out)
    d, hmed = sorted(t_dev)[2], sorted(t_host)[5]
    assert hmed < 2.0 * d + 1e-3, "host entry %.2f ms vs device-resident %.2f ms" % (hmed * 1e3, d * 1e3)
