"""-m gpu: output-set selection (DC_OPT_OUTPUTS, `Net.set_outputs`).  The demo reads `prob` and `loc_pred` only
(python/pose/estimate_pose.py:231-241 of the reference); the 364-channel `next_pred` head is 23.3 of the 241 GFLOP of a 544x736
forward.  Without it the lowering drops the head's launches and the merged heads shrink from 406 to 42 channels."""
import os

import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu


def _head_variant(caffe):
    """index of a float32 tile that takes both the 406- and the 42-channel merged heads (multi-class instantiation); layers it
    cannot take (the stem's 32-deep row taps) fall back to the cost model's tile — the same one in both nets"""
    names = [n for n, _es in caffe.conv_variants()]
    if "32x64x64_w124_p4" not in names:
        pytest.skip("tile 32x64x64_w124_p4 is not in this build")
    return names.index("32x64x64_w124_p4")


def test_selected_outputs_equal_the_full_forward_and_cost_less(gpu_caffe, synth152, monkeypatch):
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = 104, 136
    img = rand_image(3, h, w)
    monkeypatch.setenv("DC_AUTOTUNE", "0")
    monkeypatch.setenv("DC_CONV_VARIANT", str(_head_variant(gpu_caffe)))  # same tile for every GEMM: same summation order
    full = gpu_caffe.Net(deepercut_prototxt(152, h, w), path, gpu_caffe.TEST, from_text=True)
    full.blobs["data"].data[...] = img
    ref = {k: v.copy() for k, v in full.forward().items()}
    assert sorted(ref) == ["loc_pred", "next_pred", "prob"]
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w), path, gpu_caffe.TEST, from_text=True, want=("prob", "loc_pred"))
    assert net.wanted_outputs == ["loc_pred", "prob"] and net.outputs == ["loc_pred", "next_pred", "prob"]
    net.blobs["data"].data[...] = img
    out = net.forward()
    assert sorted(out) == ["loc_pred", "prob"]
    for k in out:
        assert np.array_equal(out[k], ref[k]), k
    assert "res5c_up_next" not in net.plan_text() and "res3d_next" not in net.plan_text()
    assert " N=42 " in net.plan_text()
    with pytest.raises(gpu_caffe.DeepcutError):
        net.blobs["next_pred"].data
    # the batch entry refuses a map that is not computed, and serves the others
    got = net.forward_batch(img, want=("prob", "loc_pred"))
    assert np.array_equal(got["prob"], ref["prob"]) and np.array_equal(got["loc_pred"], ref["loc_pred"])
    with pytest.raises(gpu_caffe.DeepcutError):
        net.forward_batch(img, want=("next_pred",))
    # everything back
    net.set_outputs(None)
    net.blobs["data"].data[...] = img
    again = net.forward()
    for k in ref:
        assert np.array_equal(again[k], ref[k]), k


def test_flops_drop_by_the_pairwise_head_at_544x736(gpu_caffe):
    from deepcut_tools import deepercut_prototxt

    net = gpu_caffe.Net(deepercut_prototxt(152, 544, 736), gpu_caffe.TEST, from_text=True)
    f_all = net.flops()
    net.set_outputs(["prob", "loc_pred"])
    f_two = net.flops()
    # res5c_up_next: 2 * 2048 * 34 * 46 * 364 * 9 = 20.99 GFLOP; res3d_next: 2 * 364 * 68 * 92 * 512 = 2.33 GFLOP
    want = 2.0 * 2048 * 34 * 46 * 364 * 9 + 2.0 * 364 * 68 * 92 * 512
    assert abs((f_all - f_two) - want) < 1e3, (f_all, f_two, want)
    assert abs(f_all - 241.09e9) < 0.01e9


def test_default_tiles_and_the_demo_mirror(gpu_caffe, synth152):
    """with the tiles the library chooses by itself the two maps differ from the full forward's by float32 summation order only, and
    pose.estimate_pose leaves `next_pred` out by itself (all_outputs=True keeps it)"""
    from deepcut_tools import deepercut_prototxt
    from oracle import oracle as O
    from pose import estimate_pose as EP

    path, layers = synth152
    h, w = 104, 136
    img = rand_image(5, h, w)
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w), path, gpu_caffe.TEST, from_text=True, want=("prob", "loc_pred"))
    net.blobs["data"].data[...] = img
    out = net.forward()
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(deepercut_prototxt(152, h, w), layers).forward(data=img)
    for k in out:
        assert float(np.abs(out[k] - ref[k]).max()) <= 1e-3, k
    rgb = np.random.RandomState(0).randint(0, 256, (h, w, 3)).astype(np.uint8)
    demo = gpu_caffe.Net(deepercut_prototxt(152, h, w), path, gpu_caffe.TEST, from_text=True)
    p_all = EP.estimate_pose(rgb, None, None, scales=[1.0], net=demo, all_outputs=True)
    assert demo.wanted_outputs == ["loc_pred", "next_pred", "prob"]
    p_two = EP.estimate_pose(rgb, None, None, scales=[1.0], net=demo)
    assert demo.wanted_outputs == ["loc_pred", "prob"]
    assert np.allclose(p_all, p_two, atol=1e-4), float(np.abs(p_all - p_two).max())
