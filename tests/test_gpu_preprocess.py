"""-m gpu: dc_net_forward_images — the demo's pre-processing on the device.

The canvas the HIP kernels leave in the `data` blob must equal the oracle's (oracle/preprocess.py, itself pinned to
Pillow) BIT FOR BIT: the resample is integer arithmetic.  The maps and the decoded pose of the image entry must be
those of the classic entry (`net.forward()` on the host-built canvas + pose_from_maps)."""
import os

import numpy as np
import pytest

from oracle import preprocess as OP
from test_gpu_tiling import _fill, local_fcn_prototxt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "preprocess_golden.npz"))


@pytest.fixture(scope="module")
def small_net(gpu_caffe):
    net = gpu_caffe.Net(local_fcn_prototxt(64, 64), gpu_caffe.TEST, from_text=True)
    _fill(net, 5)
    return net


@pytest.fixture(scope="module")
def small_net_f16(gpu_caffe):
    net = gpu_caffe.Net(local_fcn_prototxt(64, 64), gpu_caffe.TEST, from_text=True, dtype="f16")
    _fill(net, 5)
    return net


@pytest.mark.parametrize("i", range(int(G["n"])))
def test_device_canvas_equals_pillow_golden(small_net, i):
    img, scale = G["image_%d" % i], float(G["scale_%d" % i])
    small_net.forward_images(img, scale, want=(), pose=False)
    got = small_net.blobs["data"].data[0].transpose(1, 2, 0)
    assert np.array_equal(got, G["canvas_%d" % i])


@pytest.mark.parametrize("hw,scale", [((240, 320), 1.0), ((240, 320), 0.5), ((240, 320), 0.75), ((240, 320), 1.25),
                                      ((336, 256), 0.6180339887), ((97, 131), 1.9), ((97, 131), 0.11), ((9, 300), 1.0)])
def test_device_canvas_equals_oracle_batched(small_net, hw, scale):
    rs = np.random.RandomState(int(hw[0] * 7 + scale * 100))
    imgs = rs.randint(0, 256, (3,) + hw + (3,)).astype(np.uint8)
    imgs[1, : hw[0] // 2] = 255
    imgs[2, :, : hw[1] // 2] = 0
    small_net.forward_images(imgs, scale, want=(), pose=False)
    got = small_net.blobs["data"].data
    for b in range(3):
        want = OP.preprocess(imgs[b], scale)
        assert np.array_equal(got[b].transpose(1, 2, 0), want), "image %d" % b


def test_fp16_canvas_is_exact_too(small_net_f16):
    # |pixel - mean| <= 151: integers, exactly representable in float16
    img = np.random.RandomState(3).randint(0, 256, (120, 90, 3)).astype(np.uint8)
    for scale in (1.0, 0.7):
        small_net_f16.forward_images(img, scale, want=(), pose=False)
        assert np.array_equal(small_net_f16.blobs["data"].data[0].transpose(1, 2, 0), OP.preprocess(img, scale))


@pytest.mark.parametrize("scale", [1.0, 0.75, 1.25])
def test_image_entry_equals_classic_entry(small_net, scale):
    from pose.estimate_pose import forward_maps, pose_from_maps

    img = np.random.RandomState(21).randint(0, 256, (200, 264, 3)).astype(np.uint8)
    out = small_net.forward_images(img, scale, want=("prob", "loc_pred"), pose=True)
    prob, loc = forward_maps(small_net, OP.preprocess(img, scale))  # host canvas -> net.forward()
    assert np.array_equal(out["prob"][0], prob) and np.array_equal(out["loc_pred"][0], loc)
    want = pose_from_maps(prob, loc, scale)
    assert np.allclose(out["pose"][0], want, rtol=0, atol=1e-9)


def test_image_entry_on_the_full_network(gpu_caffe, synth152):
    from deepcut_tools import deepercut_prototxt
    from pose.estimate_pose import forward_maps, pose_from_maps

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True)
    imgs = np.random.RandomState(8).randint(0, 256, (2, 150, 210, 3)).astype(np.uint8)
    out = net.forward_images(imgs, 0.75, want=("prob", "loc_pred", "next_pred"), pose=True)
    assert out["prob"].shape == (2, 14, 15, 20) and out["next_pred"].shape[1] == 364
    for b in range(2):
        prob, loc = forward_maps(net, OP.preprocess(imgs[b], 0.75))
        # batch 2 vs batch 1 may pick different tile variants: same tolerance as the batch tests
        assert float(np.abs(out["prob"][b] - prob).max()) <= 1e-5
        assert float(np.abs(out["loc_pred"][b] - loc).max()) <= 1e-4 * max(1.0, float(np.abs(loc).max()))
        assert np.allclose(out["pose"][b], pose_from_maps(out["prob"][b], out["loc_pred"][b], 0.75), rtol=0, atol=1e-9)


def test_estimate_pose_device_and_host_routes_agree(small_net):
    from pose import estimate_pose as ep

    img = np.random.RandomState(33).randint(0, 256, (180, 240, 3)).astype(np.uint8)
    scales = [0.5, 1.0, 1.3]
    on_host = ep.estimate_pose(img, None, None, scales, net=small_net, on_device=False)
    # the reference's loop on the device (one forward per scale): the same forwards as the host route, decoded on the GPU
    looped = ep.estimate_pose(img, None, None, scales, net=small_net, grouped=False)
    assert (looped is None) == (on_host is None)
    if looped is not None:
        assert np.allclose(looped, on_host, rtol=0, atol=1e-9)
    # every scale individually, so that the comparison does not hinge on which one wins
    host_maps = {}
    for s in scales:
        a = small_net.forward_images(img, s, want=(), pose=True)["pose"][0]
        host_maps[s] = ep.forward_maps(small_net, ep.preprocess(img, s))
        b = ep.pose_from_maps(*host_maps[s], scale=s)
        assert np.allclose(a, b, rtol=0, atol=1e-9)
    # the default: several scales as ONE grouped forward (caffe.NetGroup), whose tiles may sum in another order than the single
    # net's.  Scale by scale: maps within 1e-4 of the single net's, and a joint whose arg-max cell moved must have moved to a
    # cell that TIES with the single net's maximum to that tolerance (anything else is a wrong cell, not a near-tie)
    grp = ep._scale_group(small_net, len(scales))
    outs = grp.forward_images(img, list(scales), want=("prob", "loc_pred"), pose=True)
    for s, o in zip(scales, outs):
        prob, loc = host_maps[s]
        assert float(np.abs(o["prob"][0] - prob).max()) <= 1e-4
        assert float(np.abs(o["loc_pred"][0] - loc).max()) <= 1e-4 * max(1.0, float(np.abs(loc).max()))
        pose = o["pose"][0]
        ref = ep.pose_from_maps(prob, loc, scale=s)
        rows, cols = ep.pose_cells(pose, s)
        rrows, rcols = ep.pose_cells(ref, s)
        jj = np.arange(prob.shape[0])
        assert ((rows >= 0) & (rows < prob.shape[1]) & (cols >= 0) & (cols < prob.shape[2])).all()
        assert (prob[jj, rows, cols] >= prob.reshape(len(jj), -1).max(axis=1) - 1e-4).all()
        assert np.allclose(pose[2], ref[2], rtol=0, atol=1e-4)
        same = (rows == rrows) & (cols == rcols)
        assert np.abs(pose[:, same] - ref[:, same]).max(initial=0.0) <= 1e-2
    on_dev = ep.estimate_pose(img, None, None, scales, net=small_net)
    assert (on_dev is None) == (on_host is None)
    if on_dev is not None:
        assert np.allclose(on_dev[2].min(), on_host[2].min(), rtol=0, atol=1e-4)
    # the groups kept with the net are bounded and share their clones
    for n in (2, 3, 2, 3):
        ep._scale_group(small_net, n)
    assert len(small_net.__dict__["_scale_clones"]) == 2 and len(small_net.__dict__["_scale_groups"]) <= ep._MAX_SCALE_GROUPS


def test_bad_arguments_are_refused(small_net, gpu_caffe):
    img = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(gpu_caffe.DeepcutError):
        small_net.forward_images(img, 0.0)
    with pytest.raises(ValueError):
        small_net.forward_images(np.zeros((16, 16), np.uint8))
    with pytest.raises(gpu_caffe.DeepcutError):
        small_net.forward_images(img, 0.001)  # nothing left of the image
