"""-m gpu: the whole DeeperCut ResNet-152 forward on MI355X (through the C-ABI / pycaffe shim) against
the CPU oracle on the same seeded input and the same synthetic weights.  Bound: 1e-3 max-abs on every
output map (BASELINE.json north_star), fp32."""
import os

import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _oracle(proto, layers, img):
    from oracle import oracle as O

    O.set_threads(min(16, os.cpu_count() or 1))  # the GPU box grants 16 CPUs of cgroup quota
    return O.OracleNet(proto, layers).forward(data=img)


@pytest.mark.parametrize("hw", [(64, 64), (104, 136), (72, 200)])
def test_unfused_every_blob_matches_oracle(gpu_caffe, synth152, hw):
    """DC_OPT_FUSE 0 materialises every Caffe-visible blob: compare ALL of them (in-place chains hold
    their final value on both sides)."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, fuse=0)
    img = rand_image(1, h, w)
    net.blobs["data"].data[...] = img
    net.forward()
    ref = _oracle(proto, layers, img)
    worst = ("", 0.0)
    for name, r in ref.items():
        got = net.blobs[name].data
        assert got.shape == r.shape, name
        scale = max(1.0, float(np.abs(r).max()))
        err = float(np.abs(got - r).max()) / scale
        if err > worst[1]:
            worst = (name, err)
        assert err <= TOL, "%s: rel-to-range err %g" % (name, err)
    for name in ("prob", "loc_pred", "next_pred"):
        assert float(np.abs(net.blobs[name].data - ref[name]).max()) <= TOL
    print("worst blob", worst)


@pytest.mark.parametrize("fuse", [1, 2])
@pytest.mark.parametrize("hw", [(64, 64), (104, 136), (240, 320)])
def test_fused_outputs_match_oracle(gpu_caffe, synth152, hw, fuse):
    """fuse 1: residual + head fusion; fuse 2 (default): + the three heads as one concatenated GEMM whose
    outputs are channel views (prob carries the folded sigmoid)."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, fuse=fuse)
    img = rand_image(2, h, w)
    net.blobs["data"].data[...] = img
    out = net.forward()
    ref = _oracle(proto, layers, img)
    assert sorted(out) == ["loc_pred", "next_pred", "prob"]
    for k in out:
        assert out[k].shape == ref[k].shape
        err = float(np.abs(out[k] - ref[k]).max())
        print(k, out[k].shape, "max abs err", err, "range", float(np.abs(ref[k]).max()))
        assert err <= TOL, k


def test_elided_blobs_raise_instead_of_returning_stale_memory(gpu_caffe, synth152):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True)
    net.blobs["data"].data[...] = rand_image(3, 64, 64)
    net.forward()
    for name in ("res2a_branch2c", "fc_pose", "res5c_up_next", "res3d_locref"):
        with pytest.raises(gpu_caffe.DeepcutError) as e:
            net.blobs[name].data
        assert "DC_OPT_FUSE 0" in str(e.value)
    assert net.blobs["res5c"].data.shape == (1, 2048, 4, 4)  # still materialised


def test_reshape_between_forwards_and_determinism(gpu_caffe, synth152):
    """Layer::Forward re-runs Reshape (layer.hpp:451-456; test_net.cpp:2262-2332 TestReshape): a new input
    size needs no net.reshape(); going back to the first size reproduces the first result bit for bit."""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True)
    a = rand_image(4, 64, 64)
    b = rand_image(5, 88, 120)
    net.blobs["data"].data[...] = a
    o1 = {k: v.copy() for k, v in net.forward().items()}
    net.blobs["data"].reshape(1, 3, 88, 120)
    net.blobs["data"].data[...] = b
    o2 = net.forward()
    assert o2["prob"].shape == (1, 14, 11, 15)
    net.blobs["data"].reshape(1, 3, 64, 64)
    net.blobs["data"].data[...] = a
    o3 = net.forward()
    for k in o1:
        assert np.array_equal(o1[k], o3[k]), k


def test_batched_forward_equals_single_images(gpu_caffe, synth152):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 72, 104), path, gpu_caffe.TEST, from_text=True)
    imgs = rand_image(6, 72, 104, n=3)
    batched = net.forward_batch(imgs)
    for i in range(3):
        single = net.forward_batch(imgs[i:i + 1])
        for k in batched:
            assert np.abs(batched[k][i] - single[k][0]).max() <= 1e-5, k


def test_pose_demo_pipeline_config0(gpu_caffe, synth152):
    """BASELINE configs[0]: one 320x240 uint8 image through pre-processing -> net.forward() -> pose decode,
    against the same pipeline run on the CPU oracle's maps (single tile, scale 1)."""
    from deepcut_tools import deepercut_prototxt
    from pose import estimate_pose as ep

    path, layers = synth152
    img = np.random.RandomState(0).randint(0, 256, (240, 320, 3)).astype(np.uint8)
    net = gpu_caffe.Net(deepercut_prototxt(152, 240, 320), path, gpu_caffe.TEST, from_text=True)
    pose = ep.estimate_pose(img, None, None, [1.0], net=net)
    x = ep.preprocess(img, 1.0)
    ref = _oracle(deepercut_prototxt(152, 240, 320), layers, x.transpose(2, 0, 1)[None])
    assert float(np.abs(net.blobs["prob"].data - ref["prob"]).max()) <= TOL
    assert float(np.abs(net.blobs["loc_pred"].data - ref["loc_pred"]).max()) <= TOL
    ref_pose = ep.select_best([ep.pose_from_maps(ref["prob"][0], ref["loc_pred"][0], 1.0)])
    assert pose is not None and pose.shape == (5, 14) and np.isfinite(pose).all()
    # same arg-max cells, or a cell that ties with the oracle's maximum within the tolerance (proved on the oracle's own map:
    # a joint decoded from any other cell is a wrong cell, however few there are); positions agree to sqrt(53)*1e-3 px
    rows, cols = ep.pose_cells(pose, 1.0)
    rrows, rcols = ep.pose_cells(ref_pose, 1.0)
    rp = ref["prob"][0]
    jj = np.arange(rp.shape[0])
    assert ((rows >= 0) & (rows < rp.shape[1]) & (cols >= 0) & (cols < rp.shape[2])).all()
    assert (rp[jj, rows, cols] >= rp.reshape(len(jj), -1).max(axis=1) - TOL).all()
    assert np.allclose(pose[2], ref_pose[2], rtol=0, atol=TOL)
    same = (rows == rrows) & (cols == rcols)
    assert np.abs(pose[:, same] - ref_pose[:, same]).max(initial=0.0) <= 1e-2


@pytest.mark.parametrize("fuse", [0, 2])
def test_device_pose_decode_equals_host_decode(gpu_caffe, synth152, fuse):
    """dc_net_decode_pose (arg-max + refinement on the GPU) == pose_from_maps on the downloaded maps, which
    is pinned to the reference's _pose_from_mats by tests/test_pose.py; channel views (fuse 2) and plain
    tensors (fuse 0) alike; batch of 2."""
    from deepcut_tools import deepercut_prototxt
    from pose import estimate_pose as ep

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 104, 136, 2), path, gpu_caffe.TEST, from_text=True, fuse=fuse)
    net.blobs["data"].data[...] = rand_image(8, 104, 136, n=2)
    net.forward()
    for scale in (1.0, 0.75):
        got = net.decode_pose(scale)
        assert got.shape == (2, 5, 14)
        for i in range(2):
            ref = ep.pose_from_maps(net.blobs["prob"].data[i], net.blobs["loc_pred"].data[i], scale)
            assert np.allclose(got[i], ref, rtol=0, atol=1e-9)


def test_benchmark_config_fullsize_matches_oracle(gpu_caffe, synth152):
    """BASELINE configs[1] itself: 1x3x544x736, fp32, all fusion on — every output map within 1e-3 of the CPU
    oracle (input RandomState(1).randn*50)."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    proto = deepercut_prototxt(152, 544, 736)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True)
    img = (np.random.RandomState(1).randn(1, 3, 544, 736) * 50).astype(np.float32)
    net.blobs["data"].data[...] = img
    out = net.forward()
    assert abs(net.flops() / 1e9 - 241.09) < 0.01
    ref = _oracle(proto, layers, img)
    for k in ("prob", "loc_pred", "next_pred"):
        assert out[k].shape == ref[k].shape == (1, {"prob": 14, "loc_pred": 28, "next_pred": 364}[k], 68, 92)
        err = float(np.abs(out[k] - ref[k]).max())
        print(k, "max abs err", err)
        assert err <= TOL, k


@pytest.mark.parametrize("scale_hw", [(272, 368), (408, 552), (680, 920)])
def test_pyramid_scales_batched(gpu_caffe, synth152, scale_hw):
    """BASELINE configs[2] shapes (736x544 at scales 0.5 / 0.75 / 1.25; fp32 here): a batch of 2 through one
    plan.  Checked through size-independent properties: the batch equals its images forwarded alone, a
    second forward is bit-identical, and prob stays a probability."""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    h, w = scale_hw
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w, 2), path, gpu_caffe.TEST, from_text=True)
    imgs = rand_image(11, h, w, n=2)
    a = net.forward_batch(imgs)
    b = net.forward_batch(imgs)
    one = net.forward_batch(imgs[1:2])
    for k in a:
        assert a[k].shape[2:] == (h // 8, w // 8)
        assert np.array_equal(a[k], b[k]), k
        assert np.abs(a[k][1] - one[k][0]).max() <= 1e-5, k
        assert np.isfinite(a[k]).all()
    assert (a["prob"] > 0).all() and (a["prob"] < 1).all()


def test_pipeline_of_clones_matches_sequential(gpu_caffe, synth152):
    """deepcut_tools.Pipeline: three executors (a Net + two clones sharing its packed weights) with requests in
    flight on their own streams give the same maps as one forward at a time."""
    import torch
    from deepcut_tools import Pipeline, deepercut_prototxt

    path, _ = synth152
    h, w = 104, 136
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    dev = torch.device("cuda", 0)
    imgs = [torch.from_numpy(rand_image(20 + i, h, w)).to(dev) for i in range(7)]
    ref = [net.forward_batch(im.cpu().numpy()) for im in imgs]
    pipe = Pipeline(net, depth=3)
    assert pipe.depth == 3
    outs = [[torch.empty(1, c, h // 8, w // 8, device=dev) for c in (14, 28, 364)] for _ in imgs]
    torch.cuda.synchronize()
    for i, im in enumerate(imgs):
        pipe.submit(im.data_ptr(), 1, h, w, outs[i][0].data_ptr(), outs[i][1].data_ptr(), outs[i][2].data_ptr(), tag=i)
    assert sorted(pipe.drain()) == list(range(7))  # (opportunistic coalescing by default: groups finish in any order)
    assert sum(k * v for k, v in pipe.batch_sizes.items()) == 7 and len(pipe.latencies) == 7
    for i in range(len(imgs)):
        for k, t in zip(("prob", "loc_pred", "next_pred"), outs[i]):
            assert np.abs(t.cpu().numpy() - ref[i][k]).max() <= 1e-5, (i, k)
    # cross-request batching: the same seven requests coalesced three at a time (3 + 3 + a partial batch of 1) into batch
    # forwards on two executors; every request still gets ITS maps in ITS buffers
    for o in outs:
        for t in o:
            t.zero_()
    pipe2 = Pipeline(net, depth=2, coalesce=3)
    for i, im in enumerate(imgs):
        pipe2.submit(im.data_ptr(), 1, h, w, outs[i][0].data_ptr(), outs[i][1].data_ptr(), outs[i][2].data_ptr(), tag=i)
    assert sorted(pipe2.drain()) == list(range(7))
    for i in range(len(imgs)):
        for k, t in zip(("prob", "loc_pred", "next_pred"), outs[i]):
            assert np.abs(t.cpu().numpy() - ref[i][k]).max() <= 1e-5, (i, k)


@pytest.mark.parametrize("hw", [(16, 24), (24, 16), (40, 8)])
def test_tiny_inputs(gpu_caffe, synth152, hw):
    """Edge of the shape space: res4/res5 maps of 1x2 / 2x1 / 3x1 pixels, every tile mostly padding."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = hw
    proto = deepercut_prototxt(152, h, w)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True)
    img = rand_image(9, h, w)
    net.blobs["data"].data[...] = img
    out = net.forward()
    ref = _oracle(proto, layers, img)
    for k in out:
        assert out[k].shape == ref[k].shape
        assert float(np.abs(out[k] - ref[k]).max()) <= TOL, k


def test_partial_forward_ranges_unfused(gpu_caffe, synth152):
    """Net::ForwardFromTo(start, end) with DC_OPT_FUSE 0 (net.cpp:565-581, test_net.cpp TestFromTo): run up to a
    layer, overwrite the blob on the host (HEAD_AT_CPU -> re-uploaded), continue from the next layer."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    proto = deepercut_prototxt(152, 64, 64)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, fuse=0)
    img = rand_image(10, 64, 64)
    net.blobs["data"].data[...] = img
    names = net._layer_names
    cut_idx = names.index("res2c_relu")
    net._forward(0, cut_idx)  # pycaffe's forward(end=...) needs a layer whose name is also a blob name
    mid = net.blobs["res2c"].data.copy()
    ref = _oracle(proto, layers, img)
    assert np.abs(mid - ref["res2c"]).max() <= 1e-3
    net.blobs["res2c"].data[...] = mid * 0.5  # host write: authoritative copy is now on the CPU
    net._forward(cut_idx + 1, len(names) - 1)
    out = {k: net.blobs[k].data for k in net.outputs}
    # oracle: same graph continued from the modified blob
    from oracle import oracle as O
    import re

    tail_layers = proto.split("\n")
    cut = [i for i, l in enumerate(tail_layers) if 'name: "res2c_relu"' in l][0]
    tail = 'input: "res2c" input_dim: 1 input_dim: 256 input_dim: 16 input_dim: 16\n' + "\n".join(tail_layers[cut + 1:])
    ref2 = O.OracleNet(tail, layers).forward(res2c=mid * 0.5)
    for k in ("prob", "loc_pred", "next_pred"):
        assert float(np.abs(out[k] - ref2[k]).max()) <= TOL, k
    # with fusion on, a range that cuts through a fused group is refused, not silently mis-executed
    net2 = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, fuse=2)
    net2.blobs["data"].data[...] = img
    with pytest.raises(gpu_caffe.DeepcutError) as e:
        net2._forward(0, net2._layer_names.index("bn_conv1"))
    assert "cuts through the fused group" in str(e.value)


def test_sharded_runner_on_the_hip_path(gpu_caffe, synth152):
    """deepcut_tools.ShardedPoseRunner (batching by shape + device decode + best-scale selection) == the
    reference-style sequential loop of pose.estimate_pose, image by image."""
    from deepcut_tools import ShardedPoseRunner, deepercut_prototxt
    from pose import estimate_pose as ep

    path, _ = synth152
    rs = np.random.RandomState(4)
    imgs = [rs.randint(0, 256, (h, w, 3)).astype(np.uint8) for (h, w) in [(96, 128), (96, 128), (120, 88)]]
    scales = [0.75, 1.0]
    net = gpu_caffe.Net(deepercut_prototxt(152, 96, 128), path, gpu_caffe.TEST, from_text=True)
    res = ShardedPoseRunner(net).run(imgs, scales, want_maps=True)
    assert len(res["items"]) == 6 and sorted(res["maps"]) == list(range(6))
    assert res["maps"][0]["next_pred"].shape[0] == 364
    # several batches in flight (clones, one stream each): the same poses
    piped = ShardedPoseRunner(net, max_batch=2, depth=3).run(imgs, scales)
    base = ShardedPoseRunner(net, max_batch=2, depth=1).run(imgs, scales)
    assert np.abs(piped["item_poses"] - base["item_poses"]).max() <= 1e-2 and piped["best_scale"] == base["best_scale"]
    net2 = gpu_caffe.Net(deepercut_prototxt(152, 96, 128), path, gpu_caffe.TEST, from_text=True)
    for i, im in enumerate(imgs):
        ref = ep.estimate_pose(im, None, None, scales, net=net2, on_device=False)  # Pillow + NumPy around net.forward()
        got = res["poses"][i]
        assert (ref is None) == (got is None)
        if ref is not None:
            assert np.abs(got - ref).max() <= 1e-2  # batched vs single forward: fp32 summation order only


def test_debug_info_matches_the_oracles_blobs(gpu_caffe, synth152):
    """dc_net_debug_info = Net::ForwardDebugInfo (net.cpp:648-681): the mean |x| of every Caffe-visible top blob of the last
    forward, in the reference's log format; with fusion off every one of them is a number and equals the oracle's."""
    import re

    from deepcut_tools import deepercut_prototxt
    from oracle import oracle as O

    path, layers = synth152
    proto = deepercut_prototxt(152, 64, 64)
    net = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, fuse=0)
    img = rand_image(41, 64, 64)
    net.blobs["data"].data[...] = img
    net.forward()
    text = net.debug_info()
    lines = text.splitlines()
    assert lines[0].startswith("    [Forward] Input data data: ")
    assert abs(float(lines[0].split(": ")[1]) - float(np.abs(img).mean())) < 1e-3
    O.set_threads(min(16, os.cpu_count() or 1))
    ref = O.OracleNet(proto, layers).forward(data=img)
    tops = {}
    for ln in lines:
        m = re.match(r"    \[Forward\] Layer (\S+), top blob (\S+) data: (.*)$", ln)
        if m and not m.group(3).startswith("("):
            tops[m.group(2)] = float(m.group(3))
    assert "elided" not in text
    checked = 0
    for name, val in tops.items():
        if name in ref:
            want = float(np.abs(ref[name]).mean())
            assert abs(val - want) <= 1e-4 * max(1.0, want), (name, val, want)
            checked += 1
    assert checked >= 200 and {"prob", "loc_pred", "next_pred", "res5c"} <= set(tops)
    assert sum(1 for ln in lines if ", param blob " in ln) >= 158 + 3 * 155 + 2 * 155  # conv + BN + Scale parameter blobs
    # with fusion on, swallowed blobs are reported as such (never as stale numbers)
    net2 = gpu_caffe.Net(proto, path, gpu_caffe.TEST, from_text=True, fuse=2)
    net2.blobs["data"].data[...] = img
    net2.forward()
    assert "elided by fusion" in net2.debug_info()


def test_outputs_read_through_host_pointers_are_delivered_by_the_next_forwards(gpu_caffe, synth152):
    """The drop-in sequence `blobs['data'].data[...] = x; net.forward(); blobs['prob'].data` (estimate_pose.py:104-112): an
    output downloaded on demand once travels inside the following forwards (head SYNCED when forward() returns: `.data` costs
    no kernel, copy or stream round trip); an output nobody touches is not sent; values are the same either way and the
    SyncedMemory head walk (syncedmem.cpp:25-77) stays what it was.  (pycaffe's own forward() reads every output blob into the
    dict it returns, pycaffe.py:108, so the selective part is exercised through the C-level forward.)"""
    from deepcut_tools import deepercut_prototxt

    AT_GPU, SYNCED, AT_CPU = 2, 3, 1
    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 104, 136), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    last = len(net._layer_names) - 1
    x = rand_image(12, 104, 136)
    net.blobs["data"].data[...] = x
    net._forward(0, last)
    assert net.blobs["prob"].head == AT_GPU and net.blobs["next_pred"].head == AT_GPU  # first forward: nothing was asked for yet
    first = {k: net.blobs[k].data.copy() for k in ("prob", "loc_pred")}                 # on demand ...
    assert net.blobs["prob"].head == AT_CPU                                              # ... and mutable_cpu_data marks the host copy
    for _ in range(2):
        net.blobs["data"].data[...] = x
        net._forward(0, last)
        assert net.blobs["prob"].head == SYNCED and net.blobs["loc_pred"].head == SYNCED  # delivered with the forward
        assert net.blobs["next_pred"].head == AT_GPU                                       # never read: never sent
        for k in first:
            assert np.array_equal(net.blobs[k].data, first[k]), k
    # stop reading loc_pred: one more delivery (it was touched after the last one), then none
    net.blobs["data"].data[...] = x
    net._forward(0, last)
    _ = net.blobs["prob"].data
    net.blobs["data"].data[...] = x
    net._forward(0, last)
    assert net.blobs["prob"].head == SYNCED and net.blobs["loc_pred"].head == AT_GPU
    assert np.array_equal(net.blobs["loc_pred"].data, first["loc_pred"])  # on demand again
    # pycaffe's forward(): all three maps come back in the dict, and from the second call on they were delivered
    net.blobs["data"].data[...] = x
    a = {k: v.copy() for k, v in net.forward().items()}
    net.blobs["data"].data[...] = rand_image(13, 104, 136)
    net._forward(0, last)
    assert all(net.blobs[k].head == SYNCED for k in ("prob", "loc_pred", "next_pred"))
    b = {k: net.blobs[k].data for k in a}
    assert np.array_equal(a["prob"], first["prob"]) and float(np.abs(b["prob"] - a["prob"]).max()) > 1e-6  # no stale host copy


def test_pipeline_coalesces_what_queues_up_and_equals_the_batch_forward(gpu_caffe, synth152):
    """Opportunistic cross-request batching (Pipeline's default): a request goes out alone while an executor is free; what
    queues up behind busy executors leaves as ONE batch forward of up to max_batch requests.  (a) every request gets its own
    maps whatever batch it travelled in; (b) two requests coalesced are BIT-identical to rows 0 / 1 of the batch-2 forward of
    the same two images (same plan, same tiles: dc_net_forward_requests only gathers / scatters); (c) latencies are recorded."""
    import torch
    from deepcut_tools import Pipeline, deepercut_prototxt

    path, _ = synth152
    h, w = 104, 136
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    dev = torch.device("cuda", 0)
    imgs = [torch.from_numpy(rand_image(50 + i, h, w)).to(dev) for i in range(12)]
    ref = [net.forward_batch(im.cpu().numpy()) for im in imgs]
    outs = [[torch.empty(1, c, h // 8, w // 8, device=dev) for c in (14, 28, 364)] for _ in imgs]
    pipe = Pipeline(net, depth=2, max_batch=4)
    assert pipe.opportunistic and pipe.max_queue == 8
    for rep in range(2):  # second round: every batch shape has been lowered and captured, the host floods the queue
        for o in outs:
            for t in o:
                t.zero_()
        pipe.reset_stats()
        for i, im in enumerate(imgs):
            pipe.submit(im.data_ptr(), 1, h, w, outs[i][0].data_ptr(), outs[i][1].data_ptr(), outs[i][2].data_ptr(), tag=i)
        got = [pipe.wait_one() for _ in range(4)] + pipe.drain()
        assert sorted(got) == list(range(12))
        for i in range(len(imgs)):
            for k, t in zip(("prob", "loc_pred", "next_pred"), outs[i]):
                assert np.abs(t.cpu().numpy() - ref[i][k]).max() <= 1e-5, (i, k)
    assert sum(k * v for k, v in pipe.batch_sizes.items()) == 12 and max(pipe.batch_sizes) > 1, dict(pipe.batch_sizes)
    pct = pipe.latency_percentiles((50, 99))
    assert len(pipe.latencies) == 12 and 0 < pct[50] <= pct[99]
    # (b) bit-identical to the batch forward they became
    two = Pipeline(net, depth=1, coalesce=2)
    two.nets = pipe.nets[:1]
    for i in (3, 8):
        two.submit(imgs[i].data_ptr(), 1, h, w, outs[i][0].data_ptr(), outs[i][1].data_ptr(), outs[i][2].data_ptr(), tag=i)
    assert sorted(two.drain()) == [3, 8]
    both = net.forward_batch(np.concatenate([imgs[3].cpu().numpy(), imgs[8].cpu().numpy()]))
    for row, i in enumerate((3, 8)):
        for k, t in zip(("prob", "loc_pred", "next_pred"), outs[i]):
            assert np.array_equal(t.cpu().numpy()[0], both[k][row]), (i, k)
