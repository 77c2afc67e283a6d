"""CPU: host logic of the graph runtime — Net::Init semantics, shape inference, weight I/O, blob/
SyncedMemory semantics, lowering decisions — none of which needs a GPU."""
import os

import numpy as np
import pytest

import caffe
from deepcut_tools import (deepercut_layer_table, deepercut_prototxt, read_caffemodel, synth_weights,
                           write_caffemodel)


@pytest.fixture(scope="module")
def net152():
    return caffe.Net(deepercut_prototxt(152, 240, 320), caffe.TEST, from_text=True)


def test_insert_splits_census(net152):
    # 680 layers + 54 Split layers = 734; 220 named blobs + 112 split tops
    names, types = net152._layer_names, net152.layer_types
    assert len(names) == 734
    assert types.count("Split") == 54
    assert len(net152.blobs) == 332
    # SplitBlobName (insert_splits.cpp:135-141)
    assert "pool1_pool1_0_split_0" in net152.blobs and "pool1_pool1_0_split_1" in net152.blobs
    assert "res2a_res2a_relu_0_split_1" in net152.blobs
    assert sum(1 for b in net152.blobs if b.startswith("res3b7_res3b7_relu_0_split_")) == 5
    assert sum(1 for b in net152.blobs if b.startswith("res5c_res5c_relu_0_split_")) == 3


def test_inputs_outputs(net152):
    assert net152.inputs == ["data"]
    assert net152.outputs == ["loc_pred", "next_pred", "prob"]  # std::set order (net.cpp:268-273)
    assert net152.name == "ResNet-152"


@pytest.mark.parametrize("hw,exp", [((240, 320), (30, 40)), ((544, 736), (68, 92)), ((688, 688), (86, 86)),
                                    ((104, 136), (13, 17)), ((64, 64), (8, 8)), ((72, 200), (9, 25))])
def test_shape_inference_follows_reshape(net152, hw, exp):
    net152.blobs["data"].reshape(1, 3, *hw)
    net152.reshape()
    assert net152.blobs["prob"].shape == (1, 14) + exp
    assert net152.blobs["loc_pred"].shape == (1, 28) + exp
    assert net152.blobs["next_pred"].shape == (1, 364) + exp
    h, w = hw
    c1 = ((h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1)
    assert net152.blobs["conv1"].shape == (1, 64) + c1
    p1 = tuple(int(np.ceil((d - 3) / 2.0)) + 1 for d in c1)
    assert net152.blobs["pool1"].shape == (1, 64) + p1
    r5 = net152.blobs["res5c"].shape
    assert net152.blobs["res5c_up_pose"].shape == (1, 14, 2 * r5[2] + 1, 2 * r5[3] + 1)


def test_algorithmic_flops_match_survey(net152):
    net152.blobs["data"].reshape(1, 3, 240, 320)
    assert abs(net152.flops() / 1e9 - 46.24) < 0.01
    net152.blobs["data"].reshape(1, 3, 544, 736)
    assert abs(net152.flops() / 1e9 - 241.09) < 0.01
    net152.blobs["data"].reshape(8, 3, 544, 736)
    assert abs(net152.flops() / 1e9 - 1928.7) < 0.1
    net152.blobs["data"].reshape(1, 3, 688, 688)
    assert abs(net152.flops() / 1e9 - 285.02) < 0.01


def test_lowering_fuses_the_whole_graph_into_gemm_launches(net152):
    net152.blobs["data"].reshape(1, 3, 240, 320)
    net152.set_option(1, 1)
    lines = [l for l in net152.plan_text().splitlines() if not l.startswith("#")]
    # 158 convolutions + 3 deconvolutions (the 4 output-parity classes of each are ONE multi-class launch) + pool + sigmoid
    assert len(lines) == 163
    assert sum("conv_gemm<" in l for l in lines) == 161
    assert sum("+resid" in l for l in lines) == 50 + 3
    assert any(l.endswith("conv1+bn_conv1+scale_conv1+conv1_relu") for l in lines)
    assert any("res2a_branch2c+bn2a_branch2c+scale2a_branch2c+res2a+res2a_relu" in l for l in lines)
    assert any("res5c_up_next+crop_next+next_pred [4 classes]" in l and "classes=4" in l and "K=8192 taps=4" in l for l in lines)
    # level 2: the three sibling heads run as one 406-channel skip GEMM + one 406-channel deconvolution
    net152.set_option(1, 2)
    lines2 = [l for l in net152.plan_text().splitlines() if not l.startswith("#")]
    assert len(lines2) == 163 - 2 - 2 - 1 == 158  # 3 skip convs -> 1, 3 deconvolutions -> 1, sigmoid folded
    assert sum("N=406" in l for l in lines2) == 2 and sum("+sigmoid" in l for l in lines2) == 1
    assert any(l.endswith("res3d_pose+res3d_locref+res3d_next") for l in lines2)
    assert abs(net152.flops() / 1e9 - 46.24) < 0.01  # algorithmic FLOPs do not depend on the fusion level
    net152.set_option(1, 0)
    lines0 = [l for l in net152.plan_text().splitlines() if not l.startswith("#")]
    # unfused: + 53 eltwise, 3 crop; every Caffe-visible blob materialised
    assert len(lines0) == 163 + 50 + 3 + 3
    assert sum("eltwise" in l for l in lines0) == 54
    assert sum("crop" in l.split("\t")[1] for l in lines0) == 3
    net152.set_option(1, 2)


def test_deconvolution_classes_as_separate_launches(monkeypatch):
    # DC_DECONV_MERGE=0 keeps the one-launch-per-parity-class lowering (what a tile without a multi-class instantiation gets)
    monkeypatch.setenv("DC_DECONV_MERGE", "0")
    net = caffe.Net(deepercut_prototxt(152, 240, 320), caffe.TEST, from_text=True)
    lines = [l for l in net.plan_text().splitlines() if not l.startswith("#")]
    assert len(lines) == 161 and sum("N=406" in l for l in lines) == 5 and sum("+sigmoid" in l for l in lines) == 4
    # k3 s2 p0: even output rows/columns receive kernel taps 0 and 2, odd ones tap 1 only
    assert any("[class 0,0]" in l and "K=8192 taps=4" in l for l in lines)
    assert any("[class 1,1]" in l and "K=2048 taps=1" in l for l in lines)


def test_forward_without_gpu_fails_loudly(net152):
    caffe.set_mode_cpu()
    with pytest.raises(caffe.DeepcutError) as e:
        net152.forward()
    assert e.value.code == -6 and "CPU mode" in str(e.value)
    if caffe.device_count() == 0:
        caffe.set_mode_gpu()
        with pytest.raises(caffe.DeepcutError) as e:
            net152.forward()
        assert e.value.code == -5
        caffe.set_mode_cpu()


def test_blob_data_is_a_writable_view_that_outlives_the_net():
    # python/caffe/test/test_net.py:48-60
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    d = net.blobs["data"].data
    assert d.dtype == np.float32 and d.shape == (1, 3, 64, 64) and d.flags.c_contiguous and d.flags.writeable
    assert (d == 0).all()  # first touch zero-fills (syncedmem.cpp:25-31)
    d[0, 1, 2, 3] = 7.0
    assert net.blobs["data"].data[0, 1, 2, 3] == 7.0  # same memory
    assert net.blobs["data"].head == 1  # HEAD_AT_CPU
    p = net.params["conv1"][0].data
    del net
    import gc

    gc.collect()
    p[...] = 1.0
    d[...] = 2.0
    assert p.sum() == p.size and d.sum() == 2 * d.size


def test_blob_reshape_grows_only_and_legacy_accessors():
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    b = net.blobs["data"]
    assert (b.num, b.channels, b.height, b.width, b.count) == (1, 3, 64, 64, 3 * 64 * 64)
    b.data[...] = 3.0
    b.reshape(1, 3, 32, 32)  # shrink keeps the memory (blob.cpp:37-41)
    assert b.data.shape == (1, 3, 32, 32) and (b.data == 3.0).all()
    b.reshape(2, 3, 64, 64)  # growth replaces it
    assert b.data.shape == (2, 3, 64, 64) and (b.data == 0).all()
    w = net.params["bn_conv1"][2]
    assert w.shape == (1,) and (w.num, w.channels, w.height, w.width) == (1, 1, 1, 1)


def test_param_shapes_and_default_fillers():
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    P = net.params
    assert P["conv1"][0].shape == (64, 3, 7, 7) and len(P["conv1"]) == 1
    assert P["res5a_branch2b"][0].shape == (512, 512, 3, 3)
    assert P["res5c_up_next"][0].shape == (2048, 364, 3, 3) and P["res5c_up_next"][1].shape == (364,)
    assert [b.shape for b in P["bn2a_branch1"]] == [(256,), (256,), (1,)]
    assert [b.shape for b in P["scale2a_branch1"]] == [(256,), (256,)]
    assert (P["scale2a_branch1"][0].data == 1).all() and (P["scale2a_branch1"][1].data == 0).all()
    assert len(P) == 158 + 3 + 155 + 155
    total = sum(int(np.prod(b.shape)) for n in P for b in P[n] if net.layer_types[net._layer_names.index(n)] in ("Convolution", "Deconvolution") for b in [b][:1])
    assert abs(total / 1e6 - 65.7) < 0.1  # conv+deconv weights incl. biases, SURVEY §8a7


def test_copy_from_by_name_with_checks(tmp_path):
    small = "\n".join(deepercut_prototxt(152, 64, 64).splitlines()[:12]) + "\n"  # conv1..pool1 + res2a_branch1 pieces
    net = caffe.Net(small, caffe.TEST, from_text=True)
    rs = np.random.RandomState(0)
    w = rs.randn(64, 3, 7, 7).astype(np.float32)
    good = str(tmp_path / "good.caffemodel")
    write_caffemodel(good, "x", [("conv1", "Convolution", [w]), ("not_in_net", "Convolution", [w])])
    net.copy_from(good)  # unmatched source layers are ignored (net.cpp:815-818)
    assert np.array_equal(net.params["conv1"][0].data, w)
    bad_count = str(tmp_path / "bad_count.caffemodel")
    write_caffemodel(bad_count, "x", [("conv1", "Convolution", [w, np.zeros(64, np.float32)])])
    with pytest.raises(caffe.DeepcutError) as e:
        net.copy_from(bad_count)
    assert e.value.code == -3 and "Incompatible number of blobs" in str(e.value)
    bad_shape = str(tmp_path / "bad_shape.caffemodel")
    write_caffemodel(bad_shape, "x", [("conv1", "Convolution", [np.zeros((64, 3, 5, 5), np.float32)])])
    with pytest.raises(caffe.DeepcutError) as e:
        net.copy_from(bad_shape)
    assert "shape mismatch" in str(e.value)
    with pytest.raises(caffe.DeepcutError) as e:
        net.copy_from(str(tmp_path / "missing.caffemodel"))
    assert e.value.code == -2 and "Could not open file" in str(e.value)


def test_save_reload_roundtrip(tmp_path):
    # python/caffe/test/test_net.py:62-81
    small = "\n".join(deepercut_prototxt(152, 64, 64).splitlines()[:12]) + "\n"
    net = caffe.Net(small, caffe.TEST, from_text=True)
    rs = np.random.RandomState(1)
    for name in net.params:
        for b in net.params[name]:
            b.data[...] = rs.randn(*b.shape).astype(np.float32)
    path = str(tmp_path / "saved.caffemodel")
    net.save(path)
    net2 = caffe.Net(small, caffe.TEST, from_text=True)
    net2.copy_from(path)
    for name in net.params:
        for a, b in zip(net.params[name], net2.params[name]):
            assert np.array_equal(a.data, b.data)
    # the C++ writer's file is readable by the independent Python reader and vice versa
    nm, layers = read_caffemodel(path)
    assert nm == "ResNet-152"
    got = {n: bl for n, _t, bl in layers if bl}
    assert np.array_equal(got["conv1"][0], net.params["conv1"][0].data)
    assert got["bn_conv1"][2].shape == (1,)


def test_constructor_errors(tmp_path):
    with pytest.raises(caffe.DeepcutError) as e:
        caffe.Net(str(tmp_path / "nope.prototxt"), caffe.TEST)
    assert "Could not open file" in str(e.value)
    bad = 'name: "x" input: "data" input_dim: 1 input_dim: 3 input_dim: 8 input_dim: 8\n'
    with pytest.raises(caffe.DeepcutError) as e:
        caffe.Net(bad + 'layer { name: "l" type: "LRN" bottom: "data" top: "o" }', caffe.TEST, from_text=True)
    assert e.value.code == -4 and "outside the DeeperCut forward path" in str(e.value)
    with pytest.raises(caffe.DeepcutError) as e:
        caffe.Net(bad + 'layer { name: "l" type: "ReLU" bottom: "nope" top: "o" }', caffe.TEST, from_text=True)
    assert "Unknown bottom blob 'nope'" in str(e.value)
    with pytest.raises(caffe.DeepcutError) as e:
        caffe.Net(bad + 'layer { name: "a" type: "ReLU" bottom: "data" top: "o" }'
                        'layer { name: "b" type: "Sigmoid" bottom: "data" top: "o" }', caffe.TEST, from_text=True)
    assert "produced by multiple sources" in str(e.value)
    with pytest.raises(caffe.DeepcutError):
        caffe.Net(bad + "layer { name: ", caffe.TEST, from_text=True)
    # Crop needs a strictly larger bottom[0] (crop_layer.cpp:30-32)
    with pytest.raises(caffe.DeepcutError) as e:
        caffe.Net('input: "a" input_dim: 1 input_dim: 4 input_dim: 8 input_dim: 8 '
                  'input: "b" input_dim: 1 input_dim: 4 input_dim: 8 input_dim: 7 '
                  'layer { name: "c" type: "Crop" bottom: "a" bottom: "b" top: "o" }', caffe.TEST, from_text=True)
    assert e.value.code == -3 and "invalid offset" in str(e.value)


def test_phase_filtering_and_text_format_details():
    txt = '''name: 'q'  # single quotes and comments
    input: "data" input_dim: 1 input_dim: 32 input_dim: 9 input_dim: 9
    layer { name: "drop_me" type: "Dropout" bottom: "data" top: "data" include { phase: TRAIN } }
    layer { name: "c" type: "Convolution" bottom: "data" top: "c"
            convolution_param { num_output: 8 kernel_h: 3 kernel_w: 1 stride_h: 2 stride_w: 1 pad_h: 1 pad_w: 0 } }
    layer { type: "ReLU" name: "r" bottom: "c" top: 'c' }'''
    net = caffe.Net(txt, caffe.TEST, from_text=True)
    assert net._layer_names == ["c", "r"] and net.name == "q"
    assert net.blobs["c"].shape == (1, 8, 5, 9)
    assert net.params["c"][0].shape == (8, 32, 3, 1) and net.params["c"][1].shape == (8,)


def test_resnet101_variant_builds():
    net = caffe.Net(deepercut_prototxt(101, 64, 64), caffe.TEST, from_text=True)
    t = deepercut_layer_table(101)
    assert sum(l["type"] == "Convolution" for l in t) == 104 + 3
    assert "res4b22" in net.blobs and "res3b3" in net.blobs and net.outputs == ["loc_pred", "next_pred", "prob"]


def test_clone_shares_parameters_and_plan():
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    net.params["conv1"][0].data[...] = 0.5
    net.blobs["data"].reshape(2, 3, 72, 104)
    c = net.clone()
    assert c.blobs["data"].shape == (2, 3, 72, 104) and c.blobs["prob"].shape == (2, 14, 9, 13)
    assert c._layer_names == net._layer_names and list(c.blobs) == list(net.blobs)
    assert (c.params["conv1"][0].data == 0.5).all()
    net.params["conv1"][0].data[0, 0, 0, 0] = 7.0  # same host memory
    assert c.params["conv1"][0].data[0, 0, 0, 0] == 7.0
    assert c.plan_text() == net.plan_text()
    c.blobs["data"].data[...] = 1.0  # activations are private
    assert (net.blobs["data"].data == 0).all()
    del net
    import gc

    gc.collect()
    assert c.params["conv1"][0].data[0, 0, 0, 0] == 7.0  # shared blobs outlive the parent


def test_fp16_lowering_on_the_host():
    net = caffe.Net(deepercut_prototxt(152, 240, 320), caffe.TEST, from_text=True, dtype="f16")
    text = net.plan_text()
    lines = [l for l in text.splitlines() if not l.startswith("#")]
    assert "dtype=f16" in text.splitlines()[0] and len(lines) == 158
    half_tiles = set(n for n, es in caffe.conv_variants() if es == 2)  # "h..." register-ring tiles and "d..." LDS-DMA tiles
    assert all(l.split("conv_gemm<")[1].split(">")[0] in half_tiles for l in lines if "conv_gemm" in l)
    assert "K=448 taps=7" in lines[0]  # stem: 7 row taps of 8 pixels x 8 channels (3 padded to one 16-byte vector)
    assert abs(net.flops() / 1e9 - 46.24) < 0.01
    net.set_option(3, 0)
    assert "dtype=f32" in net.plan_text() and "K=224 taps=7" in net.plan_text()
    with pytest.raises(caffe.DeepcutError):
        net.set_option(3, 7)


def test_plan_cache_serves_shapes_met_before_without_relowering():
    # Layer::Forward re-derives shapes on every call (layer.hpp:451-456); the demo changes the shape once per scale
    # (estimate_pose.py:81-128): the four pyramid shapes are lowered once each, then served from the cache
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    shapes = [(1, 3, 272, 368), (1, 3, 408, 552), (1, 3, 544, 736), (1, 3, 680, 920)]
    texts = {}
    for s in shapes:
        net.blobs["data"].reshape(*s)
        texts[s] = net.plan_text()
    st = net.stats()
    assert st["lowerings"] == 4 and st["cached_plans"] == 4 and st["plan_hits"] == 0
    for _ in range(3):
        for s in shapes:
            net.blobs["data"].reshape(*s)
            assert net.plan_text() == texts[s]
            assert abs(net.flops() - float(texts[s].split(" launches, ")[1].split(" GFLOP")[0]) * 1e9) < 1e6 * net.flops() / 1e9
    st = net.stats()
    assert st["lowerings"] == 4 and st["plan_hits"] == 12 and st["repacks"] == 1
    # the views / elided blobs of a plan are restored with it
    net.blobs["data"].reshape(1, 3, 272, 368)
    net.plan_text()
    assert net.blobs["prob"].shape == (1, 14, 34, 46)
    # an option that changes the lowering drops every cached plan
    net.set_option(1, 0)
    assert len([l for l in net.plan_text().splitlines() if not l.startswith("#")]) > 200
    assert net.stats()["cached_plans"] == 1


def test_plan_cache_is_bounded_lru(monkeypatch):
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    for k in range(20):
        net.blobs["data"].reshape(1, 3, 64 + 8 * k, 64)
        net.plan_text()
    assert net.stats()["cached_plans"] == 17  # the active plan + DC_PLAN_CACHE (16) parked ones
    net.blobs["data"].reshape(1, 3, 64 + 8 * 19, 64)
    net.plan_text()
    net.blobs["data"].reshape(1, 3, 64 + 8 * 18, 64)
    net.plan_text()
    assert net.stats()["lowerings"] == 20  # the two most recent shapes were still cached
    net.blobs["data"].reshape(1, 3, 64, 64)  # the oldest was evicted
    net.plan_text()
    assert net.stats()["lowerings"] == 21


def test_reading_a_parameter_does_not_repack_but_writing_does():
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    net.plan_text()
    assert net.stats()["repacks"] == 1
    _ = net.params["conv1"][0].data.shape  # pycaffe's .data is mutable_cpu_data: an access alone must not cost a re-pack
    _ = [l.blobs for l in net.layers]
    float(net.params["bn_conv1"][0].data.sum())
    net.plan_text()
    assert net.stats()["repacks"] == 1 and net.stats()["lowerings"] == 1
    net.params["conv1"][0].data[0, 0, 0, 0] = 3.0  # a content change
    net.plan_text()
    assert net.stats()["repacks"] == 2 and net.stats()["lowerings"] == 2
    net.params["conv1"][0].data[0, 0, 0, 0] = 3.0  # same bytes again
    net.plan_text()
    assert net.stats()["repacks"] == 2


def test_parameter_write_reaches_every_executor_and_survives_the_parent():
    import gc

    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    c = net.clone()
    net.plan_text()
    c.plan_text()
    assert c.stats()["lowerings"] == 1 and c.stats()["repacks"] == 0  # the clone found the images its parent packed
    net.params["conv1"][0].data[...] = 0.25  # written through the PARENT ...
    c.plan_text()                            # ... the clone re-lowers from the new generation
    assert c.stats()["lowerings"] == 2 and c.stats()["repacks"] == 1
    net.plan_text()
    assert net.stats()["lowerings"] == 2 and net.stats()["repacks"] == 1  # packed once for both
    w = c.params["conv1"][0]
    del net
    gc.collect()
    w.data[...] = 0.5  # the blob's model state is owned jointly: no dangling owner after the parent is gone
    c.plan_text()
    assert c.stats()["lowerings"] == 3
    c2 = c.clone()
    assert (c2.params["conv1"][0].data == 0.5).all()


def test_head_channel_split_switch(monkeypatch):
    """DC_HEAD_SPLIT=1: the two 406-channel head GEMMs as [ch 0-383] + [ch 384-405] (measured: no gain, off by default)."""
    monkeypatch.setenv("DC_HEAD_SPLIT", "1")
    net = caffe.Net(deepercut_prototxt(152, 240, 320), caffe.TEST, from_text=True, dtype="f16")
    lines = [l for l in net.plan_text().splitlines() if not l.startswith("#")]
    assert len(lines) == 160
    assert sum(1 for l in lines if "[ch 0-383]" in l) == 2 and sum(1 for l in lines if "[ch 384-405]" in l) == 2
    assert "N=384" in [l for l in lines if "[ch 0-383]" in l][0] and "N=22" in [l for l in lines if "[ch 384-405]" in l][0]
