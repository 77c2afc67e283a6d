"""-m gpu: pyramid-grouped execution (dc_group_*, caffe.NetGroup) — several executors of one model, each at its own input
shape, as ONE launch sequence of multi-problem gather-GEMMs — against the CPU oracle, against the members' own forwards,
with every tile variant forced in turn, in float32 and float16, through the host, device and image entries."""
import os

import numpy as np
import pytest

from conftest import rand_image
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SHAPES = [(2, 40, 56), (2, 64, 64), (2, 72, 104), (2, 104, 136)]  # a small 4-"scale" pyramid, batch 2 per scale
MAX_VARIANTS = 64


def _oracle(layers, n, h, w, seed):
    from deepcut_tools import deepercut_prototxt

    O.set_threads(min(16, os.cpu_count() or 1))
    img = rand_image(seed, h, w, n=n)
    return img, O.OracleNet(deepercut_prototxt(152, h, w, n), layers).forward(data=img)


@pytest.fixture(scope="module")
def refs(synth152):
    _, layers = synth152
    return [_oracle(layers, n, h, w, 40 + i) for i, (n, h, w) in enumerate(SHAPES)]


def _group(caffe, path, shapes, lanes=1, **kw):
    """lanes=1 by default HERE: most of these tests are about the merged (multi-problem) launches, and the library's automatic
    choice for two members is two lanes of one member each — nothing merged.  "auto" asks for the library's choice."""
    from deepcut_tools import deepercut_prototxt

    lanes = None if lanes == "auto" else lanes

    n, h, w = shapes[0]
    net = caffe.Net(deepercut_prototxt(152, h, w, n), path, caffe.TEST, from_text=True, **kw)
    return caffe.NetGroup.for_shapes(net, shapes, lanes=lanes)


def _check32(out, ref, tol=1e-3):
    for k in ("prob", "loc_pred", "next_pred"):
        assert out[k].shape == ref[k].shape
        assert float(np.abs(out[k] - ref[k]).max()) <= tol, k


def _check16(out, ref):
    assert float(np.abs(out["prob"] - ref["prob"]).max()) <= 2.5e-3
    for k in ("loc_pred", "next_pred"):
        assert float(np.abs(out[k] - ref[k]).max()) <= 4e-3 * max(1.0, float(np.abs(ref[k]).max())), k


@pytest.mark.parametrize("wino,lanes", [("0", 1), (None, 1), ("0", "auto"), (None, 4)])
def test_group_of_four_scales_matches_the_oracle_and_the_members(gpu_caffe, synth152, refs, monkeypatch, wino, lanes):
    """All four 'scales' in one plan vs the oracle (1e-3) and vs each member run on its own (same arithmetic: 1e-4) — as one
    lane (every layer ONE launch over the four tensors), as the default two lanes (largest + smallest scale | the middle two,
    concurrently on two streams) and as four lanes (nothing merged across members: four streams)."""
    path, _ = synth152
    if wino is not None:
        monkeypatch.setenv("DC_WINOGRAD", wino)  # 0: every convolution merges; default: the Winograd layers run member by member
    grp = _group(gpu_caffe, path, SHAPES, lanes=lanes, hipgraph=1)
    outs = grp.forward_batch([r[0] for r in refs])
    for o, (_, ref) in zip(outs, refs):
        _check32(o, ref)
    st = grp.stats()
    text = grp.plan_text()
    assert st["lanes"] == (2 if lanes == "auto" else lanes)
    if st["lanes"] == 4:
        # one member per lane: nothing to merge, every member's own 158 launches, four streams
        assert st["multi_launches"] == 0 and st["launches"] == 158 * len(SHAPES), st
    else:
        assert st["multi_launches"] >= 100 and "conv_gemm_mp<" in text, text[:400]
        if wino == "0":
            # 158 launches per forward: 157 convolutions, all merged per lane, + the max-pool member by member
            assert st["multi_launches"] == 157 * st["lanes"] and st["launches"] == 157 * st["lanes"] + len(SHAPES), st
        # the heads: (members of a lane) x 4 residue classes problems in one launch
        assert "problems=%d" % (16 // st["lanes"]) in text
    if st["lanes"] == 2:
        assert "lane=0" in text and "lane=1" in text
    keep = [{k: v.copy() for k, v in o.items()} for o in outs]
    for m, (img, _), o in zip(grp.nets, refs, keep):
        own = m.forward_batch(img)
        for k in own:
            assert float(np.abs(own[k] - o[k]).max()) <= 1e-4, k
    # again, grouped: same plan, same graph, same values
    before = grp.stats()
    lows = [m.stats()["lowerings"] for m in grp.nets]
    outs2 = grp.forward_batch([r[0] for r in refs])
    for a, b in zip(outs2, keep):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    after = grp.stats()
    assert after["merges"] == before["merges"] and after["graph_instantiations"] == before["graph_instantiations"]
    assert after["plan_hits"] == before["plan_hits"] + 1
    assert [m.stats()["lowerings"] for m in grp.nets] == lows


def test_group_fp16_batch8_pyramid(gpu_caffe, synth152):
    """BASELINE configs[2] in miniature: batch 8 at four scales of a 136x184 image, float16 operands, one grouped plan."""
    path, layers = synth152
    shapes = [(8, 72, 96), (8, 104, 144), (8, 136, 184), (8, 176, 232)]
    grp = _group(gpu_caffe, path, shapes, lanes="auto", dtype="f16", hipgraph=1)
    data = [_oracle(layers, n, h, w, 60 + i) for i, (n, h, w) in enumerate(shapes)]
    outs = grp.forward_batch([d[0] for d in data])
    for o, (_, ref) in zip(outs, data):
        _check16(o, ref)
        assert float(np.abs(o["loc_pred"] - ref["loc_pred"]).max()) > 1e-6, "suspiciously exact: is the fp16 path running?"
    text = grp.plan_text()
    assert "conv_gemm_mp<d" in text or "conv_gemm_mp<h" in text, text[:300]
    st = grp.stats()
    # two lanes of two scales each: every convolution merged, except that a lane whose members' conv1 was timed onto the float16 stem kernel
    # (stem_f16.hip) runs that one layer member by member
    assert st["lanes"] == 2 and 2 * 156 <= st["multi_launches"] <= 2 * 157, st
    # the device pose decode of every member reads the grouped results
    for m, sc in zip(grp.nets, (0.5, 0.75, 1.0, 1.25)):
        pose = m.decode_pose(sc)
        assert pose.shape == (8, 5, 14) and np.isfinite(pose).all()


def test_group_unfused_graph_and_ab_switch(gpu_caffe, synth152, refs, monkeypatch):
    """DC_OPT_FUSE 0 (every Caffe-visible blob materialised: stand-alone BatchNorm / Scale / ReLU / Eltwise / Crop launches run
    member by member between the merged convolutions) and DC_GROUP=0 (nothing merged) give the same maps."""
    path, _ = synth152
    shapes = SHAPES[:2]
    grp = _group(gpu_caffe, path, shapes, fuse=0)
    outs = grp.forward_batch([r[0] for r in refs[:2]])
    for o, (_, ref) in zip(outs, refs[:2]):
        _check32(o, ref)
    st = grp.stats()
    assert st["multi_launches"] > 100 and st["launches"] > st["multi_launches"] + 100, st
    for m, (_, ref) in zip(grp.nets, refs[:2]):  # intermediate blobs of a member after a grouped forward
        for name in ("conv1", "pool1", "res2c", "res3b7", "res4b35", "res5c"):
            assert float(np.abs(m.blobs[name].data - ref[name]).max()) <= 1e-3 * max(1.0, float(np.abs(ref[name]).max())), name
    monkeypatch.setenv("DC_GROUP", "0")
    grp0 = _group(gpu_caffe, path, shapes, fuse=2)
    outs0 = grp0.forward_batch([r[0] for r in refs[:2]])
    assert grp0.stats()["multi_launches"] == 0
    for o, (_, ref) in zip(outs0, refs[:2]):
        _check32(o, ref)


def test_group_image_entry_equals_the_members_own(gpu_caffe, synth152):
    """dc_group_forward_images: the same uint8 images at four scales, pre-processing + ONE grouped forward + pose decode,
    against dc_net_forward_images of one net scale by scale (the demo's loop, estimate_pose.py:81-128)."""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    scales = (0.5, 0.75, 1.0, 1.25)
    img8 = np.random.RandomState(5).randint(0, 256, (2, 120, 152, 3)).astype(np.uint8)
    single = gpu_caffe.Net(deepercut_prototxt(152, 120, 152, 2), path, gpu_caffe.TEST, from_text=True)
    want = [dict((k, v.copy()) for k, v in single.forward_images(img8, s, want=("prob", "loc_pred", "next_pred"), pose=True).items()) for s in scales]
    shapes = [(2,) + tuple(gpu_caffe.canvas_size(120, 152, s)) for s in scales]
    grp = _group(gpu_caffe, path, shapes)
    got = grp.forward_images(img8, scales, want=("prob", "loc_pred", "next_pred"), pose=True)
    for g, w_ in zip(got, want):
        for k in ("prob", "loc_pred", "next_pred"):
            assert g[k].shape == w_[k].shape and float(np.abs(g[k] - w_[k]).max()) <= 1e-4, k
    # the pose the device decoded for every member == pose_from_maps (pinned to the reference's _pose_from_mats by
    # tests/test_pose.py) on the maps the SAME grouped forward returned
    from pose import estimate_pose as ep

    for g, sc in zip(got, scales):
        for i in range(2):
            assert np.allclose(g["pose"][i], ep.pose_from_maps(g["prob"][i], g["loc_pred"][i], sc), rtol=0, atol=1e-9)
    assert grp.stats()["multi_launches"] >= 100


def test_group_device_entry_and_async_stream(gpu_caffe, synth152, refs):
    import torch

    path, _ = synth152
    shapes = SHAPES[1:3]
    grp = _group(gpu_caffe, path, shapes, hipgraph=1)
    xs = [torch.from_numpy(refs[i + 1][0]).cuda() for i in range(2)]
    outs = [[torch.empty(refs[i + 1][1][k].shape, device="cuda") for k in ("prob", "loc_pred", "next_pred")] for i in range(2)]
    st = torch.cuda.Stream()
    for _ in range(3):
        grp.forward_device([x.data_ptr() for x in xs], shapes, [o[0].data_ptr() for o in outs], [o[1].data_ptr() for o in outs],
                           [o[2].data_ptr() for o in outs], stream=st.cuda_stream)
    st.synchronize()
    for o, (_, ref) in zip(outs, refs[1:3]):
        _check32({"prob": o[0].cpu().numpy(), "loc_pred": o[1].cpu().numpy(), "next_pred": o[2].cpu().numpy()}, ref)


def test_group_rejects_foreign_members(gpu_caffe, synth152):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    a = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True)
    b = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True)
    with pytest.raises(gpu_caffe.DeepcutError):
        gpu_caffe.NetGroup([a, b])  # two models: no shared filter images
    with pytest.raises(gpu_caffe.DeepcutError):
        gpu_caffe.NetGroup([a, a])
    c = a.clone()
    grp = gpu_caffe.NetGroup([a, c])
    with pytest.raises(gpu_caffe.DeepcutError):
        grp.plan_text()  # before the first forward


@pytest.mark.parametrize("v", range(MAX_VARIANTS))
def test_group_with_every_tile_forced(gpu_caffe, synth152, refs, monkeypatch, v):
    """Every tile variant has a multi-problem instantiation; forced one at a time over a 2-member group (the whole graph:
    dense 1x1, strided 1x1, 3x3, dilated 3x3, the stem's row taps, the 8-problem deconvolution heads), against the oracle."""
    table = gpu_caffe.conv_variants()
    if v >= len(table):
        pytest.skip("the variant table has %d entries" % len(table))
    name, esize = table[v]
    path, _ = synth152
    monkeypatch.setenv("DC_CONV_VARIANT", str(v))
    shapes = [SHAPES[0], SHAPES[2]]
    grp = _group(gpu_caffe, path, shapes, dtype="f16" if esize == 2 else "f32")
    outs = grp.forward_batch([refs[0][0], refs[2][0]])
    used = set(ln.split()[1] for ln in grp.plan_text().splitlines()[1:] if "conv_gemm_mp<" in ln)
    assert "conv_gemm_mp<%s>" % name in used, (name, sorted(used))
    for o, (_, ref) in zip(outs, (refs[0], refs[2])):
        (_check16 if esize == 2 else _check32)(o, ref)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_head_channel_split_parity(gpu_caffe, synth152, refs, monkeypatch, dtype):
    """DC_HEAD_SPLIT=1 (the 406-channel head GEMMs as two launches on offset filter rows / constants / output channels), alone
    and grouped, against the oracle: `prob` (sigmoid prefix in the first part), `next_pred` (straddles the cut)."""
    path, _ = synth152
    monkeypatch.setenv("DC_HEAD_SPLIT", "1")
    grp = _group(gpu_caffe, path, SHAPES[1:3], dtype=dtype)
    outs = grp.forward_batch([refs[1][0], refs[2][0]])
    assert "[ch 384-405]" in grp.plan_text() and "[ch 384-405]" in grp.nets[0].plan_text()
    for o, (_, ref) in zip(outs, refs[1:3]):
        (_check16 if dtype == "f16" else _check32)(o, ref)
    own = grp.nets[1].forward_batch(refs[2][0])
    (_check16 if dtype == "f16" else _check32)(own, refs[2][1])


@pytest.mark.parametrize("dtype,depth", [("f32", 1), ("f16", 2)])
def test_sharded_runner_groups_the_scales_of_a_pyramid(gpu_caffe, synth152, dtype, depth):
    """deepcut_tools.ShardedPoseRunner (the product form of BASELINE configs 3-5): by default the batches of a rank run in
    groups of four — the four scales of the pyramid as one launch sequence; poses and maps equal the batch-by-batch run."""
    from deepcut_tools import ShardedPoseRunner, deepercut_prototxt

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 96, 128, 3), path, gpu_caffe.TEST, from_text=True, dtype=dtype, hipgraph=1)
    rs = np.random.RandomState(8)
    imgs = [rs.randint(0, 256, (96, 128, 3)).astype(np.uint8) for _ in range(5)] + [rs.randint(0, 256, (80, 112, 3)).astype(np.uint8)]
    scales = [0.5, 0.75, 1.0, 1.25]
    plain = ShardedPoseRunner(net, max_batch=3, depth=depth, group_size=1).run(imgs, scales, want_maps=True)
    runner = ShardedPoseRunner(net, max_batch=3, depth=depth)
    assert runner.group_size == 4
    res = runner.run(imgs, scales, want_maps=True)
    assert runner._groups and any(g for g in runner._groups), "the grouped path did not run"
    multi = [g.stats()["multi_launches"] for slot in runner._groups for g in slot.values()]
    assert multi and min(multi) >= 100, multi
    tol = 1e-4 if dtype == "f32" else 2e-2
    assert sorted(res["maps"]) == sorted(plain["maps"]) == list(range(len(imgs) * len(scales)))
    for k in res["maps"]:
        for name in ("prob", "loc_pred", "next_pred"):
            a, b = res["maps"][k][name], plain["maps"][k][name]
            assert a.shape == b.shape and float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max())), (k, name)
    assert np.allclose(res["item_poses"][:, 2], plain["item_poses"][:, 2], atol=tol)  # confidences of every item
    assert res["best_scale"] == plain["best_scale"] or dtype == "f16"
    again = runner.run(imgs, scales, want_maps=True)  # second run: plans, graphs and group plans are there
    assert np.array_equal(again["item_poses"], res["item_poses"])
    # poses only: the runner leaves `next_pred` out of the lowering (DC_OPT_OUTPUTS; the reference's demo reads prob and loc_pred only,
    # estimate_pose.py:231-241) — the 42-channel heads run on other tiles, so the poses agree up to the summation order
    only = runner.run(imgs, scales)
    assert net.wanted_outputs == ["loc_pred", "prob"] and "res5c_up_next" not in net.plan_text()
    assert np.allclose(only["item_poses"], res["item_poses"], atol=tol * 50), float(np.abs(only["item_poses"] - res["item_poses"]).max())
    assert runner.run(imgs, scales, want_maps=True)["maps"][0]["next_pred"].shape[0] == 364  # and back


def test_group_tile_choices_persist_in_the_tune_cache_file(gpu_caffe, synth152, refs, monkeypatch, tmp_path):
    """DC_TUNE_CACHE: the group signatures (their members' signatures joined: 150 characters for two members, several hundred
    for the 16-problem heads of a 4-scale pyramid) are written next to the single-problem ones and read back whole — a second
    model instance times nothing."""
    path, _ = synth152
    cache = tmp_path / "tune.txt"
    monkeypatch.setenv("DC_TUNE_CACHE", str(cache))
    shapes = SHAPES[1:3]
    g1 = _group(gpu_caffe, path, shapes)
    out1 = g1.forward_batch([refs[1][0], refs[2][0]])
    assert g1.stats()["autotune_runs"] == 1
    lines = cache.read_text().splitlines()
    long_keys = [ln for ln in lines if ln.startswith("G")]
    assert long_keys and max(len(ln) for ln in long_keys) > 120 and all(len(ln.rsplit(" ", 1)) == 2 for ln in lines)
    g2 = _group(gpu_caffe, path, shapes)  # a new model (own ModelShared): everything comes from the file
    out2 = g2.forward_batch([refs[1][0], refs[2][0]])
    assert g2.stats()["autotune_runs"] == 0 and all(m.stats()["autotune_runs"] == 0 for m in g2.nets)
    assert g2.plan_text() == g1.plan_text()
    for a, b in zip(out1, out2):
        for k in a:
            assert np.array_equal(a[k], b[k]), k


def test_group_tune_report_and_set_tile(gpu_caffe, synth152, refs, monkeypatch):
    """dc_group_tune_report / dc_group_set_tile (what deepcut_tools.tune_in_flight drives when the load is groups in flight): every
    merged signature with its isolated timings; an override shows in the plan, leaves the maps where they were (1e-4: another tile
    may sum in another order) and reaches a second group of the same model through the shared choice table."""
    from deepcut_tools import tune_in_flight

    monkeypatch.delenv("DC_TUNE_CACHE", raising=False)
    path, _ = synth152
    shapes = SHAPES[1:3]
    grp = _group(gpu_caffe, path, shapes, hipgraph=1)
    imgs = [refs[1][0], refs[2][0]]
    base = [{k: v.copy() for k, v in o.items()} for o in grp.forward_batch(imgs)]
    rep = grp.tune_report()
    assert len(rep) >= 20 and all(r["signature"].startswith("G") and r["timed"] for r in rep)
    assert sum(r["launches"] for r in rep) == grp.stats()["multi_launches"]
    busiest = max(rep, key=lambda r: r["launches"])
    other = [t for t, _ in busiest["timed"] if t != busiest["tile"]][0]
    grp.set_tile(busiest["signature"], other)
    assert [r["tile"] for r in grp.tune_report() if r["signature"] == busiest["signature"]] == [other]
    assert "conv_gemm_mp<%s>" % other in grp.plan_text()
    for o, b in zip(grp.forward_batch(imgs), base):
        for k in o:
            assert float(np.abs(o[k] - b[k]).max()) <= 1e-4, k
    with pytest.raises(gpu_caffe.DeepcutError):
        grp.set_tile(busiest["signature"], "no_such_tile")
    with pytest.raises(gpu_caffe.DeepcutError):
        grp.set_tile("G2:1/2/3", other)
    # the descent itself, over two groups of the model (the second one picks the override up from the shared table)
    grp2 = gpu_caffe.NetGroup.for_shapes(grp.nets[0].clone(), shapes, lanes=1)
    grp2.forward_batch(imgs)
    assert [r["tile"] for r in grp2.tune_report() if r["signature"] == busiest["signature"]] == [other]
    import time

    def load():
        t0 = time.perf_counter()
        for g in (grp, grp2):
            g.forward_batch(imgs)
        return time.perf_counter() - t0

    res = tune_in_flight([grp, grp2], load, top=2, reps=1)
    assert res["runs"] >= 2 and [r["tile"] for r in grp.tune_report()] == [r["tile"] for r in grp2.tune_report()]
    for o, b in zip(grp2.forward_batch(imgs), base):
        for k in o:
            assert float(np.abs(o[k] - b[k]).max()) <= 1e-4, k


def test_group_follows_its_members_weights_and_tiles(gpu_caffe, synth152, refs):
    """A member is an ordinary net: new weights through one executor (CopyTrainedLayersFrom semantics: every executor of the model
    re-packs) and a tile override on a member reach the next grouped forward — the group re-merges instead of replaying a graph that
    captured the old filter images / kernels."""
    from deepcut_tools import synth_weights, write_caffemodel

    path, _ = synth152
    shapes = SHAPES[:2]
    grp = _group(gpu_caffe, path, shapes, hipgraph=1)
    imgs = [refs[0][0], refs[1][0]]
    a = [{k: v.copy() for k, v in o.items()} for o in grp.forward_batch(imgs)]
    for o, (_, ref) in zip(a, refs[:2]):
        _check32(o, ref)
    merges = grp.stats()["merges"]
    # a member's own tile changes: the group's graph holds that member's (non-merged) launches too
    rep = grp.nets[1].tune_report()
    sig = max(rep, key=lambda r: r["launches"])
    other = [t for t, _ in sig["timed"] if t != sig["tile"]]
    if other:
        grp.nets[1].set_tile(sig["signature"], other[0])
        b = grp.forward_batch(imgs)
        assert grp.stats()["merges"] == merges + 1
        for o, x in zip(b, a):
            for k in o:
                assert float(np.abs(o[k] - x[k]).max()) <= 1e-4, k
    # new weights (another seed) through member 0: both members' maps change and match the oracle of the new weights
    import os
    import tempfile

    layers2 = synth_weights(152, seed=5)
    with tempfile.TemporaryDirectory() as d:
        p2 = os.path.join(d, "w2.caffemodel")
        write_caffemodel(p2, "ResNet-152", layers2)
        grp.nets[0].copy_from(p2)
    c = grp.forward_batch(imgs)
    from deepcut_tools import deepercut_prototxt

    for o, x, (n, h, w), img in zip(c, a, shapes, imgs):
        assert float(np.abs(o["loc_pred"] - x["loc_pred"]).max()) > 1e-3  # really other weights
        ref = O.OracleNet(deepercut_prototxt(152, h, w, n), layers2).forward(data=img)
        _check32(o, ref)


def test_group_plan_cache_serves_shape_sets_met_before(gpu_caffe, synth152, refs):
    """Like a net's per-shape plan cache (layer.hpp:451-456: every forward re-derives the shapes): a group keeps the merged plan and
    the graphs of every tuple of member shapes it has met; alternating two shape sets costs two merges in total, members run alone
    in between keep working, and the results do not drift."""
    path, _ = synth152
    grp = _group(gpu_caffe, path, SHAPES[:2], hipgraph=1)
    set_a = [refs[0][0], refs[1][0]]           # shapes (2,40,56), (2,64,64)
    set_b = [refs[1][0], refs[0][0]]           # the same tensors on the other members: another tuple of shapes
    first_a = [{k: v.copy() for k, v in o.items()} for o in grp.forward_batch(set_a)]
    first_b = [{k: v.copy() for k, v in o.items()} for o in grp.forward_batch(set_b)]
    base = grp.stats()
    assert base["merges"] == 2
    for rnd in range(3):
        own = grp.nets[rnd % 2].forward_batch(refs[2][0])  # a member alone, at a third shape, between grouped forwards
        _check32(own, refs[2][1])
        for imgs, first in ((set_a, first_a), (set_b, first_b)):
            outs = grp.forward_batch(imgs)
            for o, f in zip(outs, first):
                for k in o:
                    assert np.array_equal(o[k], f[k]), (rnd, k)
    st = grp.stats()
    # the lone forwards lowered a third shape on each member once: that re-merges each cached tuple at most once more
    assert st["merges"] <= base["merges"] + 4 and st["plan_hits"] >= 2, st
    for o, (_, ref) in zip(first_b, (refs[1], refs[0])):
        _check32(o, ref)


def test_automatic_lanes(gpu_caffe, synth152, refs):
    """The library's own choice: two members -> two lanes of one member (plain concurrency, nothing merged: measured faster than
    merging two tensors), three -> one lane (everything merged), four -> two lanes of two merged members; results as ever."""
    path, _ = synth152
    for k, want_lanes, merged in ((2, 2, False), (3, 1, True), (4, 2, True)):
        grp = _group(gpu_caffe, path, SHAPES[:k], lanes="auto", hipgraph=1)
        outs = grp.forward_batch([r[0] for r in refs[:k]])
        st = grp.stats()
        assert st["lanes"] == want_lanes and (st["multi_launches"] > 100) == merged, (k, st)
        for o, (_, ref) in zip(outs, refs[:k]):
            _check32(o, ref)


def test_group_profile_refuses_a_plan_its_members_have_left(gpu_caffe, synth152, refs):
    """profile_text() / set_tile() launch from the plan of the group's LAST forward.  A member that has since been run alone at
    another shape has moved on (other plan, possibly other buffers): the group says so instead of launching into the old ones,
    and the next grouped forward puts it right."""
    path, _ = synth152
    grp = _group(gpu_caffe, path, SHAPES[:2], hipgraph=1)
    with pytest.raises(gpu_caffe.DeepcutError, match="forward first"):
        grp.profile_text(1)
    outs = grp.forward_batch([refs[0][0], refs[1][0]])
    assert "conv_gemm_mp<" in grp.profile_text(1)
    rep = grp.tune_report()[0]
    key, tile = rep["signature"], rep["tile"]
    grp.nets[0].forward_batch(refs[3][0])  # member 0 alone at the largest shape: its buffers grow
    with pytest.raises(gpu_caffe.DeepcutError, match="member changed"):
        grp.profile_text(1)
    with pytest.raises(gpu_caffe.DeepcutError, match="member changed"):
        grp.set_tile(key, tile)
    again = grp.forward_batch([refs[0][0], refs[1][0]])
    for o, a, (_, ref) in zip(outs, again, refs[:2]):
        _check32(a, ref)
    assert "conv_gemm_mp<" in grp.profile_text(1)


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_groups_on_concurrent_host_threads(dtype):
    """Three groups of one model, each driven by its own host thread on its own stream (tools/stress_groups.py): alternating tuples
    of member shapes, members run alone in between — also at new, ever larger shapes: lowering, tile timing, buffer growth and a
    graph capture while the other threads run and capture — and synchronous legacy-stream work of the host application beside
    it.  Every grouped result must equal the group's first for the same inputs bit for bit, and the groups must agree with each
    other (they share the model's tile choices).  A process of its own: the HIP runtime's capture rule is process-wide
     and so is what this guards."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_groups.py"), dtype, "120", "3", "3"], capture_output=True,
                       text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "sequential round" not in r.stdout, tail  # the groups agreed before the threads started
    assert "OK mismatches [0, 0, 0] rounds done [120, 120, 120]" in r.stdout, tail


@pytest.mark.parametrize("dtype,threads", [("f32", 3), ("f16", 4)])
def test_clones_on_concurrent_host_threads(dtype, threads):
    """The plain-net companion (tools/stress_clones.py): every thread drives a clone of its own through the device, host and
    image entries over shapes old and new; every result bit-identical to executor 0's sequential one."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_clones.py"), dtype, "60", str(threads)], capture_output=True,
                       text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "OK mismatches %r rounds done %r" % ([0] * threads, [60] * threads) in r.stdout, tail


def test_nothing_stays_behind_when_executors_go():
    """tools/leak_check.py: cycles of net + clones + group, forwards through every entry at two shape sets, everything dropped —
    device memory, resident set and file descriptors must not grow from the second cycle on."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "leak_check.py"), "f32", "5"], capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "\nOK: over cycles" in r.stdout, tail
