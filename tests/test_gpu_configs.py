"""-m gpu: the BASELINE.json configurations that are not the bench line, each checked (not just timed) on one GPU.

configs[2]  batch 8 x 4-scale pyramid, fp16 MFMA / fp32 accumulate: every scale of the pyramid at batch 8
configs[3]  batch 64 sharded 8-way: ONE rank's share (8 images of 1x3x544x736: seeds of rank 0 under the LPT schedule),
            fp32 within 1e-3 and fp16 within the stated fp16 bounds of the CPU oracle
configs[4]  32 person crops x 4 scales with the pairwise maps on: all 128 work items through ShardedPoseRunner with
            want_maps (next_pred included), a sample of items against the CPU oracle (its own pre-processing + forward),
            the rest through properties
and the N>1 product path on hardware: 2 ranks sharing GPU 0 over gloo == world size 1 (tests/_gpu_rank_worker.py).
SURVEY §8d gives the synthetic inputs (seeds, sizes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from oracle import preprocess as OP

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP32_TOL = 1e-3


def _fp16_check(out, ref):
    """the float16 bounds of tests/test_gpu_fp16.py"""
    assert float(np.abs(out["prob"] - ref["prob"]).max()) <= 2.5e-3
    for k in ("loc_pred", "next_pred"):
        rng = max(1.0, float(np.abs(ref[k]).max()))
        assert float(np.abs(out[k] - ref[k]).max()) <= 4e-3 * rng, k


def _oracle_threads():
    O.set_threads(min(16, os.cpu_count() or 1))  # the GPU box grants 16 CPUs of cgroup quota


def _config3_share():
    """Input of rank 0 in configs[3]: 64 images (seeds 100..163) dealt to 8 ranks by the product's LPT schedule."""
    from deepcut_tools import lpt_shards

    mine = lpt_shards([544 * 736] * 64, 8)[0]
    assert len(mine) == 8
    return np.concatenate([(np.random.RandomState(100 + i).randn(1, 3, 544, 736) * 50).astype(np.float32) for i in mine])


@pytest.fixture(scope="module")
def config3_reference(synth152):
    from deepcut_tools import deepercut_prototxt

    _path, layers = synth152
    x = _config3_share()
    _oracle_threads()
    return x, O.OracleNet(deepercut_prototxt(152, 544, 736, 8), layers).forward(data=x)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_config3_one_ranks_share_batch8_matches_oracle(gpu_caffe, synth152, config3_reference, dtype):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    x, ref = config3_reference
    net = gpu_caffe.Net(deepercut_prototxt(152, 544, 736, 8), path, gpu_caffe.TEST, from_text=True, dtype=dtype, hipgraph=1)
    out = net.forward_batch(x)
    assert abs(net.flops() / 1e9 - 8 * 241.09) < 0.1
    for k, c in (("prob", 14), ("loc_pred", 28), ("next_pred", 364)):
        assert out[k].shape == ref[k].shape == (8, c, 68, 92)
    if dtype == "f32":
        for k in out:
            err = float(np.abs(out[k] - ref[k]).max())
            print(k, "max abs err", err)
            assert err <= FP32_TOL, k
    else:
        _fp16_check(out, ref)
    # the maps as the gather payload: device copy-out in the net's own element type equals the host copy-out
    import torch

    n14, n28 = 8 * 14 * 68 * 92, 8 * 28 * 68 * 92
    buf = torch.empty(8 * 406 * 68 * 92, dtype=torch.float16 if dtype == "f16" else torch.float32, device="cuda:0")
    net.emit_maps_device(buf[:n14].data_ptr(), buf[n14:n14 + n28].data_ptr(), buf[n14 + n28:].data_ptr(), half=dtype == "f16")
    got = buf.float().cpu().numpy()
    assert np.array_equal(got[:n14].reshape(8, 14, 68, 92), out["prob"])
    assert np.array_equal(got[n14 + n28:].reshape(8, 364, 68, 92), out["next_pred"])
    if dtype == "f32":
        with pytest.raises(gpu_caffe.DeepcutError):  # half payloads are an fp16 net's
            net.emit_maps_device(buf.data_ptr(), half=True)


@pytest.mark.parametrize("scale_hw", [(408, 552), (544, 736), (680, 920)])
def test_config2_fp16_batch8_every_pyramid_scale(gpu_caffe, synth152, scale_hw):
    """configs[2] at batch 8 for the scales tests/test_gpu_fp16.py does not cover (272x368 is there).  One image of the batch
    against the CPU oracle at the fp16 bounds; the others through the batch property (equal to a forward of their own
    up to fp16 rounding of a different tile schedule), bit-identical repetition, prob a probability."""
    from deepcut_tools import deepercut_prototxt

    path, layers = synth152
    h, w = scale_hw
    net = gpu_caffe.Net(deepercut_prototxt(152, h, w, 8), path, gpu_caffe.TEST, from_text=True, dtype="f16", hipgraph=1)
    imgs = np.concatenate([(np.random.RandomState(10 + i).randn(1, 3, h, w) * 50).astype(np.float32) for i in range(8)])
    a = net.forward_batch(imgs)
    b = net.forward_batch(imgs)
    for k in a:
        assert a[k].shape[0] == 8 and a[k].shape[2:] == (h // 8, w // 8)
        assert np.array_equal(a[k], b[k]), k
        assert np.isfinite(a[k]).all()
    assert (a["prob"] > 0).all() and (a["prob"] < 1).all()
    _oracle_threads()
    j = 3
    ref = O.OracleNet(deepercut_prototxt(152, h, w, 1), layers).forward(data=imgs[j:j + 1])
    _fp16_check({k: v[j:j + 1] for k, v in a.items()}, ref)
    pose = net.decode_pose(h / 544.0)  # the device decode reads the half maps of the batch
    assert pose.shape == (8, 5, 14) and np.isfinite(pose).all()
    one = net.forward_batch(imgs[5:6])
    for k in a:
        rng = max(1.0, float(np.abs(a[k]).max()))
        assert float(np.abs(a[k][5] - one[k][0]).max()) <= 4e-3 * rng, k


def _crowd_crops():
    # configs[4]: 32 crops W=256, H=336, seeds 200..231
    return [np.random.RandomState(200 + i).randint(0, 256, (336, 256, 3)).astype(np.uint8) for i in range(32)]


CROWD_SCALES = [0.5, 0.75, 1.0, 1.25]


@pytest.mark.parametrize("depth", [1, 3])
def test_config4_crowd_crops_with_pairwise_maps(gpu_caffe, synth152, depth):
    from deepcut_tools import ShardedPoseRunner, deepercut_prototxt
    from pose import estimate_pose as ep

    path, layers = synth152
    crops = _crowd_crops()
    net = gpu_caffe.Net(deepercut_prototxt(152, 336, 256), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    runner = ShardedPoseRunner(net, max_batch=16, depth=depth)
    res = runner.run(crops, CROWD_SCALES, want_maps=True)
    items = res["items"]
    assert len(items) == 128 and sorted(res["maps"]) == list(range(128))
    assert [it[2] for it in items[:4]] == [(168, 128), (256, 192), (336, 256), (424, 320)]
    # a sample of (crop, scale) items against the CPU oracle, one per scale and a second one of the largest
    _oracle_threads()
    for k in (0, 4 * 7 + 1, 4 * 19 + 2, 4 * 31 + 3, 4 * 12 + 3):
        i, s, (H, W) = items[k]
        x = OP.preprocess(crops[i], s).transpose(2, 0, 1)[None].astype(np.float32)
        assert x.shape == (1, 3, H, W)
        ref = O.OracleNet(deepercut_prototxt(152, H, W), layers).forward(data=x)
        for name in ("prob", "loc_pred", "next_pred"):
            got = res["maps"][k][name]
            assert got.shape == ref[name][0].shape
            err = float(np.abs(got - ref[name][0]).max())
            assert err <= FP32_TOL, (k, name, err)
    # every item: shapes, finiteness, prob a probability, the device decode == the reference decode of the same maps
    for k, (i, s, (H, W)) in enumerate(items):
        m = res["maps"][k]
        assert m["prob"].shape == (14, H // 8, W // 8) and m["next_pred"].shape == (364, H // 8, W // 8)
        assert all(np.isfinite(v).all() for v in m.values())
        assert (m["prob"] > 0).all() and (m["prob"] < 1).all()
        assert np.allclose(res["item_poses"][k], ep.pose_from_maps(m["prob"], m["loc_pred"], s), rtol=0, atol=1e-9)
    # best-scale rule (estimate_pose.py:119-126) over the gathered items
    for i in range(32):
        confs = [float(res["item_poses"][4 * i + q][2].min()) for q in range(4)]
        best, bc = None, 0.0
        for q, c in enumerate(confs):
            if c > bc:
                best, bc = q, c
        assert res["best_scale"][i] == (None if best is None else CROWD_SCALES[best])
    # shapes met once: lowered once, served from the plan cache afterwards; a second run re-lowers nothing
    before = [e.stats() for e in runner._execs]
    res2 = runner.run(crops, CROWD_SCALES, want_maps=True)
    after = [e.stats() for e in runner._execs]
    for b, a in zip(before, after):
        assert a["lowerings"] == b["lowerings"] and a["graph_instantiations"] == b["graph_instantiations"], (b, a)
    assert np.array_equal(res2["item_poses"], res["item_poses"])


def test_pyramid_shapes_are_lowered_and_captured_once(gpu_caffe, synth152):
    """Layer::Forward reshapes on every call (layer.hpp:451-456) and the demo's scale loop changes the shape on every
    iteration (estimate_pose.py:81-128): cycling the four pyramid shapes costs four lowerings and four graph
    instantiations in total — none after the first cycle — and reproduces every result bit for bit."""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    shapes = [(136, 184), (208, 280), (272, 368), (344, 464)]  # a 4-scale pyramid (0.5 .. 1.25 of 272x368)
    net = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    net.reserve(2, *shapes[-1])  # the largest shape first: no buffer grows afterwards
    imgs = {s: (np.random.RandomState(s[0]).randn(2, 3, *s) * 50).astype(np.float32) for s in shapes}
    first, st1 = {}, None
    for cycle in range(3):
        for s in shapes:
            out = net.forward_batch(imgs[s])
            if cycle == 0:
                first[s] = {k: v.copy() for k, v in out.items()}
            else:
                for k in out:
                    assert np.array_equal(out[k], first[s][k]), (cycle, s, k)
        if cycle == 0:
            st1 = net.stats()
    st = net.stats()
    assert st1["lowerings"] == 4 and st1["graph_instantiations"] == 4 and st1["cached_plans"] == 4
    assert st["lowerings"] == 4 and st["graph_instantiations"] == 4 and st["plan_hits"] >= 8
    assert st["buffer_growths"] == st1["buffer_growths"] and st["repacks"] == 1
    # without the reservation a growing buffer costs re-captures of the stale graphs, never a re-lowering
    net2 = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    own = {}
    for cycle in range(3):
        for s in shapes:
            out = net2.forward_batch(imgs[s])
            for k in out:
                # another net times its tiles itself: equal to the first net up to fp32 summation order, bit-identical to itself
                assert np.abs(out[k] - first[s][k]).max() <= 1e-4, (s, k)
                assert np.array_equal(out[k], own.setdefault((s, k), out[k].copy())), (cycle, s, k)
    assert net2.stats()["lowerings"] == 4 and net2.stats()["graph_instantiations"] <= 8


def test_parameter_write_reaches_a_clone_on_the_device(gpu_caffe, synth152):
    """Executors of one model share parameters AND the weight generation: a write through the parent re-packs for both."""
    from deepcut_tools import deepercut_prototxt
    from conftest import rand_image

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 64, 64), path, gpu_caffe.TEST, from_text=True)
    c = net.clone()
    x = rand_image(7, 64, 64)
    a = net.forward_batch(x)
    b = c.forward_batch(x)
    for k in a:
        assert np.array_equal(a[k], b[k])
    net.params["conv1"][0].data[...] *= 0.5
    a2 = net.forward_batch(x)
    b2 = c.forward_batch(x)
    for k in a:
        assert np.array_equal(a2[k], b2[k]), k
    assert not np.array_equal(a2["loc_pred"], a["loc_pred"])
    assert c.stats()["lowerings"] == 2 and net.stats()["repacks"] + c.stats()["repacks"] == 2


def _run_ranks(nproc, script_args, timeout=900):
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_two_ranks_sharing_gpu0_equal_world_size_1(gpu_caffe, synth152, tmp_path):
    """The N>1 product path on hardware: ShardedPoseRunner with the real Net in 2 processes over gloo (both on GPU 0),
    maps exchanged batch by batch, against the single-process run of the same schedule."""
    from deepcut_tools import ShardedPoseRunner, deepercut_prototxt

    path, _ = synth152
    out = str(tmp_path / "ranks.npz")
    r = _run_ranks(2, [os.path.join(ROOT, "tests", "_gpu_rank_worker.py"), path, out])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    from _gpu_rank_worker import SCALES, worker_images

    imgs = worker_images()
    net = gpu_caffe.Net(deepercut_prototxt(152, 96, 128), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    single = ShardedPoseRunner(net, max_batch=4, depth=2).run(imgs, SCALES, want_maps=True)
    # the two runs batch the items differently (each rank batches its own share): fp32 summation order only
    assert np.abs(got["item_poses"] - single["item_poses"]).max() <= 1e-2
    assert list(got["best_scale"]) == [s if s is not None else -1.0 for s in single["best_scale"]]
    for k in range(len(single["items"])):
        for name in ("prob", "loc_pred", "next_pred"):
            assert np.abs(got["%s_%d" % (name, k)] - single["maps"][k][name]).max() <= 1e-4, (k, name)


def test_eight_ranks_sharing_gpu0_equal_world_size_1(gpu_caffe, synth152, tmp_path, monkeypatch):
    """The design point of the N>1 path, on the hardware there is: EIGHT ranks (all on GPU 0, over gloo) run
    ShardedPoseRunner with the real Net on configs[3] in small — 64 equal images dealt 8 per rank, the maps of every batch
    sent to rank 0, which posts 7 receives per exchange round — and equal the single-process run."""
    from deepcut_tools import ShardedPoseRunner, deepercut_prototxt

    path, _ = synth152
    out = str(tmp_path / "ranks8.npz")
    monkeypatch.setenv("DC_AUTOTUNE", "0")  # eight processes timing tiles on one GPU at once would only measure each other
    r = _run_ranks(8, [os.path.join(ROOT, "tests", "_gpu_rank_worker.py"), path, out, "c3"], timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = np.load(out)
    from _gpu_rank_worker import worker_images

    imgs = worker_images("c3")
    net = gpu_caffe.Net(deepercut_prototxt(152, 96, 128), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    single = ShardedPoseRunner(net, max_batch=8, depth=2).run(imgs, [1.0], want_maps=True)
    assert len(single["items"]) == 64
    assert np.abs(got["item_poses"] - single["item_poses"]).max() <= 1e-2
    for k in range(64):
        for name in ("prob", "loc_pred", "next_pred"):
            assert np.abs(got["%s_%d" % (name, k)] - single["maps"][k][name]).max() <= 1e-4, (k, name)


def test_bench_eight_ranks_gloo_smoke(gpu_caffe, monkeypatch):
    """bench.py --gpus 8 --config 3 (what the driver's scaling run launches, small maps, gloo transport, one GPU)."""
    import json

    monkeypatch.setenv("DC_AUTOTUNE", "0")
    r = _run_ranks(8, [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--config", "3", "--height", "96", "--width", "128",
                       "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 8 and res["steps"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config"]["per_gpu_batch"] == 8 and res["config"]["global_batch"] == 64


@pytest.mark.parametrize("extra", [[], ["--config", "3", "--dtype", "f16"]])
def test_bench_two_ranks_gloo_smoke(gpu_caffe, extra):
    """bench.py --gpus 2 over gloo with both ranks on GPU 0: the N>1 bench path (per-rank forward, gather of the maps to
    rank 0, max-over-ranks timing) runs on hardware and prints one well-formed line."""
    import json

    r = _run_ranks(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                       "--no-cpu-baseline", "--height", "272", "--width", "368"] + extra)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 2 * res["config"]["per_gpu_batch"]
    if extra:
        assert res["config"]["per_gpu_batch"] == 8 and res["dtype"] == "f16"


def test_float16_net_ships_float16_maps_and_coalesces_requests(gpu_caffe, synth152):
    """A float16 net hands its maps over as float16 (half the gather payload): the runner's device pipeline returns exactly
    the values the host copy-out of the same forwards gives; and cross-request batching on a float16 net."""
    import torch
    from deepcut_tools import Pipeline, ShardedPoseRunner, deepercut_prototxt

    path, _ = synth152
    rs = np.random.RandomState(6)
    imgs = [rs.randint(0, 256, (96, 128, 3)).astype(np.uint8) for _ in range(5)]
    net = gpu_caffe.Net(deepercut_prototxt(152, 96, 128), path, gpu_caffe.TEST, from_text=True, dtype="f16", hipgraph=1)
    assert net.dtype == "f16"
    res = ShardedPoseRunner(net, max_batch=4, depth=2).run(imgs, [1.0, 0.75], want_maps=True)
    for k, (i, s, hw) in enumerate(res["items"]):
        out = net.forward_images(imgs[i][None], s, want=("prob", "loc_pred", "next_pred"), pose=True)
        for name in ("prob", "loc_pred", "next_pred"):
            # batch composition differs (runner batches same-shape items), so fp16 rounding of a different tile schedule
            assert np.abs(res["maps"][k][name] - out[name][0]).max() <= 4e-3 * max(1.0, float(np.abs(out[name]).max())), (k, name)
            assert res["maps"][k][name].dtype == np.float32
    # cross-request batching on the float16 net: each request's maps equal its own single forward up to that rounding
    dev = torch.device("cuda", 0)
    h, w = 96, 128
    xs = [torch.from_numpy((rs.randn(1, 3, h, w) * 50).astype(np.float32)).to(dev) for _ in range(4)]
    outs = [[torch.empty(1, c, h // 8, w // 8, device=dev) for c in (14, 28, 364)] for _ in xs]
    pipe = Pipeline(net, depth=2, coalesce=2)
    for i, x in enumerate(xs):
        pipe.submit(x.data_ptr(), 1, h, w, outs[i][0].data_ptr(), outs[i][1].data_ptr(), outs[i][2].data_ptr(), tag=i)
    assert sorted(pipe.drain()) == [0, 1, 2, 3]
    for i, x in enumerate(xs):
        ref = net.forward_batch(x.cpu().numpy())
        for name, t in zip(("prob", "loc_pred", "next_pred"), outs[i]):
            assert np.abs(t.cpu().numpy() - ref[name]).max() <= 4e-3 * max(1.0, float(np.abs(ref[name]).max())), (i, name)
