"""CPU: deepcut_tools.ShardedPoseRunner — the multi-scale / multi-crop product path of BASELINE configs 3-5 —
single process and world_size 2 over gloo, with a deterministic stand-in for the network (the real forward
needs a GPU; tests/test_gpu_fullnet.py runs the runner on the HIP path)."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from deepcut_tools import ShardedPoseRunner, net_input_shape, plan_work


class FakeNet(object):
    """Deterministic 'detector': maps are a fixed function of the input batch, pose decode as the reference."""

    def forward_batch(self, images, want=()):
        n, _, h, w = images.shape
        hh, ww = h // 8, w // 8
        base = images.reshape(n, 3, hh, 8, ww, 8).mean(axis=(3, 5))  # [n,3,hh,ww]
        ch = np.arange(14).reshape(1, 14, 1, 1)
        self.prob = 1.0 / (1.0 + np.exp(-(base[:, :1] * 0.02 + np.sin(ch + base[:, 1:2] * 0.01))))
        self.loc = np.tanh(np.repeat(base[:, 2:3], 28, axis=1) * 0.01 + np.arange(28).reshape(1, 28, 1, 1) * 0.1)
        out = {}
        if "prob" in want:
            out = {"prob": self.prob.astype(np.float32), "loc_pred": self.loc.astype(np.float32),
                   "next_pred": np.zeros((n, 364, hh, ww), np.float32)}
        return out

    def decode_pose(self, scale):
        from pose.estimate_pose import pose_from_maps

        return np.stack([pose_from_maps(self.prob[i], self.loc[i], scale) for i in range(self.prob.shape[0])])


def _images():
    rs = np.random.RandomState(3)
    return [rs.randint(0, 256, (h, w, 3)).astype(np.uint8) for (h, w) in [(96, 128), (96, 128), (120, 88), (64, 64), (97, 130)]]


def test_plan_is_balanced_and_deterministic():
    shapes = [(336, 256)] * 32
    items, shards = plan_work(shapes, [0.5, 0.75, 1.0, 1.25], 8)  # BASELINE configs[4]
    assert len(items) == 128 and sorted(i for s in shards for i in s) == list(range(128))
    assert [it[2] for it in items[:4]] == [(168, 128), (256, 192), (336, 256), (424, 320)]
    loads = [sum(items[k][2][0] * items[k][2][1] for k in s) for s in shards]
    assert max(loads) / (sum(loads) / 8.0) < 1.02
    assert net_input_shape((544, 736), 1.25) == (680, 920)


def test_single_process_matches_the_sequential_reference_loop():
    from pose.estimate_pose import pose_from_maps, preprocess, select_best

    imgs = _images()
    scales = [0.75, 1.0, 1.25]
    res = ShardedPoseRunner(FakeNet()).run(imgs, scales)
    net = FakeNet()
    for i, im in enumerate(imgs):
        poses = []
        for s in scales:
            x = preprocess(im, s).transpose(2, 0, 1)[None]
            net.forward_batch(x)
            poses.append(pose_from_maps(net.prob[0], net.loc[0], s))
        ref = select_best(poses)
        assert np.allclose(res["poses"][i], ref, atol=1e-9)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, n_images=5, scales=(0.75, 1.0, 1.25)):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "..", "deepcut-cnn_amd", "python"))
    sys.path.insert(0, here)
    from test_runner_gloo import FakeNet, _images
    from deepcut_tools import ShardedPoseRunner

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = ShardedPoseRunner(FakeNet()).run(_images()[:n_images], list(scales), want_maps=True)
        if rank == 0:
            q.put((res["item_poses"], [p is not None for p in res["poses"]], sorted(res["maps"].keys()),
                   {k: v["prob"].shape for k, v in res["maps"].items()}))
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world2_gloo_equals_single_process():
    single = ShardedPoseRunner(FakeNet()).run(_images(), [0.75, 1.0, 1.25], want_maps=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    item_poses, have, map_keys, map_shapes = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.allclose(item_poses, single["item_poses"], atol=1e-12)
    assert map_keys == list(range(15)) and all(have)
    for k, shp in map_shapes.items():
        assert shp == single["maps"][k]["prob"].shape


def test_world2_gloo_with_a_rank_that_has_no_work():
    """One image at one scale on two ranks: rank 1's shard is empty — it posts no send, rank 0 no receive for it."""
    single = ShardedPoseRunner(FakeNet()).run(_images()[:1], [1.0], want_maps=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 1, (1.0,))) for r in range(2)]
    for p in procs:
        p.start()
    item_poses, have, map_keys, _shapes = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.allclose(item_poses, single["item_poses"], atol=1e-12) and map_keys == [0] and have == [True]


def _run_world(world, n_images, scales, images=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q, n_images, tuple(scales))) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return out


def _images64(n):
    """n equal-size 96x128 images (BASELINE configs[3]'s 64 equal images, small) / crops of one size (configs[4])."""
    rs = np.random.RandomState(11)
    return [rs.randint(0, 256, (96, 128, 3)).astype(np.uint8) for _ in range(n)]


def _worker8(rank, world, port, q, n_images, scales):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "..", "deepcut-cnn_amd", "python"))
    sys.path.insert(0, here)
    from test_runner_gloo import FakeNet, _images64
    from deepcut_tools import ShardedPoseRunner

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = ShardedPoseRunner(FakeNet()).run(_images64(n_images), list(scales), want_maps=True)
        if rank == 0:
            q.put((res["item_poses"], [p is not None for p in res["poses"]], sorted(res["maps"].keys()),
                   float(sum(float(v["prob"].sum()) for v in res["maps"].values()))))
        else:
            assert res is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world8_gloo_config3_64_equal_images():
    """BASELINE configs[3] at its design point: 64 equal images dealt 8 per rank over 8 ranks, maps gathered to rank 0
    (7 peers per exchange round), equal to the single-process run."""
    from deepcut_tools import plan_work

    items, shards = plan_work([(96, 128)] * 64, [1.0], 8)
    assert [len(s) for s in shards] == [8] * 8
    single = ShardedPoseRunner(FakeNet()).run(_images64(64), [1.0], want_maps=True)
    item_poses, have, map_keys, checksum = _run_world(8, 64, [1.0])
    assert np.allclose(item_poses, single["item_poses"], atol=1e-12)
    assert map_keys == list(range(64)) and all(have)
    assert abs(checksum - sum(float(v["prob"].sum()) for v in single["maps"].values())) < 1e-6 * abs(checksum)


def test_world8_gloo_config4_crops_times_scales_lpt():
    """BASELINE configs[4]: 32 crops x 4 scales = 128 items of four different costs, LPT over 8 ranks, maps on."""
    scales = [0.5, 0.75, 1.0, 1.25]
    single = ShardedPoseRunner(FakeNet()).run(_images64(32), scales, want_maps=True)
    item_poses, have, map_keys, checksum = _run_world(8, 32, scales)
    assert np.allclose(item_poses, single["item_poses"], atol=1e-12)
    assert map_keys == list(range(128)) and all(have)
    assert abs(checksum - sum(float(v["prob"].sum()) for v in single["maps"].values())) < 1e-6 * abs(checksum)


def test_world8_gloo_with_idle_ranks():
    """3 images at one scale on 8 ranks: five ranks are idle in every exchange round."""
    single = ShardedPoseRunner(FakeNet()).run(_images64(3), [1.0], want_maps=True)
    item_poses, have, map_keys, _c = _run_world(8, 3, [1.0])
    assert np.allclose(item_poses, single["item_poses"], atol=1e-12) and map_keys == [0, 1, 2] and have == [True] * 3


def test_interleaved_batches_and_group_units():
    """The grouped device pipeline's schedule is a pure function of the work list (every rank computes every rank's): batches
    round-robin over the (source size, scale) classes, units = runs of pairwise different shapes of at most group_size."""
    from deepcut_tools import group_units, plan_work, rank_batches

    shapes = [(336, 256)] * 32  # BASELINE configs[4]: 32 crops x 4 scales on one rank
    scales = [0.5, 0.75, 1.0, 1.25]
    items, shards = plan_work(shapes, scales, 1)
    plain = rank_batches(shapes, items, shards[0], 16)
    inter = rank_batches(shapes, items, shards[0], 16, interleave=True)
    assert sorted(tuple(b[3]) for b in plain) == sorted(tuple(b[3]) for b in inter)  # the same batches, another order
    assert [b[1] for b in plain] == [1.25, 1.25, 1.0, 1.0, 0.75, 0.75, 0.5, 0.5]
    assert [b[1] for b in inter] == [1.25, 1.0, 0.75, 0.5, 1.25, 1.0, 0.75, 0.5]     # largest first inside every round
    assert group_units(inter, 4) == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert group_units(plain, 4) == [[0], [1, 2], [3, 4], [5, 6], [7]]               # equal shapes never share a unit
    assert group_units(inter, 1) == [[i] for i in range(8)] and group_units(inter, 3) == [[0, 1, 2], [3, 4, 5], [6, 7]]
    # one scale only (configs[3]'s share): nothing to group, the plain path runs
    items1, shards1 = plan_work([(544, 736)] * 8, [1.0], 1)
    b1 = rank_batches([(544, 736)] * 8, items1, shards1[0], 4, interleave=True)
    assert group_units(b1, 4) == [[0], [1]]
    # a ragged last batch differs in batch size: it may join a unit with the full batches of other scales
    items2, shards2 = plan_work([(544, 736)] * 5, [1.0, 0.5], 1)
    b2 = rank_batches([(544, 736)] * 5, items2, shards2[0], 4, interleave=True)
    assert [(len(b[3]), b[1]) for b in b2] == [(4, 1.0), (4, 0.5), (1, 1.0), (1, 0.5)] and group_units(b2, 4) == [[0, 1, 2, 3]]
