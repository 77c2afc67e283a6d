"""CPU, world_size 2 over gloo: the N>1 path of bench.py — LPT sharding of work items and the gather of
variable-size score-map buffers to rank 0 (deepcut_tools/shard.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepcut_tools import lpt_shards


def test_lpt_shards_balance_and_cover():
    # BASELINE config 4: 64 equal images over 8 GPUs -> 8 each
    s = lpt_shards([1.0] * 64, 8)
    assert sorted(i for r in s for i in r) == list(range(64)) and all(len(r) == 8 for r in s)
    # config 5: 32 crops x 4 scales, cost ~ H*W
    sizes = [(168, 128), (256, 192), (336, 256), (424, 320)]
    costs = [h * w for _ in range(32) for (h, w) in sizes]
    s = lpt_shards(costs, 8)
    assert sorted(i for r in s for i in r) == list(range(128))
    loads = [sum(costs[i] for i in r) for r in s]
    assert max(loads) / (sum(loads) / 8.0) < 1.02
    assert lpt_shards([], 4) == [[], [], [], []]
    assert lpt_shards([5.0], 3) == [[0], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, n_items=7):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "..", "deepcut-cnn_amd", "python"))
    from deepcut_tools import gather_maps, gather_maps_known, lpt_shards as lpt

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank "forwards" its shard with a deterministic fake detector, then maps are gathered
        items = [(i, 8 * (2 + i % 3), 8 * (3 + i % 2)) for i in range(n_items)]
        shards = lpt([h * w for _, h, w in items], world)

        def fake_maps(i, h, w):
            g = torch.Generator().manual_seed(i)
            return torch.randn(406 * (h // 8) * (w // 8), generator=g)

        local = torch.cat([fake_maps(*items[i]) for i in shards[rank]]) if shards[rank] else torch.zeros(0)
        got = gather_maps(local, dst=0)
        sizes = [sum(406 * (items[i][1] // 8) * (items[i][2] // 8) for i in shards[r]) for r in range(world)]
        got2 = gather_maps_known(local.contiguous().view(-1), sizes, dst=0)
        if rank == 0:
            ok = True
            for r in range(world):
                exp = torch.cat([fake_maps(*items[i]) for i in shards[r]]) if shards[r] else torch.zeros(0)
                ok = ok and torch.equal(got[r].view(-1), exp) and torch.equal(got2[r].view(-1), exp)
            q.put(("ok" if ok else "mismatch", [int(t.numel()) for t in got]))
        else:
            assert got is None and got2 is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_maps_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    status, sizes = q.get(timeout=5)
    assert status == "ok" and len(sizes) == 2 and sum(sizes) > 0


def _spawn(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_items)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    return q.get(timeout=5)


def test_gather_maps_world8_gloo():
    """The design point: 8 ranks, rank 0 posts 7 grouped receives per exchange (shard.py), sizes from the LPT deal."""
    status, sizes = _spawn(8, 21)
    assert status == "ok" and len(sizes) == 8 and all(s > 0 for s in sizes)


def test_gather_maps_world8_gloo_with_idle_ranks():
    """5 items on 8 ranks: three ranks have nothing to send — no message is posted for them."""
    status, sizes = _spawn(8, 5)
    assert status == "ok" and len(sizes) == 8 and sorted(s > 0 for s in sizes) == [False] * 3 + [True] * 5
