"""-m gpu: the tile choices of a shape seen from outside (dc_net_tune_report) and overridden (dc_net_set_tile) — what
deepcut_tools.tune_in_flight drives — leave the results where they were and are visible in the plan."""
import os

import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu
H, W = 72, 104


def _net(gpu_caffe, synth152, dtype="f32"):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, H, W), path, gpu_caffe.TEST, from_text=True, dtype=dtype)
    net.blobs["data"].data[...] = rand_image(11, H, W)
    net.forward()
    return net


def test_tune_report_lists_every_signature_with_its_timings(gpu_caffe, synth152, monkeypatch):
    monkeypatch.delenv("DC_TUNE_CACHE", raising=False)
    net = _net(gpu_caffe, synth152)
    rep = net.tune_report()
    names = set(n for n, _ in gpu_caffe.conv_variants()) | {"wino_f23", "wino_f23_w16", "ws1x1f", "ws7x7f"}  # (float32: the forms outside the tile table)
    assert len(rep) >= 20 and sum(r["launches"] for r in rep) == sum(
        1 for ln in net.plan_text().splitlines() if "conv_gemm<" in ln or "wino_f23<" in ln or "ws1x1f<" in ln or "ws7x7f<" in ln)
    for r in rep:
        assert r["tile"] in names and r["launches"] >= 1
        assert r["timed"], r  # tuned in this process: every signature carries the isolated timings, fastest first
        assert [us for _, us in r["timed"]] == sorted(us for _, us in r["timed"])
        assert all(t in names for t, _ in r["timed"])


def test_set_tile_changes_the_plan_not_the_maps(gpu_caffe, synth152, monkeypatch):
    monkeypatch.delenv("DC_TUNE_CACHE", raising=False)
    net = _net(gpu_caffe, synth152)
    clone = net.clone()
    ref = {k: net.blobs[k].data.copy() for k in ("prob", "loc_pred", "next_pred")}
    changed = 0
    for r in net.tune_report():
        others = [t for t, _ in r["timed"] if t != r["tile"]]
        if not others or changed >= 6:
            continue
        net.set_tile(r["signature"], others[-1])  # the slowest eligible one
        changed += 1
        now = [q for q in net.tune_report() if q["signature"] == r["signature"]][0]
        assert now["tile"] == others[-1]
    assert changed >= 3
    net.forward()
    for k in ref:  # other tiles sum in another order: float32 rounding only
        assert float(np.abs(net.blobs[k].data - ref[k]).max()) <= 2e-4, k
    # the choice table is shared: a clone lowering the shape now takes the overridden tiles
    clone.blobs["data"].data[...] = net.blobs["data"].data
    clone.forward()
    assert [q["tile"] for q in clone.tune_report()] == [q["tile"] for q in net.tune_report()]


def test_set_tile_refuses_what_cannot_run(gpu_caffe, synth152):
    net = _net(gpu_caffe, synth152)
    sig = net.tune_report()[0]["signature"]
    with pytest.raises(gpu_caffe.DeepcutError):
        net.set_tile(sig, "no_such_tile")
    with pytest.raises(gpu_caffe.DeepcutError):
        net.set_tile("1/2/3/4/1x1/1,1/0/5", net.tune_report()[0]["tile"])
    half = [n for n, es in gpu_caffe.conv_variants() if es == 2][0]
    with pytest.raises(gpu_caffe.DeepcutError):
        net.set_tile(sig, half)  # a float16 tile on a float32 net


def test_tune_in_flight_keeps_or_improves(gpu_caffe, synth152, monkeypatch):
    import time

    from deepcut_tools import tune_in_flight

    monkeypatch.delenv("DC_TUNE_CACHE", raising=False)
    net = _net(gpu_caffe, synth152)
    nets = [net, net.clone()]
    x = rand_image(11, H, W)
    for n in nets:
        n.blobs["data"].data[...] = x
        n.forward()
    ref = net.blobs["prob"].data.copy()

    def load():
        t0 = time.perf_counter()
        for _ in range(3):
            for n in nets:
                n.forward()
        return time.perf_counter() - t0

    res = tune_in_flight(nets, load, top=3, reps=1)
    assert res["after"] <= res["before"] * 1.5 and res["runs"] >= 2
    net.forward()
    assert float(np.abs(net.blobs["prob"].data - ref).max()) <= 2e-4


def test_pipeline_tune_runs_its_own_load(gpu_caffe, synth152, monkeypatch):
    import torch

    from deepcut_tools import Pipeline

    monkeypatch.delenv("DC_TUNE_CACHE", raising=False)
    net = _net(gpu_caffe, synth152)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(rand_image(11, H, W)).to(dev)
    ref = net.blobs["prob"].data.copy()
    pipe = Pipeline(net, depth=2, max_batch=2)
    with pytest.raises(ValueError):
        pipe.tune([(x.data_ptr(), 1, H, W, None, None, None)] * 3)  # not a multiple of batch size x depth
    outs = [[torch.empty(1, c, H // 8, W // 8, device=dev) for c in (14, 28, 364)] for _ in range(4)]
    reqs = [(x.data_ptr(), 1, H, W, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr()) for o in outs]
    res = pipe.tune(reqs, rounds=2, top=3, reps=1)
    assert res["runs"] >= 2 and res["after"] <= res["before"] * 1.5
    pipe.submit(*reqs[0], tag="t")
    assert pipe.drain() == ["t"]
    torch.cuda.synchronize(dev)
    assert float(np.abs(outs[0][0].cpu().numpy() - ref).max()) <= 2e-4
