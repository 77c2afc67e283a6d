"""CPU: the host-only parts of round 5's additions to the C ABI — the longest-processing-time-first schedule dc_forward_batch deals
images by (dc_lpt_schedule == deepcut_tools.lpt_shards, the schedule of the one-process-per-GPU path), argument checking of
dc_comm_* / dc_nets_choose_streams / dc_net_forward_host_async without a GPU (refused loudly, never computed elsewhere), and the
result-array pool of the pycaffe shim (weak references, no reference-count arithmetic)."""
import gc
import os

import numpy as np
import pytest


def test_lpt_schedule_equals_the_per_rank_schedule_of_the_distributed_path():
    import caffe
    from deepcut_tools import lpt_shards

    rs = np.random.RandomState(5)
    for n, k in ((0, 3), (1, 1), (7, 3), (64, 8), (128, 8), (33, 5), (5, 8)):
        costs = [float(c) for c in rs.randint(1, 50, n) * 64]
        assert caffe.lpt_schedule(costs, k) == lpt_shards(costs, k), (n, k)
    # BASELINE configs[3]: 64 equal images on 8 executors -> 8 each, round robin
    shares = caffe.lpt_schedule([544.0 * 736] * 64, 8)
    assert [len(s) for s in shares] == [8] * 8 and shares[0] == list(range(0, 64, 8))
    # configs[4]: 32 crops x 4 scales: the shares' loads differ by less than one largest item
    hw = [(168, 128), (256, 192), (336, 256), (424, 320)]
    costs = [float(h * w) for _ in range(32) for h, w in hw]
    shares = caffe.lpt_schedule(costs, 8)
    loads = [sum(costs[i] for i in s) for s in shares]
    assert max(loads) - min(loads) <= max(costs) and sorted(i for s in shares for i in s) == list(range(128))
    with pytest.raises(caffe.DeepcutError):
        caffe.lpt_schedule([1.0], 0)


def test_multi_gpu_entries_refuse_without_a_device_or_with_bad_arguments():
    import caffe
    import caffe.pycaffe as pc
    from deepcut_tools import deepercut_prototxt

    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    if caffe.device_count() == 0:
        with pytest.raises(caffe.DeepcutError) as e:
            caffe.Comm([net])
        assert "no HIP device" in str(e.value)
        with pytest.raises(caffe.DeepcutError):
            caffe.pinned_empty((4,))
    caffe.set_mode_cpu()
    with pytest.raises(caffe.DeepcutError):  # no lowered shape, CPU mode: refused either way
        caffe.choose_streams([net])
    x = np.zeros((1, 3, 64, 64), np.float32)
    with pytest.raises(caffe.DeepcutError):
        net.forward_host_async(x)
    with pytest.raises(ValueError):
        net.forward_host_async(x.astype(np.float64))
    assert pc._lib.dc_comm_destroy(None) == 0 and pc._lib.dc_host_free(None) == 0
    assert pc._lib.dc_comm_transport(None) < 0 and pc._lib.dc_comm_item_executor(None, 0) < 0
    assert pc._lib.dc_nets_choose_streams(None, 1, 0, 0, None, None) < 0
    assert pc._lib.dc_forward_batch(None, None, 1, None, None, 0, None, None, None) < 0


def test_result_arrays_are_recycled_only_when_no_view_is_left():
    from caffe.pycaffe import _OutPool

    pool = _OutPool()
    a = pool.take((2, 3, 4))
    a[...] = 7
    view = a[1, :, 2]
    b = pool.take((2, 3, 4))
    assert not np.shares_memory(a, b)
    del a
    gc.collect()
    c = pool.take((2, 3, 4))  # a view of the first hand-out is alive: its memory must not come back
    assert not np.shares_memory(c, view) and float(view[0]) == 7
    addr = view.__array_interface__["data"][0]
    del view
    gc.collect()
    d = pool.take((2, 3, 4))  # now it does
    lo = d.__array_interface__["data"][0]
    assert lo <= addr < lo + d.nbytes
    assert len(pool.entries) <= pool.keep
    for _ in range(10):  # holding everything: the pool stops growing, new memory is simply not pooled
        pool.take((2, 3, 4))
    assert len(pool.entries) <= pool.keep


def test_set_tile_keeps_what_the_tune_cache_file_already_held(tmp_path, monkeypatch):
    """Round-4 advice: write_tune_cache rewrote DC_TUNE_CACHE from the in-memory table, and the file was only loaded inside the
    autotuner — a process that only ever overrides a tile (or runs with DC_AUTOTUNE=0) truncated an existing cache to its one
    override.  The file is now loaded before any write, united with the table (the table wins) and replaced atomically."""
    import caffe
    from deepcut_tools import deepercut_prototxt

    cache = tmp_path / "tune.txt"
    names = [name for name, _es in caffe.conv_variants()]
    old = ["f999/64/64/1x1/s1/r0 %s" % names[1], "some/other/model/signature %s" % names[2], "G4:a|b|c|d %s" % names[0], "3x3/eligible wino_f23"]
    cache.write_text("\n".join(old) + "\n")
    monkeypatch.setenv("DC_TUNE_CACHE", str(cache))
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    net.plan_text()  # lower (host only)
    rep = net.tune_report()
    tiles = [name for name, esize in caffe.conv_variants() if esize == 4]
    done = None
    for r in rep:
        for t in tiles:
            if t == r["tile"]:
                continue
            try:
                net.set_tile(r["signature"], t)
                done = (r["signature"], t)
                break
            except caffe.DeepcutError:
                continue
        if done:
            break
    assert done, "no signature took another tile"
    lines = [ln for ln in cache.read_text().splitlines() if ln.strip()]
    for ln in old:
        assert ln in lines, ln  # nothing the file held was lost
    assert "%s %s" % done in lines
    assert not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]  # the temporary file was renamed into place
    # a second model in the same process: the file wins only where the process knows nothing
    net2 = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    net2.plan_text()
    net2.set_tile(done[0], [r for r in rep if r["signature"] == done[0]][0]["tile"])
    lines = [ln for ln in cache.read_text().splitlines() if ln.strip()]
    assert len([ln for ln in lines if ln.startswith(done[0] + " ")]) == 1 and all(ln in lines for ln in old)


def test_host_async_entry_checks_every_array_against_the_nets_shapes():
    """forward_host_async hands raw host pointers to asynchronous DMA: a wrongly shaped array must be refused in Python, before the
    library reads n*C*h*w floats or writes a whole map through it (the checks are host-only shape inference: no GPU needed)"""
    import caffe
    from deepcut_tools import deepercut_prototxt

    net = caffe.Net(deepercut_prototxt(152, 64, 80), caffe.TEST, from_text=True)
    x = np.zeros((1, 3, 64, 80), np.float32)
    prob = np.zeros((1, 14, 8, 10), np.float32)
    with pytest.raises(ValueError, match="channels"):
        net.forward_host_async(np.zeros((1, 4, 64, 80), np.float32), prob=prob)
    with pytest.raises(ValueError, match="prob has"):
        net.forward_host_async(x, prob=np.zeros((1, 14, 8, 9), np.float32))
    with pytest.raises(ValueError, match="next_pred has"):
        net.forward_host_async(x, next_pred=np.zeros((1, 28, 8, 10), np.float32))
    with pytest.raises(ValueError, match="batch"):
        net.forward_host_async(np.zeros((3, 64, 80), np.float32))
    with pytest.raises(ValueError, match="C-contiguous"):
        net.forward_host_async(x, prob=np.zeros((1, 14, 8, 20), np.float32)[..., ::2])
    # well-formed arrays pass the checks and reach the library, which refuses for ITS reason (CPU mode / no device), not a shape
    with pytest.raises(caffe.DeepcutError):
        net.forward_host_async(x, prob=prob, loc_pred=np.zeros((1, 28, 8, 10), np.float32), next_pred=np.zeros((1, 364, 8, 10), np.float32))


def test_output_selection_on_the_host(tmp_path):
    """DC_OPT_OUTPUTS is a property of the lowering (host only): launches, FLOPs and the refusal of bad selections"""
    import caffe
    from deepcut_tools import deepercut_prototxt

    net = caffe.Net(deepercut_prototxt(152, 544, 736), caffe.TEST, from_text=True)
    f_all, n_all = net.flops(), len(net.plan_text().splitlines())
    net.set_outputs(["prob", "loc_pred"])
    assert net.wanted_outputs == ["loc_pred", "prob"] and net.outputs == ["loc_pred", "next_pred", "prob"]
    t = net.plan_text()
    assert "next" not in t and " N=42 " in t and len(t.splitlines()) == n_all  # still one skip GEMM + one merged deconvolution
    assert abs((f_all - net.flops()) - (2.0 * 2048 * 34 * 46 * 364 * 9 + 2.0 * 364 * 68 * 92 * 512)) < 1e3
    clone = net.clone()
    assert clone.wanted_outputs == ["loc_pred", "prob"]  # executors of one model agree on what they compute
    net.set_outputs(["next_pred"])
    assert "res5c_up_pose" not in net.plan_text() and "res5c_up_next" in net.plan_text()
    net.set_outputs(None)
    assert net.flops() == f_all
    with pytest.raises(ValueError):
        net.set_outputs(["res5c"])
    with pytest.raises(caffe.DeepcutError):
        net.set_option(4, 0)
    with pytest.raises(caffe.DeepcutError):
        net.set_option(4, 8)
