"""-m gpu: round 5's additions to the C ABI on the device —
* dc_nets_choose_streams: executors adopt pool streams chosen by timing their real forwards; results unchanged, streams distinct,
  a Pipeline picks its streams itself;
* dc_net_forward_host_async / Pipeline.submit_host: host-in / host-out requests in flight (pinned and pageable buffers) equal the
  synchronous entry;
* dc_comm_create / dc_forward_batch: the in-process multi-executor forward — 1 executor over RCCL (librccl.so by dlopen: all that
  a 1-GPU box can run of that transport), 8 executors sharing GPU 0 over the loop-back (peer-copy) transport with images of four
  shapes — equals the single-executor results; the C++ facade's ForwardPool drives the same path from a g++-built program."""
import os
import subprocess

import numpy as np
import pytest

from conftest import rand_image
from test_cxx_facade import _compile

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def base_net(gpu_caffe, synth152):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    return gpu_caffe.Net(deepercut_prototxt(152, 64, 80), path, gpu_caffe.TEST, from_text=True)


def _maps(net, img):
    return {k: v.copy() for k, v in net.forward_batch(img).items()}


def test_executors_adopt_measured_streams(gpu_caffe, base_net):
    nets = [base_net] + [base_net.clone() for _ in range(3)]
    img = rand_image(4, 64, 80)
    want = _maps(base_net, img)
    with pytest.raises(gpu_caffe.DeepcutError):  # the clones have no lowered shape yet
        gpu_caffe.choose_streams(nets)
    for n in nets:
        n.reserve(1, 64, 80)
    r = gpu_caffe.choose_streams(nets, candidates=8, reps=2)
    assert r["forwards_per_s_chosen"] > 0 and r["forwards_per_s_first_created"] > 0
    handles = [n.stream_handle() for n in nets]
    assert len(set(handles)) == 4 and all(handles)
    for n in nets:  # same results on the adopted streams, synchronous and asynchronous entries
        got = _maps(n, img)
        for k in want:
            assert np.array_equal(got[k], want[k]), k
    again = gpu_caffe.choose_streams(nets[:2], reps=1)  # choosing again (a subset) releases and re-adopts
    assert again["forwards_per_s_chosen"] > 0 and nets[0].stream_handle() != nets[1].stream_handle()
    del nets[1:]  # executors go, their pool streams stay valid for the next user
    other = [base_net, base_net.clone()]
    for n in other:
        n.reserve(1, 64, 80)
    gpu_caffe.choose_streams(other)
    assert np.array_equal(_maps(other[1], img)["prob"], want["prob"])


def test_host_requests_in_flight_equal_the_synchronous_entry(gpu_caffe, base_net):
    from deepcut_tools import Pipeline

    imgs = [rand_image(20 + i, 64, 80) for i in range(6)]
    want = [_maps(base_net, x) for x in imgs]
    shp = {k: want[0][k].shape for k in want[0]}
    pipe = Pipeline(base_net, depth=3, coalesce=1)
    slots = []
    for i, x in enumerate(imgs):
        pinned = i % 2 == 0  # pinned and pageable buffers alike
        mk = gpu_caffe.pinned_empty if pinned else (lambda s: np.empty(s, np.float32))
        xi = mk(x.shape)
        xi[...] = x
        slots.append((xi, {k: mk(shp[k]) for k in shp}))
    for rep in range(3):
        for i, (xi, o) in enumerate(slots):
            for a in o.values():
                a[...] = -1
            pipe.submit_host(xi, o["prob"], o["loc_pred"], o["next_pred"], tag=i)
        assert sorted(pipe.drain()) == list(range(len(slots)))
        for i, (_xi, o) in enumerate(slots):
            for k in shp:
                assert np.array_equal(o[k], want[i][k]), (rep, i, k)
    assert pipe.stream_choice is not None and pipe.stream_choice["forwards_per_s_chosen"] > 0
    # outputs may be skipped; a pinned array outlives the call that made it and is freed with its last view
    xi, o = slots[0]
    o["prob"][...] = -1
    pipe.nets[0].forward_host_async(xi, prob=o["prob"])
    pipe.nets[0].synchronize()
    assert np.array_equal(o["prob"], want[0]["prob"])
    v = gpu_caffe.pinned_empty((3, 5))[1]
    v[...] = 2.0
    assert float(v.sum()) == 10.0


SHAPES = [(64, 80), (48, 56), (64, 64), (56, 72)]


def test_in_process_multi_executor_forward(gpu_caffe, base_net):
    rs = np.random.RandomState(9)
    shapes5 = SHAPES + [(60, 76)]  # not a multiple of 8: the maps are 8 x 10 (ceil-mode pooling), not 60 // 8 x 76 // 8
    imgs = [(rs.randn(3, *shapes5[i % 5]) * 50).astype(np.float32) for i in range(19)]
    want = [_maps(base_net, x[None]) for x in imgs]
    assert want[4]["prob"].shape == (1, 14, 8, 10)
    # one executor, RCCL transport: dlopen + ncclCommInitAll on the one device; nothing to exchange
    c1 = gpu_caffe.Comm([base_net], transport="rccl")
    assert c1.transport == "rccl"
    got = c1.forward(imgs[:5])
    for i in range(5):
        for k in want[i]:
            assert got[i][k].shape == want[i][k][0].shape and float(np.abs(got[i][k] - want[i][k][0]).max()) <= 1e-5, (i, k)
    # eight executors sharing GPU 0, loop-back transport: schedule, threads, batches per shape, gather, scatter
    nets = [base_net] + [base_net.clone() for _ in range(7)]
    c8 = gpu_caffe.Comm(nets, devices=[0] * 8, transport="peer")
    assert c8.transport == "peer"
    for rep in range(2):
        got = c8.forward(imgs)
        for i in range(len(imgs)):
            for k in want[i]:
                # an executor forwards its same-shape images as ONE batch: another tile may sum in another order
                assert float(np.abs(got[i][k] - want[i][k][0]).max()) <= 1e-5 * max(1.0, float(np.abs(want[i][k]).max())), (rep, i, k)
    # pinned arrays of the caller travel in place (no staging copy in, no scatter copy out): the same bits
    pimgs = []
    for x in imgs:
        a = gpu_caffe.pinned_empty(x.shape)
        a[...] = x
        pimgs.append(a)
    gotp = c8.forward(pimgs, pinned=True)
    mixed = c8.forward([pimgs[i] if i % 3 else imgs[i] for i in range(len(imgs))], pinned=False)  # pinned and pageable arrays in one call
    for i in range(len(imgs)):
        for k in want[i]:
            assert np.array_equal(gotp[i][k], got[i][k]) and np.array_equal(mixed[i][k], got[i][k]), (i, k)
    shares = gpu_caffe.lpt_schedule([float(x.shape[1] * x.shape[2]) for x in imgs], 8)
    for kx, share in enumerate(shares):
        assert all(c8.executor_of(i) == kx for i in share)
    p, l, x, dims = c8.root_maps(3)
    assert p and l and x and dims == [14, 28, 364, SHAPES[3][0] // 8, SHAPES[3][1] // 8]
    assert c8.root_maps(4)[3] == [14, 28, 364, 8, 10] and got[4]["next_pred"].shape == (364, 8, 10)
    # two host threads on ONE communicator: the calls are serialised inside the library (ctypes drops the GIL), both complete
    import threading

    res = [None, None]

    def call(j):
        res[j] = c8.forward(imgs[j::2])

    ths = [threading.Thread(target=call, args=(j,)) for j in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for j in range(2):
        for got_i, i in zip(res[j], range(j, len(imgs), 2)):
            for k in want[i]:
                assert float(np.abs(got_i[k] - want[i][k][0]).max()) <= 1e-5 * max(1.0, float(np.abs(want[i][k]).max())), (j, i, k)
    assert c8.forward([]) == []
    with pytest.raises(gpu_caffe.DeepcutError):
        c8.root_maps(0)  # an empty call is the last forward now: nothing to point at
    with pytest.raises(gpu_caffe.DeepcutError):
        gpu_caffe.Comm([base_net, base_net], devices=[0, 0], transport="peer").forward(imgs[:2])  # one net, two executors
    with pytest.raises(gpu_caffe.DeepcutError):
        gpu_caffe.Comm(nets[:2], devices=[0, 0], transport="rccl")  # RCCL wants one executor per device
    auto = gpu_caffe.Comm(nets[:2], devices=[0, 0])
    assert auto.transport == "peer"


POOL_SRC = r'''
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "caffe_facade.hpp"
using namespace caffe;
// argv: prototxt caffemodel dir n  — images dir/in<i>.bin of shapes given in dir/shapes.txt; writes dir/pool<i>.bin (prob|loc|next)
int main(int argc, char** argv) {
  Caffe::set_mode(Caffe::GPU);
  Caffe::SetDevice(0);
  Net<float> net(argv[1], TEST);
  net.CopyTrainedLayersFrom(argv[2]);
  std::vector<shared_ptr<Net<float> > > clones;
  std::vector<Net<float>*> execs(1, &net);
  for (int k = 1; k < 8; ++k) {
    clones.push_back(net.Clone());
    execs.push_back(clones.back().get());
  }
  const std::string dir = argv[3];
  const int n = std::atoi(argv[4]);
  std::vector<std::pair<int, int> > hw;
  std::vector<std::vector<float> > imgs;
  FILE* f = std::fopen((dir + "/shapes.txt").c_str(), "r");
  for (int i = 0; i < n; ++i) {
    int h, w;
    if (std::fscanf(f, "%d %d", &h, &w) != 2) return 2;
    hw.push_back(std::make_pair(h, w));
    imgs.push_back(std::vector<float>((size_t)3 * h * w));
    FILE* g = std::fopen((dir + "/in" + std::to_string(i) + ".bin").c_str(), "rb");
    if (std::fread(imgs.back().data(), 4, imgs.back().size(), g) != imgs.back().size()) return 3;
    std::fclose(g);
  }
  std::fclose(f);
  std::vector<const float*> ptr;
  for (auto& v : imgs) ptr.push_back(v.data());
  ForwardPool pool(execs, std::vector<int>(8, 0), DC_COMM_PEER);
  std::vector<ForwardPool::Maps> out = pool.Forward(ptr, hw);
  for (int i = 0; i < n; ++i) {
    FILE* g = std::fopen((dir + "/pool" + std::to_string(i) + ".bin").c_str(), "wb");
    std::fwrite(out[i].prob.data(), 4, out[i].prob.size(), g);
    std::fwrite(out[i].loc_pred.data(), 4, out[i].loc_pred.size(), g);
    std::fwrite(out[i].next_pred.data(), 4, out[i].next_pred.size(), g);
    std::fclose(g);
  }
  std::printf("executors %d %d %d transport %d OK\n", pool.executor_of(0), pool.executor_of(1), pool.executor_of(n - 1), pool.transport());
  return 0;
}
'''


def test_cxx_facade_drives_eight_executors(tmp_path, gpu_caffe, synth152, base_net):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    proto = tmp_path / "net.prototxt"
    proto.write_text(deepercut_prototxt(152, 64, 80))
    rs = np.random.RandomState(12)
    n = 11
    shapes = SHAPES[:3] + [(60, 76)]
    imgs = [(rs.randn(3, *shapes[i % 4]) * 50).astype(np.float32) for i in range(n)]
    with open(str(tmp_path / "shapes.txt"), "w") as f:
        for i, x in enumerate(imgs):
            f.write("%d %d\n" % x.shape[1:])
            x.tofile(str(tmp_path / ("in%d.bin" % i)))
    exe = _compile(tmp_path, "facade_pool", POOL_SRC, hip=True)
    out = subprocess.run([exe, str(proto), path, str(tmp_path), str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
    assert "transport 2" in out.stdout
    for i, x in enumerate(imgs):
        want = _maps(base_net, x[None])
        got = np.fromfile(str(tmp_path / ("pool%d.bin" % i)), np.float32)
        ref = np.concatenate([want[k].ravel() for k in ("prob", "loc_pred", "next_pred")])
        assert got.shape == ref.shape and float(np.abs(got - ref).max()) <= 1e-5 * max(1.0, float(np.abs(ref).max())), i


def test_communicators_and_pipelines_leave_nothing_behind(gpu_caffe, base_net):
    """Create / run / destroy cycles of the round-5 objects: 8-executor communicators (threads, communication streams, events, send /
    receive / pinned buffers), executor sets that adopt pool streams, host pipelines with pinned arrays.  The first cycle pays for
    what is process-wide by design (the stream pool, librccl.so); from the second on device memory, threads and file descriptors
    may not grow."""
    import gc
    import threading

    import torch

    from deepcut_tools import Pipeline

    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(3)
    imgs = [(rs.randn(3, *SHAPES[i % 4]) * 50).astype(np.float32) for i in range(9)]

    def cycle():
        nets = [base_net] + [base_net.clone() for _ in range(7)]
        comm = gpu_caffe.Comm(nets, devices=[0] * 8, transport="peer")
        comm.forward(imgs)
        c1 = gpu_caffe.Comm([base_net], transport="rccl")
        c1.forward(imgs[:2])
        pipe = Pipeline(nets[1], depth=3, coalesce=1)
        x = gpu_caffe.pinned_empty((1, 3, 64, 80))
        x[...] = rand_image(1, 64, 80)
        outs = [gpu_caffe.pinned_empty(base_net.blobs[k].shape) for k in ("prob", "loc_pred", "next_pred")]
        base_net.blobs["data"].reshape(1, 3, 64, 80)
        base_net.reshape()
        outs = [gpu_caffe.pinned_empty(tuple(base_net.blobs[k].shape)) for k in ("prob", "loc_pred", "next_pred")]
        for i in range(4):
            pipe.submit_host(x, *outs, tag=i)
            pipe.drain()
        del comm, c1, pipe, nets, x, outs
        gc.collect()
        torch.cuda.synchronize(dev)

    rows = []
    for c in range(5):
        cycle()
        rows.append((torch.cuda.mem_get_info(dev)[0] / 2 ** 20, threading.active_count(), len(os.listdir("/proc/self/fd")),
                     len(os.listdir("/proc/self/task"))))
    assert rows[1][0] - rows[-1][0] <= 2.0, rows      # device memory (MiB)
    assert rows[-1][2] <= rows[1][2], rows            # file descriptors
    assert rows[-1][3] <= rows[1][3], rows            # native threads (the communicators' workers are joined)


def test_config3_through_the_in_process_path(gpu_caffe, synth152):
    """BASELINE configs[3] at its real size through dc_forward_batch: 64 images of 544x736 (seeds 100..163) on 8 executors — which
    share GPU 0 here; on an 8-GPU node executor k is a replica on device k and the same call gathers over RCCL —, 8 per executor
    by the LPT schedule, each share forwarded as TWO sub-batches of 4 (the executor's pipeline: staging of the second under the forward
    of the first, the way back of the first under the forward of the second).  Checked: executor 0's share against the single net's two
    batch-4 forwards (themselves within 1e-3 of the oracle: tests/test_gpu_configs.py), every image against itself across two calls (determinism), and the
    maps of an image landing in that image's slot (the input seeds make every image distinct: a permutation would show)."""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    net = gpu_caffe.Net(deepercut_prototxt(152, 544, 736, 8), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    nets = [net] + [net.clone() for _ in range(7)]
    imgs = [(np.random.RandomState(100 + i).randn(3, 544, 736) * 50).astype(np.float32) for i in range(64)]
    comm = gpu_caffe.Comm(nets, devices=[0] * 8, transport="peer")
    got = comm.forward(imgs)
    shares = gpu_caffe.lpt_schedule([544.0 * 736] * 64, 8)
    assert [len(s) for s in shares] == [8] * 8 and all(comm.executor_of(i) == k for k, s in enumerate(shares) for i in s)
    for half in (shares[0][:4], shares[0][4:]):
        want = {k: v.copy() for k, v in net.forward_batch(np.stack([imgs[i] for i in half])).items()}
        for b, i in enumerate(half):
            for k in ("prob", "loc_pred", "next_pred"):
                assert got[i][k].shape == want[k][b].shape
                assert np.array_equal(got[i][k], want[k][b]), (i, k)  # same net, same sub-batch, same tiles: bit-identical
    again = comm.forward(imgs)
    single = gpu_caffe.Net(deepercut_prototxt(152, 544, 736, 1), path, gpu_caffe.TEST, from_text=True)
    for i in (1, 17, 42, 63):  # images of other executors: against a batch-1 forward of that image (another tile may sum in another order)
        ref = single.forward_batch(imgs[i][None])
        for k in ref:
            assert np.array_equal(got[i][k], again[i][k])
            assert float(np.abs(got[i][k] - ref[k][0]).max()) <= 1e-5 * max(1.0, float(np.abs(ref[k]).max())), (i, k)


def test_pipeline_tunes_for_its_own_load_when_a_tune_cache_is_set(gpu_caffe, synth152, tmp_path, monkeypatch):
    """DC_TUNE_CACHE set: the first device-resident request shape a Pipeline meets is re-tuned under the pipeline's own load (what
    bench.py does for `value`), the overrides land in the cache file, the side-car lists the shape, and a second pipeline in the
    same or a later process does not tune again; results are the forward's own."""
    import json
    import torch
    from deepcut_tools import Pipeline, deepercut_prototxt

    path, _ = synth152
    cache = tmp_path / "tune.txt"
    monkeypatch.setenv("DC_TUNE_CACHE", str(cache))
    net = gpu_caffe.Net(deepercut_prototxt(152, 64, 80), path, gpu_caffe.TEST, from_text=True, hipgraph=1)
    dev = torch.device("cuda", 0)
    x = (torch.randn(1, 3, 64, 80) * 50).to(dev)
    want = net.forward_batch(x.cpu().numpy())
    pipe = Pipeline(net, depth=2, max_batch=2)
    outs = [[torch.empty(want[k].shape, device=dev) for k in ("prob", "loc_pred", "next_pred")] for _ in range(6)]
    for i, o in enumerate(outs):
        pipe.submit(x.data_ptr(), 1, 64, 80, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), tag=i)
    assert sorted(pipe.drain()) == list(range(6))
    (key, rep), = pipe.auto_tune_report.items()
    assert "64x80" in key and "depth2" in key and rep["runs"] > 0 and rep["after"] <= rep["before"] * 1.05
    assert key in json.load(open(str(cache) + ".inflight")) and cache.exists()
    for o in outs:
        for t, k in zip(o, ("prob", "loc_pred", "next_pred")):
            assert float((t.cpu() - torch.from_numpy(want[k])).abs().max()) <= 1e-5 * max(1.0, float(np.abs(want[k]).max())), k
    again = Pipeline(net, depth=2, max_batch=2)
    again.submit(x.data_ptr(), 1, 64, 80, outs[0][0].data_ptr(), outs[0][1].data_ptr(), outs[0][2].data_ptr(), tag=0)
    again.drain()
    assert again.auto_tune_report == {}  # the side-car says it is done
    monkeypatch.delenv("DC_TUNE_CACHE")
    assert Pipeline(net, depth=2)._auto_tune is False
