"""-m gpu, TWO OR MORE GPUs (skipped on a 1-GPU box — where everything else of the suite runs): the first things a multi-GPU node
has to get right before a scaling curve means anything.

* `python bench.py --gpus 2 --backend nccl`, started the way the driver starts the N=1 line (no launcher, no WORLD_SIZE): bench.py
  becomes the launcher, two ranks on two distinct devices, the RCCL gather to rank 0 measured;
* dc_forward_batch over DC_COMM_RCCL with one executor per device — the grouped ncclRecv / ncclSend exchange with more than one
  executor, which no 1-GPU box can run — equals the single-executor maps; DC_COMM_AUTO ends on a transport that works (it probes the
  RCCL communicators at creation and falls back to peer copies); DC_COMM_PEER across devices (hipMemcpyPeerAsync).

The reference has nothing to mirror here (src/caffe/parallel.cpp:232,287-322 is training-only)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import caffe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(caffe.device_count() < 2, reason="needs two or more GPUs (device_count() = %d)" % caffe.device_count())]


def test_bench_gpus2_over_rccl_starts_itself():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", "--steps", "5", "--warmup", "2", "--regions", "1",
           "--no-cpu-baseline", "--no-f16-line", "--no-resnet101", "--coalesce", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    rep = d["config"]["distributed"]
    assert rep["backend"] == "nccl" and rep["ranks_seen"] == 2 and rep["distinct_devices"] == 2
    assert rep["gather"]["measured_gather_gbps_into_rank0"] is not None and rep["gather"]["measured_gather_gbps_into_rank0"] > 0


@pytest.fixture(scope="module")
def replicas(synth152):
    """one replica of the model per device (weights replicated per GPU, as in the one-process-per-GPU path)"""
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    caffe.set_mode_gpu()
    nets = []
    ndev = min(caffe.device_count(), 8)
    for dv in range(ndev):
        caffe.set_device(dv)
        nets.append(caffe.Net(deepercut_prototxt(152, 64, 80), path, caffe.TEST, from_text=True))
    caffe.set_device(0)
    return nets


SHAPES = [(64, 80), (48, 56), (64, 64), (56, 72)]


@pytest.mark.parametrize("transport", ["rccl", "peer", "auto"])
def test_in_process_forward_over_distinct_devices(replicas, transport):
    rs = np.random.RandomState(11)
    imgs = [(rs.randn(3, *SHAPES[i % 4]) * 50).astype(np.float32) for i in range(4 * len(replicas) + 3)]
    want = [{k: v.copy() for k, v in replicas[0].forward_batch(x[None]).items()} for x in imgs]
    comm = caffe.Comm(replicas, devices=list(range(len(replicas))), transport=transport)
    assert comm.transport in (("rccl", "peer") if transport == "auto" else (transport,))
    for rep in range(2):
        got = comm.forward(imgs)
        for i in range(len(imgs)):
            for k in want[i]:
                assert float(np.abs(got[i][k] - want[i][k][0]).max()) <= 1e-5 * max(1.0, float(np.abs(want[i][k]).max())), (transport, rep, i, k)
    # the gathered maps sit on the ROOT executor's device whichever executor computed them
    execs = {comm.executor_of(i) for i in range(len(imgs))}
    assert execs == set(range(len(replicas)))
    p, l, x, dims = comm.root_maps(len(imgs) - 1)
    assert p and l and x and dims[0] == 14 and dims[1] == 28 and dims[2] == 364
