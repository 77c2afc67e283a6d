"""CPU: `bench.py --gpus 8 --backend gloo --dry-run` — the N>1 control path of the bench (process group, rank identities in
config.distributed, the gather of payloads of the real run's size to rank 0, max-over-ranks timing) at the design point, world
size 8, on a box without a GPU; and the rule that refuses two ranks on one GPU under nccl."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world8_gloo_dry_run_of_the_fp16_config3_path():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--config", "3",
           "--dtype", "f16", "--dry-run", "--steps", "2", "--warmup", "1"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 8 and d["scaling"] == "weak"
    rep = d["config"]["distributed"]
    assert rep["backend"] == "gloo" and rep["ranks_seen"] == 8
    assert sorted(i["rank"] for i in rep["ranks"]) == list(range(8)) and len(set(i["pid"] for i in rep["ranks"])) == 8
    # fp16 payload of 8 images of 736x544: (14 + 28 + 364) channels x 68 x 92 cells x 8 images x 2 bytes
    assert rep["gather"]["payload_bytes_per_rank_per_step"] == 406 * 68 * 92 * 8 * 2
    assert rep["gather"]["bytes_into_rank0_per_step"] == 7 * rep["gather"]["payload_bytes_per_rank_per_step"]
    assert rep["gather"]["measured_gather_gbps_into_rank0"] > 0


def test_two_ranks_on_one_gpu_are_refused_under_nccl():
    sys.path.insert(0, ROOT)
    import bench

    a = {"rank": 0, "host": "n0", "uuid": "GPU-aa", "pci": "0000:05:00"}
    b = {"rank": 1, "host": "n0", "uuid": "GPU-bb", "pci": "0000:15:00"}
    bench.one_gpu_per_rank([a, b], "nccl")
    bench.one_gpu_per_rank([a, dict(a, rank=1)], "gloo")  # the one-GPU smoke tests share a device on purpose
    with pytest.raises(SystemExit):
        bench.one_gpu_per_rank([a, dict(a, rank=1)], "nccl")
    with pytest.raises(SystemExit):
        bench.one_gpu_per_rank([dict(a, uuid=""), dict(a, uuid="", rank=1)], "nccl")  # no uuid: the PCI address decides


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_starts_itself_when_invoked_like_the_n1_line():
    """`python3 bench.py --gpus 8 ...` with no launcher around it (no WORLD_SIZE): bench.py becomes the launcher
    (torch.distributed.run, one rank per GPU, 127.0.0.1) instead of exiting; rank 0 prints ONE JSON line, rc 0."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8
    rep = d["config"]["distributed"]
    assert rep["ranks_seen"] == 8 and len(set(i["pid"] for i in rep["ranks"])) == 8


def test_self_launch_hands_back_a_failing_rank():
    """a child that dies (here: a net input no layer stack survives) makes the launcher's exit code non-zero"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--height", "-8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
