#!/usr/bin/env python
"""bench.py — DeeperCut part-detector forward throughput on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one forward of the hot path (ResNet-152 FCN + deconvolution heads -> prob / loc_pred /
next_pred) over one batch of synthetic input already resident in HBM, on every rank, followed (N>1)
by the RCCL gather of the three score maps to rank 0.  Workload at N=1 = BASELINE.json configs[1]:
batch=1, 1x3x544x736, fp32 (the shipped prototxt is ResNet-152 — SURVEY F1 — not the "ResNet-101"
of the config string).  Weak scaling: every rank forwards its own image each step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "deepcut-cnn_amd")
for p in (ROOT, PKG, os.path.join(PKG, "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def inject_weights(net, layers):
    for name, _typ, blobs in layers:
        ps = net.params[name]
        for p, b in zip(ps, blobs):
            p.data[...] = b


def cpu_baseline(proto_fn, layers, flops_full, full_hw):
    """The reference's CPU algorithm (oracle/: im2col+SGEMM, unfused layers) timed on this host's cores,
    on a bounded sample: one 240x320 forward (BASELINE configs[0] size) with all cores, one 104x136
    forward single-threaded.  Reported as images/s of the FULL workload by FLOP scaling (the path's
    cost is linear in H*W: SURVEY §8a T1)."""
    import numpy as np
    from oracle import oracle as O

    res = {}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(cores, O.lib().oracle_max_threads())
    for tag, (h, w), nt in (("all", (240, 320), threads), ("one", (104, 136), 1)):
        O.set_threads(nt)
        proto = proto_fn(h, w)
        net = O.OracleNet(proto, layers)
        img = (np.random.RandomState(0).randn(1, 3, h, w) * 50).astype(np.float32)
        fl = flops_full * (h * w) / float(full_hw[0] * full_hw[1])
        t0 = time.time()
        net.forward(data=img)
        dt = time.time() - t0
        res[tag] = dict(seconds=dt, gflops=fl / dt / 1e9, hw=(h, w), threads=nt)
    a, o = res["all"], res["one"]
    return {
        "value": a["gflops"] * 1e9 / flops_full,
        "unit": "images/s",
        "cores": a["threads"],
        "kind": "port",
        "sample": "oracle (C restatement of Caffe im2col+SGEMM path, OpenMP SGEMM) on ONE 1x3x%dx%d forward, %.1fs, "
                  "%.1f GFLOP/s on %d threads; value = that rate / %.2f GFLOP per 544x736 image"
                  % (a["hw"][0], a["hw"][1], a["seconds"], a["gflops"], a["threads"], flops_full / 1e9),
        "single_thread_value": o["gflops"] * 1e9 / flops_full,
        "single_thread_sample": "same code, 1 thread, one 1x3x%dx%d forward, %.1fs, %.1f GFLOP/s"
                                % (o["hw"][0], o["hw"][1], o["seconds"], o["gflops"]),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=544)
    ap.add_argument("--width", type=int, default=736)
    ap.add_argument("--batch", type=int, default=1, help="images per rank per step")
    ap.add_argument("--depth", type=int, default=152)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--breakdown", default="", help="write the per-launch hipEvent table to this file")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    import caffe
    from deepcut_tools import deepercut_prototxt, synth_weights, gather_maps_known

    caffe.set_mode_gpu()
    caffe.set_device(local_rank)
    H, W, B = args.height, args.width, args.batch
    layers = synth_weights(args.depth, seed=0)
    net = caffe.Net(deepercut_prototxt(args.depth, H, W, B), caffe.TEST, from_text=True,
                    hipgraph=0 if args.no_graph else 1)
    inject_weights(net, layers)
    net.blobs["data"].reshape(B, 3, H, W)
    net.reshape()
    flops_img = net.flops() / B
    shp = {k: net.blobs[k].shape for k in ("prob", "loc_pred", "next_pred")}
    nel = {k: int(np.prod(s)) for k, s in shp.items()}

    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    x = (torch.randn(B, 3, H, W, generator=g) * 50).to(dev)
    out = torch.empty(sum(nel.values()), dtype=torch.float32, device=dev)
    o_prob = out[: nel["prob"]]
    o_loc = out[nel["prob"]: nel["prob"] + nel["loc_pred"]]
    o_next = out[nel["prob"] + nel["loc_pred"]:]
    recv = None
    if world > 1 and rank == 0:
        recv = [torch.empty_like(out) for _ in range(world)]
    sizes = [out.numel()] * world
    stream = torch.cuda.current_stream(dev)

    def step():
        # asynchronous on torch's current stream; inputs and outputs stay in HBM
        net.forward_device(x.data_ptr(), B, H, W, o_prob.data_ptr(), o_loc.data_ptr(), o_next.data_ptr(),
                           stream.cuda_stream)
        if world > 1:
            gather_maps_known(out, sizes, 0, None, out=recv)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        step()
    e1.record(stream)
    fence()
    dt = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    if rank == 0:
        total_images = args.steps * B * world
        launches = net.num_launches()
        conv_launches = sum(1 for ln in net.plan_text().splitlines() if "conv_gemm<" in ln)
        # roofline of the dominant kernel family (conv_gemm: every convolution/deconvolution launch):
        # algorithmic FLOPs per launch / average launch duration, both over the timed region.  The
        # hipEvents bracket the region on the launch stream, so gaps and the few non-GEMM kernels
        # (max-pool, sigmoid, layout converts) are charged to the GEMM launches: a lower bound.
        per_launch_flops = flops_img * B / conv_launches
        avg_launch_s = (ev_ms / 1e3) / (args.steps * conv_launches)
        achieved = per_launch_flops / avg_launch_s / 1e12
        res = {
            "metric": "images/sec, DeeperCut ResNet-%d FCN forward (prob+loc_pred+next_pred), whole node" % args.depth,
            "value": total_images / dt,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (randn*50 images resident in HBM; conditioned random-init weights, seed 0)",
            "config": {
                "workload": "batch=%d single-scale %dx%d (WxH) ResNet-%d DeeperCut forward per GPU, fp32 "
                            "(BASELINE configs[1]%s)" % (B, W, H, args.depth, "" if (B, H, W) == (1, 544, 736) else ", resized"),
                "per_gpu_batch": B,
                "global_batch": B * world,
                "input": [B, 3, H, W],
                "gflop_per_image": flops_img / 1e9,
                "launches_per_forward": launches,
                "hipgraph": not args.no_graph,
                "parallelism": "dp%d (images sharded, maps gathered to rank 0 by RCCL send/recv)" % world if world > 1 else "single GPU",
            },
            "tflops": total_images * flops_img / dt / 1e12,
            "roofline": {
                "bound": "mfma",
                "kernel": "conv_gemm (fp32 v_mfma_f32_32x32x2_f32 gather-GEMM, all tile variants)",
                "achieved": achieved,
                "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "flops_per_launch": per_launch_flops,
                "avg_launch_us": avg_launch_s * 1e6,
                "launches_per_image": conv_launches,
                "traffic": None,
            },
        }
        if args.breakdown:
            net.blobs["data"].data[...] = x.cpu().numpy()
            net.forward()
            with open(args.breakdown, "w") as f:
                f.write(net.plan_text())
                f.write(net.profile_text(20))
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(lambda h, w: deepercut_prototxt(args.depth, h, w), layers,
                                               flops_img, (H, W))
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
