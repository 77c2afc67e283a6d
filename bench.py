#!/usr/bin/env python
"""bench.py — DeeperCut part-detector forward throughput on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one forward of the hot path (ResNet-152 FCN + deconvolution heads -> prob / loc_pred /
next_pred) over one batch of synthetic input already resident in HBM, on every rank, followed (N>1)
by the RCCL gather of the three score maps to rank 0.  Two timed regions of exactly K steps each are
run: one forward at a time (kernel durations for the roofline), then with `--streams` (default 4 at batch 1)
independent batch-B forwards in flight on separate HIP streams — a batch-1 layer of this net fills
only ~3/4 of the 256 CUs, the next request's kernels fill the rest; `value` is that throughput and the
one-at-a-time figure is reported beside it.  The streams of the forwards in flight are the executors' own, chosen by the library
(`dc_nets_choose_streams`: the real forwards timed on a pool of `--stream-candidates` streams; `config.stream_choice`).  Workload at N=1 = BASELINE.json configs[1]:
batch=1, 1x3x544x736, fp32 (the shipped prototxt is ResNet-152 — SURVEY F1 — not the "ResNet-101"
of the config string).  Weak scaling: every rank forwards its own image each step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# The host driver of these boxes supports dmabuf IPC only: without this RCCL's peer buffers (and CUDA-tensor sharing across processes)
# fail with `hipIpcGetMemHandle: invalid argument`.  Set before torch / the HIP runtime is loaded, in every process of the bench —
# the launcher's children inherit it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "deepcut-cnn_amd")
for p in (ROOT, PKG, os.path.join(PKG, "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (no sparsity)


def inject_weights(net, layers):
    for name, _typ, blobs in layers:
        ps = net.params[name]
        for p, b in zip(ps, blobs):
            p.data[...] = b


def effective_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU
    box exposes 256 hardware threads but grants 16 CPUs of quota; oversubscribing them makes the OpenMP
    GEMM slower than one thread)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(proto_fn, layers, flops_full, full_hw):
    """The reference's CPU algorithm (oracle/: per-image im2col + SGEMM, unfused BatchNorm / Scale / ReLU /
    Eltwise passes, NCHW fp32) timed on this host's cores on a bounded sample of the SAME workload, `caffe time` style
    (tools/caffe.cpp:302-388: one warm-up forward, then timed forwards, median and min):
    whole 544x736 forwards on all cores (>= 10 forwards or 10 s, whichever comes first), then whole 544x736 forwards
    on ONE thread — the reference's default BLAS, ATLAS, is single-threaded: this is the denominator of the
    north-star's ">= 50x" — (>= 3 forwards, up to 10 or 15 s).  Both are measured on the configuration itself; nothing is
    extrapolated from a smaller image."""
    import numpy as np

    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle import oracle as O

    threads = max(1, effective_cores())
    h, w = full_hw
    net = O.OracleNet(proto_fn(h, w), layers)
    img = (np.random.RandomState(0).randn(1, 3, h, w) * 50).astype(np.float32)

    def timed(nthreads, min_n, max_n, budget):
        O.set_threads(nthreads)
        net.forward(data=img)  # warm-up (page faults of the blob dict, OpenMP team start-up)
        ts, t_start = [], time.time()
        while len(ts) < max_n and (len(ts) < min_n or time.time() - t_start < budget):
            t0 = time.time()
            net.forward(data=img)
            ts.append(time.time() - t0)
        return ts

    ts_all = timed(threads, 3, 50, 10.0)
    ts_one = timed(1, 3, 10, 15.0)
    med_all, med_one = sorted(ts_all)[len(ts_all) // 2], sorted(ts_one)[len(ts_one) // 2]
    return {
        "value": 1.0 / med_all,
        "unit": "images/s",
        "cores": threads,
        "kind": "port",
        "sample": "oracle (C restatement of Caffe's im2col+SGEMM CPU path, OpenMP) on %d whole 1x3x%dx%d forward(s) after one warm-up, "
                  "median %.3f s (min %.3f s), %.1f GFLOP/s on %d threads (= the cgroup CPU quota of this box; %d hardware threads visible)"
                  % (len(ts_all), h, w, med_all, min(ts_all), flops_full / med_all / 1e9, threads, os.cpu_count() or 0),
        "single_thread_value": 1.0 / med_one,
        "single_thread_sample": "same code, 1 thread, %d whole 1x3x%dx%d forward(s) after one warm-up: median %.2f s (min %.2f s) = %.1f GFLOP/s "
                                "(measured on the configuration, not scaled from a smaller image)"
                                % (len(ts_one), h, w, med_one, min(ts_one), flops_full / med_one / 1e9),
    }


def hbm_traffic_from_profile(workload=("f32", 1, 544, 736)):
    """HBM bytes per conv_gemm launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    corrected as MI355X_MICROARCH.md prescribes) — counters cannot be read inside the timed run.  Passes exist for
    the headline workload and for float16 at batch 8; any other workload reports null."""
    import glob

    suffix = {("f32", 1, 544, 736): "", ("f16", 8, 544, 736): "_f16_b8"}.get(tuple(workload))
    if suffix is None:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_hbm_traffic%s.json" % suffix)), reverse=True):  # newest round first
        try:
            return json.load(open(path))["hbm_bytes_per_launch"], os.path.basename(path)
        except Exception:
            continue
    return None, None


def mfma_counters_from_profile(workload=("f32", 1, 544, 736)):
    """What the committed counter passes say about the matrix pipes (profiles/rNN_pmc_mfma_util_in_flight_by_stage.txt, made by
    tools/pmc_in_flight.sh: SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE per dispatch window, one executor profiled alone and with the
    executors of a second, unprofiled process in flight beside it) — counters cannot be read inside the timed run.
    -> {"busy_cycles_per_forward": MFMA busy cycles of one forward summed over the 1024 SIMDs (a property of the launches, from the
        ALONE pass), "by_stage_alone": {...}, "by_stage_in_flight": {stage: raw MfmaUtil %}, "source": file} or None."""
    import glob
    import re

    want = {("f32", 1, 544, 736): "float32, batch 1", ("f16", 8, 544, 736): "float16, batch 8"}.get(tuple(workload))
    if want is None:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_mfma_util_in_flight_by_stage.txt")), reverse=True):
        try:
            blocks = re.split(r"^################ ", open(path).read(), flags=re.M)[1:]
            out = {"source": os.path.basename(path)}
            for b in blocks:
                head = b.splitlines()[0]
                if not head.startswith(want):
                    continue
                stages = {}
                for ln in b.splitlines():
                    m = re.match(r"^(conv1|res2|res3|res4|res5|heads)\b.*?\s(\d+)\s+([\d.]+)%\s+([\d.]+)%", ln)
                    if m:
                        stages[m.group(1)] = float(m.group(4))
                m = re.search(r"MFMA busy cycles per forward .*?: ([\d.e+]+)", b)
                if "ALONE" in head:
                    out["by_stage_alone"] = stages
                    if m:
                        out["busy_cycles_per_forward"] = float(m.group(1))
                else:
                    out["by_stage_in_flight"] = stages
            if "busy_cycles_per_forward" in out and "by_stage_in_flight" in out:
                return out
        except Exception:
            continue
    return None


def config2_f16_line(caffe, layers, depth, steps, dev, inject, execs=2, tune=True, regions=5):
    """BASELINE configs[2] beside the headline: batch 8 x the 4-scale pyramid of 736x544 (272x368, 408x552, 544x736,
    680x920), float16 operands with float32 accumulation, device-resident.  A step = one pyramid batch (32 forwards = 8
    images).  Measured two ways in the same run: GROUPED (caffe.NetGroup: the four scales as ONE launch sequence of
    multi-problem gather-GEMMs, in two concurrent lanes of two scales: 318 launches per pyramid batch instead of 632 — `value`) and, as in rounds 1-3, scale by
    scale (four batch-8 forwards per step; reported as `scale_by_scale`).  Each: one step at a time, then `execs` in flight."""
    import torch
    from deepcut_tools import deepercut_prototxt

    shapes = [(272, 368), (408, 552), (544, 736), (680, 920)]
    net = caffe.Net(deepercut_prototxt(depth, 544, 736, 8), caffe.TEST, from_text=True, hipgraph=1, dtype="f16")
    inject(net, layers)
    net.reserve(8, *shapes[-1])
    g = torch.Generator(device="cpu").manual_seed(10)
    xs = {s: (torch.randn(8, 3, s[0], s[1], generator=g) * 50).to(dev) for s in shapes}
    flops = 0.0
    for s in shapes:
        net.blobs["data"].reshape(8, 3, *s)
        flops += net.flops()
    nets = [net] + [net.clone() for _ in range(execs - 1)]  # executors (shared weights and tile choices): `execs` batch-8 forwards in flight
    for n in nets[1:]:
        n.reserve(8, *shapes[-1])
    # every executor on a stream of its own (never the default stream: its handle is 0, which the C entries read as "no stream
    # given: the net's own, synchronous" — rounds 1-3 measured "in flight" with executor 0 blocking the host after each forward)
    streams = [torch.cuda.Stream(dev) for _ in nets]
    outs = [{s: [torch.empty(8, c, s[0] // 8, s[1] // 8, device=dev) for c in (14, 28, 364)] for s in shapes} for _ in nets]
    # the grouped form: one group per step in flight, a member per scale (largest first: nothing grows afterwards)
    groups = []
    for e in range(execs):
        members = [net.clone() for _ in shapes]
        for m, s in zip(members, shapes):
            m.reserve(8, *s)
        groups.append(caffe.NetGroup(members))
    gshapes = [(8, s[0], s[1]) for s in shapes]
    # ... and the `execs` pyramid batches in flight COALESCED into one group (a net may sit in several groups as long as they do
    # not run at the same time): every layer once over 4 x execs tensors — more rows per launch for the matrix-class layers
    # (one lane: the point of this form is ONE launch per layer over all 4 x execs tensors; the groups above run in two lanes)
    big = caffe.NetGroup([m for grp in groups for m in grp.nets], lanes=1) if execs > 1 else None

    def pyramid(inflight, k0=0):
        for i, s in enumerate(shapes):
            e = (k0 + i) % inflight
            o = outs[e][s]
            nets[e].forward_device(xs[s].data_ptr(), 8, s[0], s[1], o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), streams[e].cuda_stream)

    def pyramid_grouped(inflight, k0=0):
        e = k0 % inflight
        o = outs[e]
        groups[e].forward_device([xs[s].data_ptr() for s in shapes], gshapes, [o[s][0].data_ptr() for s in shapes],
                                 [o[s][1].data_ptr() for s in shapes], [o[s][2].data_ptr() for s in shapes], stream=streams[e].cuda_stream)

    def pyramid_coalesced(inflight, k0=0):
        if k0 % execs:
            return  # one call = `execs` steps
        big.forward_device([xs[s].data_ptr() for _ in range(execs) for s in shapes], gshapes * execs,
                           [outs[e][s][0].data_ptr() for e in range(execs) for s in shapes], [outs[e][s][1].data_ptr() for e in range(execs) for s in shapes],
                           [outs[e][s][2].data_ptr() for e in range(execs) for s in shapes], stream=streams[0].cuda_stream)

    def counters():
        ns = [n.stats() for n in nets] + [m.stats() for grp in groups for m in grp.nets]
        gs = [grp.stats() for grp in groups] + ([big.stats()] if big else [])
        return (sum(x["lowerings"] for x in ns) + sum(x["merges"] for x in gs), sum(x["graph_instantiations"] for x in ns + gs))

    def timed(fn, inflight):
        for k in range(4):  # lower, tune and capture every shape on every executor it will meet there
            fn(inflight, k)
        torch.cuda.synchronize(dev)
        before = counters()
        dts = []
        for _ in range(regions):
            for st in streams[1:]:
                st.wait_stream(streams[0])
            t0 = time.perf_counter()
            for k in range(steps):
                fn(inflight, k)  # the scales (the groups) rotate over the executors
            torch.cuda.synchronize(dev)
            dts.append(time.perf_counter() - t0)
        after = counters()
        return dts, after[0] - before[0], after[1] - before[1]

    def med(v):
        return sorted(v)[len(v) // 2]

    def figures(dts):
        dt = med(dts)
        tf = steps * flops / dt / 1e12
        return {"value": steps * 8 / dt, "value_min": steps * 8 / max(dts), "value_max": steps * 8 / min(dts), "unit": "image-pyramids/s",
                "ms_per_pyramid_batch": dt / steps * 1e3, "tflops": tf, "roofline_frac_f16": tf / PEAK_FP16_MFMA_TFLOPS}

    g1, relow_g1, inst_g1 = timed(pyramid_grouped, 1)
    gretiled = None
    if execs > 1 and tune:
        # the group's tiles were chosen for ONE grouped forward at a time; with `execs` in flight on their own streams the descent
        # of deepcut_tools.tune_in_flight runs over the groups (same report / override interface as a net), untimed
        from deepcut_tools import tune_in_flight

        def gload():
            t0 = time.perf_counter()
            for r in range(3 * execs):
                pyramid_grouped(execs, r)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        try:
            gretiled = len(tune_in_flight(groups, gload, top=4)["changed"])
        except Exception as e:  # noqa: BLE001
            gretiled = "failed: %s" % e
    g2, relow_g2, inst_g2 = timed(pyramid_grouped, execs)
    gc, relow_gc, inst_gc = timed(pyramid_coalesced, execs) if big and steps % execs == 0 else (None, 0, 0)
    dts1, relow1, inst1 = timed(pyramid, 1)
    # between the regions: the tiles of every scale re-tuned for `execs` forwards in flight (untimed; deepcut_tools.tune_in_flight)
    retiled = None
    if execs > 1 and tune:
        from deepcut_tools import tune_in_flight

        retiled = 0
        for s in shapes:
            def load(s=s):
                t0 = time.perf_counter()
                for r in range(3 * execs):
                    e = r % execs
                    o = outs[e][s]
                    nets[e].forward_device(xs[s].data_ptr(), 8, s[0], s[1], o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), streams[e].cuda_stream)
                torch.cuda.synchronize(dev)
                return time.perf_counter() - t0

            load()  # every executor at this scale: the report and the overrides address its current plan
            retiled += len(tune_in_flight(nets[:execs], load, top=4)["changed"])
    dts2, relow2, inst2 = timed(pyramid, execs)
    f1, f2, fc = figures(g1), figures(g2), (figures(gc) if gc else None)
    # `value` = the best way this build runs the workload: the groups' own concurrent lanes already fill the chip, so a second pyramid
    # batch in flight (or coalesced into the same launches) may or may not add anything — all three are in the line
    forms = [(f2, "%d pyramid batches in flight on %d groups / streams" % (execs, execs)), (f1, "one pyramid batch at a time (its two lanes run concurrently)")]
    if fc is not None:
        forms.append((fc, "%d pyramid batches coalesced into one group of %d executors" % (execs, 4 * execs)))
    best, how = max(forms, key=lambda t: t[0]["value"])
    res = dict(best)
    res.update({"workload": "batch=8 x 4-scale pyramid (272x368, 408x552, 544x736, 680x920) of 736x544 images, fp16 MFMA with fp32 "
                            "accumulate (BASELINE configs[2]); the four scales as ONE grouped forward (caffe.NetGroup: %s); value = %s"
                            % (groups[0].plan_text().splitlines()[0][2:], how),
                "regions": regions, "steps": steps, "forwards_in_flight": execs, "forwards_per_s": res["value"] * 4,
                "gflop_per_image_pyramid": flops / 8 / 1e9, "tile_tuning": "latency (group signatures timed alone, then inside the group's own sequence)" + (
                    "" if gretiled is None else "; in flight: %s signatures re-tiled under %d groups in flight" % (gretiled, execs)),
                "one_forward_at_a_time": f1, "in_flight_on_streams": f2, "in_flight_coalesced": fc})
    sbs = figures(dts2)
    sbs.update({"note": "rounds 1-3 form: four batch-8 forwards per pyramid batch, the scales rotating over %d executors" % execs,
                "tile_tuning": "latency" if retiled is None else "in flight (%d signatures re-tiled over the four scales)" % retiled,
                "one_forward_at_a_time": figures(dts1)})
    res["scale_by_scale"] = sbs
    res["relowerings_in_timed_region"] = relow1 + relow2 + relow_g1 + relow_g2 + relow_gc
    res["graph_instantiations_in_timed_region"] = inst1 + inst2 + inst_g1 + inst_g2 + inst_gc
    return res


def rank_identity(backend, dev):
    """What THIS rank runs on — gathered from every rank into config.distributed so that the N>1 line verifies itself: under
    `nccl` (= RCCL) every rank must sit on a GPU of its own (uuid / PCI bus id), and the line shows what RCCL saw."""
    import socket

    import torch

    ident = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "host": socket.gethostname(),
             "pid": os.getpid(), "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", ""))}
    if dev is not None and dev.type == "cuda":
        p = torch.cuda.get_device_properties(dev)
        ident.update({"device_index": dev.index, "device_name": p.name, "uuid": str(getattr(p, "uuid", "")),
                      "pci": "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0)),
                      "hbm_gb": round(p.total_memory / 2 ** 30, 1)})
    else:
        ident.update({"device_index": None, "device_name": "none (dry run: no forwards)", "uuid": "", "pci": ""})
    return ident


def one_gpu_per_rank(ids, backend):
    """Under nccl (= RCCL) two ranks on one GPU would make every scaling number meaningless: refuse."""
    if backend != "nccl" or len(ids) < 2:
        return
    seen = {}
    for i in ids:
        key = (i["host"], i["uuid"] or i["pci"])
        if key in seen:
            raise SystemExit("bench.py: ranks %d and %d both run on GPU %s of %s under nccl: one process per GPU is the contract"
                             % (seen[key], i["rank"], key[1], key[0]))
        seen[key] = i["rank"]


def distributed_report(backend, dev, world, payload_bytes, gather_gbps):
    """All-gather the rank identities, refuse two ranks on one GPU under nccl, and describe the exchange."""
    import torch
    import torch.distributed as dist

    me = rank_identity(backend, dev)
    ids = [None] * world
    if world > 1:
        dist.all_gather_object(ids, me)
    else:
        ids = [me]
    one_gpu_per_rank(ids, backend)
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:  # noqa: BLE001
        rccl = None
    return {"backend": backend, "ranks_seen": dist.get_world_size() if world > 1 else 1, "rccl_version": rccl,
            "ranks": ids, "distinct_devices": len(set((i["host"], i["uuid"] or i["pci"] or i["rank"]) for i in ids)),
            "gather": {"pattern": "grouped isend/irecv of prob|loc_pred|next_pred to rank 0 (no ring, no reduction): every peer sends over its own "
                                  "xGMI link into the root",
                       "payload_bytes_per_rank_per_step": payload_bytes, "bytes_into_rank0_per_step": payload_bytes * (world - 1),
                       "expected_per_link_bytes_per_step": payload_bytes,
                       "measured_gather_gbps_into_rank0": gather_gbps}}


def dry_run(args):
    """`--dry-run`: the N>1 control path WITHOUT forwards, for a box with no GPU (the CPU test suite runs it at world size 8 over
    gloo): process group, rank identities, the gather of payloads of exactly the size the real run sends (shapes from the
    prototxt, host only), barriers and the max-over-ranks timing, one JSON line marked `dry_run`.  `value` is null: nothing
    was forwarded."""
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    import caffe
    from deepcut_tools import deepercut_prototxt, gather_maps_known

    H, W, B = args.height, args.width, args.batch
    net = caffe.Net(deepercut_prototxt(args.depth, H, W, B), caffe.TEST, from_text=True, dtype=args.dtype)  # host only: shapes
    nel = sum(int(np.prod(net.blobs[k].shape)) for k in ("prob", "loc_pred", "next_pred"))
    dt_t = torch.float16 if args.dtype == "f16" else torch.float32
    out = torch.full((nel,), float(rank), dtype=dt_t)
    sizes = [nel] * world
    recvs = [torch.empty_like(out) for _ in range(world)] if rank == 0 and world > 1 else None

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        gather_maps_known(out, sizes, 0, None, out=recvs)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gather_maps_known(out, sizes, 0, None, out=recvs)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0 and world > 1:
        for r in range(1, world):
            assert float(recvs[r][0]) == float(r) and float(recvs[r][-1]) == float(r), "rank %d's payload did not arrive" % r
    payload = nel * out.element_size()
    rep = distributed_report("gloo", None, world, payload, payload * (world - 1) * args.steps / float(tt.item()) / 1e9 if world > 1 else None)
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN: control path of `bench.py --gpus %d` without forwards (no GPU needed)" % world, "dry_run": True,
                          "value": None, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": float(tt.item()) / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": args.dtype, "data": "synthetic payloads of the real run's size",
                          "config": {"workload": "gather of batch=%d x %dx%d maps per rank (BASELINE configs[%d]), %s payload" % (B, W, H, args.config, args.dtype),
                                     "per_gpu_batch": B, "global_batch": B * world, "distributed": rep}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): start the run the way
    the contract describes it — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>`, one rank per GPU — and hand its exit code back.  Rank 0's JSON line goes to
    this process's stdout (the children inherit it)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.setdefault("OMP_NUM_THREADS", str(max(1, effective_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=5,
                    help="how many times each timed region (W warm-up + exactly K timed steps) is run; value = the median region, "
                         "value_min / value_max the spread (one 45-ms region cannot resolve a 1 %% change)")
    ap.add_argument("--height", type=int, default=544)
    ap.add_argument("--width", type=int, default=736)
    ap.add_argument("--batch", type=int, default=0, help="images per rank per step (default: 1, or 8 with --config 3)")
    ap.add_argument("--config", type=int, default=1, choices=[1, 3],
                    help="BASELINE.json configs index: 1 = batch 1 per GPU (the headline), 3 = 8 images of 736x544 per GPU "
                         "(batch 64 sharded 8-way) with the gather of the maps to rank 0")
    ap.add_argument("--coalesce", type=int, default=4,
                    help="also measure cross-request batching (deepcut_tools.Pipeline, opportunistic): at most k batch-1 requests per batch forward (0/1: skip)")
    ap.add_argument("--no-f16-line", action="store_true", help="skip the configs[2] (fp16 pyramid) measurement printed beside the headline")
    ap.add_argument("--depth", type=int, default=152)
    ap.add_argument("--no-resnet101", action="store_true", help="skip the ResNet-101 side figure (the depth BASELINE.json names)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="device element type: f32 (the headline, BASELINE configs[1]) or f16 operands with fp32 "
                         "accumulation (BASELINE configs[2])")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-tune-in-flight", action="store_true",
                    help="keep the latency-tuned tiles for the in-flight region too (default: deepcut_tools.tune_in_flight between the regions)")
    ap.add_argument("--breakdown", default="", help="write the per-launch hipEvent table to this file")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DC_BENCH_STREAMS", "0")),
                    help="independent batch-B forwards kept in flight per GPU (each on its own HIP stream and Net)")
    ap.add_argument("--stream-candidates", type=int, default=int(os.environ.get("DC_BENCH_STREAM_CANDIDATES", "8")),
                    help="the streams of the forwards in flight are chosen by the library (dc_nets_choose_streams) among this many "
                         "process-wide candidates, by timing the forwards (0: streams created here, unmeasured)")
    ap.add_argument("--backend", default=os.environ.get("DC_BENCH_BACKEND", "nccl"),
                    help="torch.distributed backend for N>1: nccl (= RCCL, the real thing) or gloo (lets two ranks share one "
                         "GPU to smoke-test the N>1 code path on a 1-GPU box)")
    ap.add_argument("--dry-run", action="store_true",
                    help="N>1 control path only (process group, rank identities, gather of real-size payloads over gloo), no forwards, no GPU")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = 8 if args.config == 3 else 1
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # started like the N=1 line: become the launcher
    if args.dry_run:
        return dry_run(args)
    if args.streams <= 0:  # forwards kept in flight: 4 at batch 1 (one per hardware queue of a HIP process), 2 at batch 8
        args.streams = 4 if args.batch < 4 else 2

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=1" % args.gpus)
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())  # smoke test: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

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
    S = max(1, args.streams)
    proto = deepercut_prototxt(args.depth, H, W, B)
    net = caffe.Net(proto, caffe.TEST, from_text=True, hipgraph=0 if args.no_graph else 1, dtype=args.dtype)
    inject_weights(net, layers)
    net.blobs["data"].reshape(B, 3, H, W)
    net.reshape()
    # one executor per in-flight forward: clones own their activations / stream / graph and share the
    # parameters and the packed weights in HBM with `net`
    nets = [net] + [net.clone() for _ in range(S - 1)]
    flops_img = net.flops() / B
    shp = {k: net.blobs[k].shape for k in ("prob", "loc_pred", "next_pred")}
    nel = {k: int(np.prod(s)) for k, s in shp.items()}

    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    # every executor on a stream of its own — never the default stream: its handle is 0, which dc_net_forward_batch reads as
    # "no stream given: the net's own stream, synchronous" (rounds 1-3 ran executor 0 that way: the host blocked on it after
    # every one of its forwards while the other executors ran ahead)
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    xs = [(torch.randn(B, 3, H, W, generator=g) * 50).to(dev) for _ in range(S)]
    # the maps leave the net in its own element type: float16 payloads from a float16 net (half the gather bytes)
    half = args.dtype == "f16"
    outs = [torch.empty(sum(nel.values()), dtype=torch.float16 if half else torch.float32, device=dev) for _ in range(S)]
    a, b = nel["prob"], nel["prob"] + nel["loc_pred"]
    recvs = [None] * S
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")  # gloo moves host buffers
    if world > 1 and rank == 0:  # one set of receive buffers per in-flight forward
        recvs = [[torch.empty_like(outs[0], device=comm_dev) for _ in range(world)] for _ in range(S)]
    sizes = [outs[0].numel()] * world
    # N > 1: the gather runs on ONE communication stream, ordered after each forward by an event — RCCL's send/recv kernels
    # are then in no executor stream's dependency chain (with the gather waited for on the compute stream, every one of the
    # S streams would carry them and rank 0's 7 receives per step would hold up the next forward of that slot)
    comm = torch.cuda.Stream(dev) if world > 1 else None
    sent = [None] * S  # per in-flight slot: event on `comm` after which outs[k] / recvs[k] may be reused

    def forward_slot(k):
        st, out, x = streams[k], outs[k], xs[k]
        if half:
            nets[k].forward_device(x.data_ptr(), B, H, W, None, None, None, st.cuda_stream)
            nets[k].emit_maps_device(out[:a].data_ptr(), out[a:b].data_ptr(), out[b:].data_ptr(), half=True, stream=st.cuda_stream)
        else:
            nets[k].forward_device(x.data_ptr(), B, H, W, out[:a].data_ptr(), out[a:b].data_ptr(), out[b:].data_ptr(),
                                   st.cuda_stream)

    def step(i, nstreams):
        # asynchronous on stream i % nstreams; inputs and outputs stay in HBM
        k = i % nstreams
        st, out, x = streams[k], outs[k], xs[k]
        if world > 1 and sent[k] is not None:
            st.wait_event(sent[k])  # the previous payload of this slot has left
        forward_slot(k)
        if world > 1:
            if args.backend == "nccl":
                done = torch.cuda.Event()
                done.record(st)
                with torch.cuda.stream(comm):
                    comm.wait_event(done)
                    _bufs, reqs = gather_maps_known(out, sizes, 0, None, out=recvs[k], async_op=True)
                    for q in reqs:
                        q.wait()  # orders `comm` after the transfer; the host does not block
                    sent[k] = torch.cuda.Event()
                    sent[k].record(comm)
            else:  # smoke-test transport (gloo): through the host, synchronously
                st.synchronize()
                gather_maps_known(out.cpu(), sizes, 0, None, out=recvs[k])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_region(nstreams, steps, warmup):
        """W untimed + exactly `steps` timed steps, barrier + synchronize on both sides; returns
        (max-over-ranks wall seconds, hipEvent ms on this rank's launch streams)."""
        for i in range(warmup * nstreams):
            step(i, nstreams)
        fence()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        main = streams[0]
        e0.record(main)
        for st in streams[1:nstreams]:
            st.wait_stream(main)
        for i in range(steps):
            step(i, nstreams)
        for st in streams[1:nstreams]:
            main.wait_stream(st)
        e1.record(main)
        fence()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()), e0.elapsed_time(e1)

    def repeated(nstreams):
        """`--regions` timed regions of exactly K steps each (W warm-up steps in front of every one): the MEDIAN region's
        (wall seconds, hipEvent ms) and the list of all wall times.  Every rank runs the same number of regions."""
        rs = [timed_region(nstreams, args.steps, args.warmup) for _ in range(max(1, args.regions))]
        order = sorted(range(len(rs)), key=lambda i: rs[i][0])
        m = rs[order[len(rs) // 2]]
        return m[0], m[1], [r[0] for r in rs]

    # (1) one forward at a time: per-kernel durations are undisturbed -> the roofline figure
    lat_dt, lat_ev_ms, lat_all = repeated(1)
    # (2) the reported throughput: S independent batch-B forwards in flight (S streams, S Nets).  The tiles region (1) ran with were
    # chosen for the latency of one forward; a service that keeps S forwards in flight tunes for THAT load (untimed, like the
    # autotuning inside the warm-up): deepcut_tools.tune_in_flight, coordinate descent over the busiest GEMM signatures.
    # WHICH S streams?  A HIP process has four hardware queues and the runtime binds every stream to one of them when it is
    # created; S forwards "in flight" on streams that share queues are fewer forwards in flight (tools/stream_subset_probe.py, four
    # executors on the 70 four-subsets of eight streams: 12 subsets at 478-492 images/s, 53 at 419-434, 5 at 376-390 — and which
    # subset the first S streams of a process are depends on every stream created before them).  So, like the lanes of a group
    # (NetGroup::choose_lane_streams): candidates, the real forwards on every S-subset, the fastest subset stays.  Untimed.
    stream_choice = None
    if S > 1 and args.stream_candidates > 0:
        # the LIBRARY's choice (dc_nets_choose_streams, csrc/streams.cpp): the executors' real forwards timed on assignments of a
        # process-wide pool of candidate streams (greedy, ~20 bursts); every executor adopts its stream as its own, and the bench
        # drives exactly those streams.  Round 4 searched the 70 subsets here, in the bench: a caller of the library got luck.
        for e in nets:
            e.reserve(B, H, W)
        stream_choice = caffe.choose_streams(nets, candidates=args.stream_candidates)
        stream_choice = {"by": "dc_nets_choose_streams", "candidates": args.stream_candidates,
                         "images_per_s_chosen": stream_choice["forwards_per_s_chosen"] * B,
                         "images_per_s_first_created": stream_choice["forwards_per_s_first_created"] * B}
        for k, e in enumerate(nets):
            streams[k] = torch.cuda.ExternalStream(e.stream_handle(), device=dev)
    tuning = None
    if S > 1 and not args.no_tune_in_flight:
        from deepcut_tools import tune_in_flight

        def load():
            t0 = time.perf_counter()
            for i in range(6 * S):
                forward_slot(i % S)  # the forwards of a step without its gather (rank-local: no collective inside the tuner)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        try:
            tuning = tune_in_flight(nets, load)
        except Exception as e:  # noqa: BLE001  (the line is printed whatever happens here)
            tuning = {"error": "%s: %s" % (type(e).__name__, e)}
    if S > 1:
        dt, ev_ms, dt_all = repeated(S)
    else:
        dt, ev_ms, dt_all = lat_dt, lat_ev_ms, lat_all
    x = xs[0]

    # N > 1: the exchange alone (no forwards): what rank 0's seven receives reach, and who the ranks are
    gather_gbps = None
    payload_bytes = outs[0].numel() * outs[0].element_size()
    if world > 1:
        fence()
        g0 = time.perf_counter()
        for _ in range(10):
            if args.backend == "nccl":
                _b, reqs = gather_maps_known(outs[0], sizes, 0, None, out=recvs[0], async_op=True)
                for q in reqs:
                    q.wait()
            else:
                gather_maps_known(outs[0].cpu(), sizes, 0, None, out=recvs[0])
        fence()
        gt = torch.tensor([time.perf_counter() - g0], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gather_gbps = payload_bytes * (world - 1) * 10 / float(gt.item()) / 1e9
    dist_report = distributed_report(args.backend if world > 1 else "none", dev, world, payload_bytes, gather_gbps)

    if rank == 0:
        total_images = args.steps * B * world
        launches = net.num_launches()
        conv_launches = sum(1 for ln in net.plan_text().splitlines() if "conv_gemm<" in ln or "wino_f23<" in ln or "wino_h23<" in ln or "ws1x1<" in ln or "ws1x1f<" in ln or "ws7x7f<" in ln or "stem7x7<" in ln)
        # roofline of the dominant kernel family (conv_gemm: every convolution/deconvolution launch):
        # algorithmic FLOPs per launch / average launch duration over the ONE-FORWARD-AT-A-TIME timed
        # region (launches do not overlap there, so the duration is the kernel's own and agrees with
        # rocprofv3 --stats).  The hipEvents bracket the region on the launch stream, so gaps and the few
        # non-GEMM kernels (max-pool, layout converts) are charged to the GEMM launches: a lower bound.
        per_launch_flops = flops_img * B / conv_launches
        avg_launch_s = (lat_ev_ms / 1e3) / (args.steps * conv_launches)
        achieved = per_launch_flops / avg_launch_s / 1e12
        res = {
            "metric": "images/sec, DeeperCut ResNet-%d FCN forward (prob+loc_pred+next_pred), whole node" % args.depth,
            "value": total_images / dt,
            "value_min": total_images / max(dt_all),
            "value_max": total_images / min(dt_all),
            "regions": len(dt_all),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (randn*50 images resident in HBM; conditioned random-init weights, seed 0)",
            "config": {
                "workload": "batch=%d single-scale %dx%d (WxH) ResNet-%d DeeperCut forward per GPU, %s "
                            "(BASELINE configs[%d]%s)" % (B, W, H, args.depth, "fp16 MFMA / fp32 accumulate" if args.dtype == "f16" else "fp32",
                                                         args.config, "" if (H, W) == (544, 736) and B == (8 if args.config == 3 else 1) else ", resized"),
                "per_gpu_batch": B,
                "global_batch": B * world,
                "input": [B, 3, H, W],
                "gflop_per_image": flops_img / 1e9,
                "launches_per_forward": launches,
                "hipgraph": not args.no_graph,
                "forwards_in_flight": S,
                "stream_choice": stream_choice,
                "tile_tuning": ("in flight (deepcut_tools.tune_in_flight: %d signatures re-tiled, %d without isolated timings skipped, %.2f -> %.2f ms per %d forwards, %d untimed runs)"
                                % (len(tuning["changed"]), tuning.get("skipped", 0), tuning["before"] * 1e3, tuning["after"] * 1e3, 6 * S, tuning["runs"])
                                if tuning and "error" not in tuning else ("latency" if not tuning else tuning["error"])),
                "parallelism": "dp%d (images sharded, maps gathered to rank 0 by RCCL send/recv)" % world if world > 1 else "single GPU",
                "distributed": dist_report,
            },
            "tflops": total_images * flops_img / dt / 1e12,
            "one_forward_at_a_time": {"value": total_images / lat_dt, "value_min": total_images / max(lat_all),
                                      "value_max": total_images / min(lat_all), "unit": "images/s",
                                      "ms_per_step": lat_dt / args.steps * 1e3,
                                      "tflops": total_images * flops_img / lat_dt / 1e12},
            "roofline": {
                "bound": "mfma",
                "kernel": "conv_gemm + wino_f23 / wino_h23 + ws1x1f / ws1x1 / stem7x7 (%s gather-GEMM, all tile variants; Winograd F(2x2,3x3) and the streaming 1x1 forms where they are faster)" % (
                    "f16 v_mfma_f32_32x32x16_f16, fp32 accumulate" if args.dtype == "f16" else "fp32 v_mfma_f32_32x32x2_f32"),
                "achieved": achieved,
                "peak": PEAK_FP16_MFMA_TFLOPS if args.dtype == "f16" else PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / (PEAK_FP16_MFMA_TFLOPS if args.dtype == "f16" else PEAK_FP32_MFMA_TFLOPS),
                "flops_per_launch": per_launch_flops,
                "avg_launch_us": avg_launch_s * 1e6,
                "launches_per_image": conv_launches,
                "traffic": hbm_traffic_from_profile((args.dtype, B, H, W))[0],
                "traffic_source": "committed profile profiles/%s (rocprofv3 PMC passes of this workload; counters cannot be read inside the timed run)"
                                  % hbm_traffic_from_profile((args.dtype, B, H, W))[1],
                "traffic_unit": "HBM bytes per conv_gemm launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/%s); "
                                "algorithmic minimum %.1f MB" % (hbm_traffic_from_profile((args.dtype, B, H, W))[1],
                                                                 (2.07e9 * (H * W) / (544.0 * 736.0) * B + 0.263e9) * (0.5 if args.dtype == "f16" else 1.0) / conv_launches / 1e6),
                "achieved_with_forwards_in_flight": total_images * flops_img / dt / 1e12,
                # `value` is the in-flight regime, `frac` above the one-forward-at-a-time one (launches do not overlap there, so a launch
                # duration exists): the same ratio for the regime of `value`, and what the counters say about it
                "frac_in_flight": total_images * flops_img / dt / 1e12 / (PEAK_FP16_MFMA_TFLOPS if args.dtype == "f16" else PEAK_FP32_MFMA_TFLOPS),
            },
        }
        mc = mfma_counters_from_profile((args.dtype, B, H, W))
        if mc:
            # share of the wall time the 1024 matrix pipes are busy = busy cycles one forward issues (counted) x forwards per second
            # (this run) / (2.4e9 x 1024): at the 2.4 GHz the loaded chip does not always hold — an upper bound on the clock, a lower
            # bound on the share.  by_stage: raw SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of the committed counter windows
            res["roofline"].update({
                "mfma_busy_cycles_per_forward": mc["busy_cycles_per_forward"],
                "mfma_busy_frac_in_flight": mc["busy_cycles_per_forward"] * (total_images / dt / B) / (2.4e9 * 1024),
                "mfma_busy_frac_one_at_a_time": mc["busy_cycles_per_forward"] * (total_images / lat_dt / B) / (2.4e9 * 1024),
                "mfma_util_by_stage_in_flight_pct": mc["by_stage_in_flight"],
                "mfma_util_by_stage_alone_pct": mc.get("by_stage_alone"),
                "mfma_counters_source": "committed profile profiles/%s (tools/pmc_in_flight.sh)" % mc["source"],
            })
        # --- measurements reported BESIDE the headline (never as `value`).  Each one is guarded: whatever happens in them,
        #     the line with `value`, `roofline` (and, if it ran, `cpu_baseline`) is printed.
        def beside(key, fn):
            try:
                res[key] = fn()
            except Exception as e:  # noqa: BLE001
                res[key] = {"error": "%s: %s" % (type(e).__name__, e)}
                print("bench.py: %s failed: %s" % (key, e), file=sys.stderr)

        def pcie_inclusive():
            # the boundary as pycaffe uses it: host NCHW buffers in, host maps out (4.8 MB up + 10.2 MB down per
            # 544x736 image over PCIe, pageable numpy memory)
            xh = x.cpu().numpy()
            net.forward_batch(xh)
            t1 = time.perf_counter()
            for _ in range(n_pcie):
                net.forward_batch(xh)
            dt_pcie = time.perf_counter() - t1
            return {"value": n_pcie * B / dt_pcie, "unit": "images/s", "ms_per_forward": dt_pcie / n_pcie * 1e3,
                    "host_path_overhead_ms": dt_pcie / n_pcie * 1e3 - lat_dt / args.steps * 1e3,
                    "note": "dc_net_forward_batch with host buffers (pageable numpy memory), synchronous, one forward at a time; "
                            "host_path_overhead_ms = this minus the device-resident forward of the same net (%s batch %d)" % (args.dtype, B)}

        def pycaffe_forward():
            # the reference's own call sequence (python/pose/estimate_pose.py:104-112): write blobs['data'].data, net.forward(),
            # read the three output blobs — host blobs are pinned memory owned by the library
            xh = x.cpu().numpy()
            net.blobs["data"].data[...] = xh
            net.forward()
            t1 = time.perf_counter()
            for _ in range(n_pcie):
                net.blobs["data"].data[...] = xh
                net.forward()
                outs_h = [net.blobs[k].data for k in ("prob", "loc_pred", "next_pred")]
            dt_py = time.perf_counter() - t1
            assert all(o.size for o in outs_h)
            return {"value": n_pcie * B / dt_py, "unit": "images/s", "ms_per_forward": dt_py / n_pcie * 1e3,
                    "note": "pycaffe drop-in path: blobs['data'].data[...] = image; net.forward(); read prob / loc_pred / next_pred"}

        def image_entry():
            # uint8 pixels up (1.2 MB), pre-processing + forward + pose decode on the device, 5x14 doubles down — what the
            # demo needs per image.  As pose.estimate_pose runs it since round 6: the net computes `prob` and `loc_pred` only
            # (DC_OPT_OUTPUTS; the reference's demo never reads `next_pred`, python/pose/estimate_pose.py:231-241: 23.3 of the 241 GFLOP)
            img8 = np.random.RandomState(2).randint(0, 256, (B, H, W, 3)).astype(np.uint8)

            def rate(e):
                e.forward_images(img8, 1.0, want=(), pose=True)
                t1 = time.perf_counter()
                for _ in range(n_pcie):
                    e.forward_images(img8, 1.0, want=(), pose=True)
                return n_pcie * B / (time.perf_counter() - t1)

            demo = net.clone()
            demo.set_outputs(["prob", "loc_pred"])
            v_two, v_all = rate(demo), rate(net)
            return {"value": v_two, "unit": "images/s", "ms_per_forward": B / v_two * 1e3, "gflop_per_image": demo.flops() / B / 1e9,
                    "all_three_outputs_computed": {"value": v_all, "ms_per_forward": B / v_all * 1e3, "gflop_per_image": flops_img / 1e9},
                    "note": "dc_net_forward_images: uint8 HWC in, pose out, synchronous; outputs prob + loc_pred only (DC_OPT_OUTPUTS), beside it the "
                            "same call on a net that computes next_pred as well"}

        def cross_request_batching():
            # deepcut_tools.Pipeline, its default policy: independent batch-1 requests; one goes out alone while an executor is free,
            # what queues up behind busy executors leaves as ONE batch forward of up to --coalesce requests (dc_net_forward_requests).
            # Closed loop: `window` requests outstanding (queued + in flight), a new one submitted for every one that finishes —
            # the service figure, with the latency a request sees under that load; `value` stays strict batch-1 forwards in flight.
            from deepcut_tools import Pipeline

            pipe = Pipeline(net, depth=1, max_batch=args.coalesce)
            pipe.nets = nets[:3]  # reuse the executors (and their tuned plans); three of them: a fourth adds latency, not throughput
            pipe.max_queue = pipe.max_batch * len(pipe.nets)
            window = pipe.max_queue + len(pipe.nets) * pipe.max_batch
            nreq = max(args.steps, 10) * 8
            bufs = [(xs[i % S], [torch.empty(B, c, H // 8, W // 8, device=dev) for c in (shp["prob"][1], shp["loc_pred"][1], shp["next_pred"][1])])
                    for i in range(2 * window)]

            def closed_loop(count):
                sent = done = 0
                while done < count:
                    while sent < count and sent - done < window:
                        xi, o = bufs[sent % len(bufs)]
                        pipe.submit(xi.data_ptr(), 1, H, W, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), tag=sent)
                        sent += 1
                    pipe.wait_one()
                    done += 1

            closed_loop(6 * window)  # every batch size it will form has been lowered, tuned and captured on every executor
            torch.cuda.synchronize(dev)
            pipe.reset_stats()
            dts = []
            for _ in range(max(1, args.regions)):
                t1 = time.perf_counter()
                closed_loop(nreq)
                torch.cuda.synchronize(dev)
                dts.append(time.perf_counter() - t1)
            dtc = sorted(dts)[len(dts) // 2]
            pct = pipe.latency_percentiles((50, 90, 99))
            return {"value": nreq / dtc, "value_min": nreq / max(dts), "value_max": nreq / min(dts), "unit": "images/s",
                    "policy": "opportunistic (whatever is queued when an executor frees, up to %d requests per batch forward)" % pipe.max_batch,
                    "executors": len(pipe.nets), "requests_outstanding": window, "requests_per_region": nreq, "regions": len(dts),
                    "latency_ms": {"p50": pct.get(50), "p90": pct.get(90), "p99": pct.get(99)},
                    "batch_sizes": {str(k): v for k, v in sorted(pipe.batch_sizes.items())},
                    "note": "independent batch-1 requests through deepcut_tools.Pipeline (dc_net_forward_requests); latency = submit -> seen finished, closed loop"}

        def pcie_inclusive_pipelined():
            # host in / host out with requests IN FLIGHT: deepcut_tools.Pipeline.submit_host on the same executors (their streams
            # chosen by the library), pinned buffers from caffe.pinned_empty: a request's upload, forward and the downloads of all
            # three maps sit on its executor's own stream, so the DMA engines work beside the other executors' kernels.
            # Closed loop, 2 x depth requests outstanding; SURVEY 8(d): "the whole forward() including H2D of the input and D2H
            # of the three maps".
            from deepcut_tools import Pipeline

            pipe = Pipeline(net, depth=1, coalesce=1, choose_streams=False)
            pipe.nets = nets[:S]
            nslot = 2 * len(pipe.nets)
            slots = []
            for i in range(nslot):
                xi = caffe.pinned_empty((B, 3, H, W))
                xi[...] = xs[i % S].cpu().numpy()
                slots.append((xi, [caffe.pinned_empty(tuple(shp[k])) for k in ("prob", "loc_pred", "next_pred")]))
            nreq = max(args.steps, 10) * 4

            def closed_loop(count):
                sent = done = 0
                while done < count:
                    while sent < count and sent - done < nslot:
                        xi, o = slots[sent % nslot]
                        pipe.submit_host(xi, o[0], o[1], o[2], tag=sent)
                        sent += 1
                    pipe.wait_one()
                    done += 1

            closed_loop(2 * nslot)
            dts = []
            for _ in range(max(1, args.regions)):
                t1 = time.perf_counter()
                closed_loop(nreq)
                dts.append(time.perf_counter() - t1)
            dtp = sorted(dts)[len(dts) // 2]
            ref = None
            if args.dtype == "f32":  # the maps of the last request of slot 0 against the device-resident forward of the same input
                o = [torch.empty(tuple(shp[k]), device=dev) for k in ("prob", "loc_pred", "next_pred")]
                net.forward_device(xs[0].data_ptr(), B, H, W, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
                ref = max(float((o[j].cpu() - torch.from_numpy(slots[0][1][j])).abs().max()) for j in range(3))
            return {"value": nreq * B / dtp, "value_min": nreq * B / max(dts), "value_max": nreq * B / min(dts), "unit": "images/s",
                    "executors": len(pipe.nets), "requests_outstanding": nslot, "requests_per_region": nreq, "regions": len(dts),
                    "bytes_per_image": {"host_to_device": 3 * H * W * 4, "device_to_host": int(sum(nel.values()) // B) * 4},
                    "max_abs_diff_vs_device_resident_forward": ref,
                    "note": "deepcut_tools.Pipeline.submit_host (dc_net_forward_host_async): pinned host buffers in and out, all three maps, "
                            "closed loop; what `pcie_inclusive` (one synchronous request at a time, pageable memory) becomes with requests in flight"}

        def resnet101():
            # the depth BASELINE.json's metric names (the reference ships only ResNet-152, SURVEY F1: that is the headline); same
            # workload, same protocol, conditioned synthetic weights of the 101 table
            l101 = synth_weights(101, seed=0)
            n101 = caffe.Net(deepercut_prototxt(101, H, W, B), caffe.TEST, from_text=True, hipgraph=0 if args.no_graph else 1, dtype=args.dtype)
            inject_weights(n101, l101)
            e101 = [n101] + [n101.clone() for _ in range(S - 1)]
            for e in e101:
                e.reserve(B, H, W)
            choice = caffe.choose_streams(e101, candidates=max(args.stream_candidates, S)) if S > 1 else None
            st101 = [torch.cuda.ExternalStream(e.stream_handle(), device=dev) for e in e101]
            o101 = [torch.empty(sum(nel.values()), device=dev) for _ in e101]
            fl = n101.flops() / B

            def run(ns, count):
                for i in range(count):
                    k = i % ns
                    e101[k].forward_device(xs[k % S].data_ptr(), B, H, W, o101[k][:a].data_ptr(), o101[k][a:b].data_ptr(), o101[k][b:].data_ptr(),
                                           st101[k].cuda_stream)
                torch.cuda.synchronize(dev)

            out = {}
            for key, ns in (("one_forward_at_a_time", 1), ("in_flight", S)):
                run(ns, 2 * ns + args.warmup)
                dts = []
                for _ in range(max(1, args.regions)):
                    t1 = time.perf_counter()
                    run(ns, args.steps)
                    dts.append(time.perf_counter() - t1)
                dtm = sorted(dts)[len(dts) // 2]
                out[key] = {"value": args.steps * B / dtm, "value_min": args.steps * B / max(dts), "value_max": args.steps * B / min(dts),
                            "unit": "images/s", "ms_per_step": dtm / args.steps * 1e3, "tflops": args.steps * B * fl / dtm / 1e12}
            res101 = dict(out["in_flight"])
            res101.update({"workload": "batch=%d single-scale %dx%d (WxH) ResNet-101 DeeperCut forward, %s — BASELINE.json's model name; "
                                       "same protocol as the headline (device-resident, %d forwards in flight on library-chosen streams)"
                                       % (B, W, H, args.dtype, S),
                           "gflop_per_image": fl / 1e9, "launches_per_forward": n101.num_launches(), "forwards_in_flight": S,
                           "stream_choice": choice, "one_forward_at_a_time": out["one_forward_at_a_time"],
                           "roofline_frac_one_at_a_time": out["one_forward_at_a_time"]["tflops"] / (PEAK_FP16_MFMA_TFLOPS if args.dtype == "f16" else PEAK_FP32_MFMA_TFLOPS)})
            return res101

        n_pcie = max(3, min(20, args.steps))
        if world == 1:
            beside("pcie_inclusive_pipelined", pcie_inclusive_pipelined)
            beside("pcie_inclusive", pcie_inclusive)
            beside("pcie_inclusive_image_entry", image_entry)
            beside("pycaffe_forward", pycaffe_forward)
        if world == 1 and args.config == 1 and args.coalesce > 1 and B == 1:
            beside("cross_request_batching", cross_request_batching)
        if world == 1 and args.config == 1 and not args.no_resnet101 and args.depth != 101:
            beside("resnet101", resnet101)
        if world == 1 and args.dtype == "f32" and args.config == 1 and not args.no_f16_line:
            # the other single-GPU configuration of BASELINE.json, timed by the same run
            beside("config2_f16", lambda: config2_f16_line(caffe, layers, args.depth, 10, dev, inject_weights,
                                                          int(os.environ.get("DC_BENCH_F16_EXECS", "2")), not args.no_tune_in_flight, max(1, args.regions)))
        if args.breakdown:
            net.blobs["data"].data[...] = x.cpu().numpy()
            net.forward()
            with open(args.breakdown, "w") as f:
                f.write(net.plan_text())
                f.write(net.profile_text(20))
        if world == 1 and not args.no_cpu_baseline:
            beside("cpu_baseline", lambda: cpu_baseline(lambda h, w: deepercut_prototxt(args.depth, h, w), layers, flops_img, (H, W)))
            if "value" in res["cpu_baseline"]:
                res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
