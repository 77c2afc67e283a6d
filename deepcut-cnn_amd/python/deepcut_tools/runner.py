"""Multi-scale / multi-crop pose estimation sharded over the GPUs of a node (BASELINE configs 3-5).

Every (image, scale) pair is an independent forward, so the work is dealt out statically:
  1. work items = (image index, scale), cost ~ pixels of the net input;
  2. longest-processing-time-first assignment to ranks (`lpt_shards`, identical on every rank — no
     scheduling traffic); on a rank, items of the same source size and scale form batches (`rank_batches`);
     every rank can therefore compute every other rank's batch list and buffer sizes;
  3. device pipeline (a caffe.Net of this package on a GPU): per batch the uint8 pixels go up, pre-processing
     (estimate_pose.py:83-103, bit-exact), forward and pose decode (estimate_pose.py:131-143) run on the device,
     `depth` batches in flight on executor clones; poses and — if asked for — the three maps stay in HBM
     (maps in the net's own element type: float16 payloads from a float16 net);
  4. exchange: the maps of batch k are sent to rank 0 as soon as batch k is done, on a side stream, while batch
     k+1 computes (round k = one grouped send/recv: every rank that has a k-th batch sends it, rank 0 posts the
     matching receives; sizes come from the schedule, no headers); the poses (70 doubles per item) follow in one
     final gather;
  5. rank 0 keeps, per image, the scale whose minimum joint confidence is highest (estimate_pose.py:119-126).
A host pipeline (any object with forward_batch(images)->dict and decode_pose(scale)->[n,5,J], or a custom
`preprocess`) does the same with NumPy buffers and one gather at the end; single-process use (no process
group) is the world-size-1 case of the same code.
"""
import numpy as np

from .shard import gather_maps, gather_maps_known, lpt_shards

MAP_NAMES = ("prob", "loc_pred", "next_pred")


def _dist():
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def net_input_shape(image_hw, scale, stride=8):
    """(H, W) of the net input for an image at a scale (estimate_pose.py:85-88)."""
    h, w = image_hw
    return (int(np.ceil(float(h) * scale / stride) * stride), int(np.ceil(float(w) * scale / stride) * stride))


def plan_work(image_shapes, scales, world):
    """-> (items, shards): items = [(image index, scale, (H, W))], shards[r] = item indices of rank r."""
    items = []
    for i, hw in enumerate(image_shapes):
        for s in scales:
            items.append((i, float(s), net_input_shape(hw, s)))
    costs = [hw[0] * hw[1] for _, _, hw in items]
    return items, lpt_shards(costs, world)


def rank_batches(image_shapes, items, shard, max_batch, interleave=False):
    """Batches of one rank, in execution order: [(source (h, w), scale, net input (H, W), [item indices])].  A pure
    function of the schedule, so every rank knows every other rank's batches (and the size of every payload).
    interleave: round-robin over the (source size, scale) classes instead of class after class — the r-th batch of every
    class, largest net input first, then the (r+1)-th of every class —, so that consecutive batches have DIFFERENT shapes
    and can run as one grouped launch sequence (`group_units`)."""
    groups = {}
    for k in shard:
        groups.setdefault((tuple(image_shapes[items[k][0]]), items[k][1]), []).append(k)
    per_class = []
    # largest net input first: its buffers are allocated before the smaller shapes are met, so nothing grows later
    # (a reallocated buffer makes the captured graphs of the other shapes stale)
    for key in sorted(groups, key=lambda g: (-items[groups[g][0]][2][0] * items[groups[g][0]][2][1], g)):
        ks = groups[key]
        per_class.append([(key[0], key[1], items[ks[0]][2], ks[b0:b0 + max_batch]) for b0 in range(0, len(ks), max_batch)])
    if not interleave:
        return [b for c in per_class for b in c]
    out = []
    for r in range(max([len(c) for c in per_class]) if per_class else 0):
        out.extend(c[r] for c in per_class if r < len(c))
    return out


def group_units(batches, group_size):
    """Consecutive batches that run as ONE launch sequence: up to `group_size` batches of pairwise different (batch size, net
    input) shapes (two batches of one shape gain nothing from sharing a launch — they could have been one bigger batch — and
    lose the overlap of running on two executors).  -> [[batch indices]]."""
    units, cur, seen = [], [], set()
    for bi, b in enumerate(batches):
        key = (len(b[3]), tuple(b[2]))
        if cur and (key in seen or len(cur) >= group_size):
            units.append(cur)
            cur, seen = [], set()
        cur.append(bi)
        seen.add(key)
    if cur:
        units.append(cur)
    return units


class ShardedPoseRunner(object):
    """`net`: a caffe.Net of this package (or anything with forward_batch(images)->dict and
    decode_pose(scale)->[n,5,J]).  `preprocess(image, scale) -> HxWx3 float32` defaults to pose.estimate_pose's."""

    def __init__(self, net, preprocess=None, group=None, max_batch=16, device=None, device_preprocess=True, depth=1,
                 half_maps=None, group_size=4):
        """depth > 1 (device pipeline): that many batches are kept in flight on this GPU, each on its own executor
        (`net.clone()`: shared weights, shared tile choices) and stream — upload, pre-processing, forward and decode of
        one batch overlap the others'; results equal depth 1 up to the batch composition (same kernels, same order).
        group_size > 1 (device pipeline, default 4): up to that many consecutive batches of this rank — the scales of a
        pyramid, the crop sizes of a crowd image — run as ONE grouped launch sequence (caffe.NetGroup: every layer once over
        all of them, dc_group_forward_images) instead of one forward each; results equal the ungrouped run up to the fp32
        summation order of the tiles chosen.  half_maps: send float16 maps (default: whenever the net computes in float16)."""
        self.net = net
        self.group = group
        self.max_batch = max_batch
        self.device = device
        self.depth = max(1, int(depth))
        self.half_maps = half_maps
        self.group_size = max(1, int(group_size))
        self._execs = None
        self._members = None   # grouped: per executor slot, the member nets (a net and its clones)
        self._groups = None    # grouped: per executor slot, {members used: caffe.NetGroup}
        self._seen = set()
        self.image_entry = bool(device_preprocess and preprocess is None and hasattr(net, "forward_images"))
        if preprocess is None:
            from pose.estimate_pose import preprocess as _pp

            preprocess = _pp
        self.preprocess = preprocess

    # ------------------------------------------------------------------------------------------------------------
    def _torch_device(self):
        """The torch device of this runner's GPU work: the device the NET executes on (caffe.set_device), never torch's
        current device — a rank that set one and not the other must not hand cuda:0 buffers to kernels on another GPU."""
        import torch

        idx = getattr(self.net, "device", -1)
        if idx is None or idx < 0:
            import caffe

            idx = caffe.pycaffe._lib.dc_get_device()
        return torch.device("cuda", int(idx))

    def _comm_device(self, dist, dev):
        if self.device is not None:
            return self.device
        if dist is not None and dist.get_backend(self.group) == "nccl":
            if dev is None:  # host pipeline under RCCL: the tables still have to travel in device memory
                import torch

                dev = torch.device("cuda", torch.cuda.current_device())
            return dev  # RCCL moves device buffers
        return "cpu"

    def _run_device(self, images, items, batches_of, rank, world, want_maps, dist):
        """The device pipeline of this rank.  Returns (poses {item: [5,J]}, maps {item: dict of arrays} on rank 0 or {})."""
        import torch

        dev = self._torch_device()
        comm_dev = self._comm_device(dist, dev) if world > 1 else None
        G = self.group_size
        try:
            import caffe

            grouped = G > 1 and hasattr(caffe, "NetGroup") and isinstance(self.net, caffe.Net)
        except Exception:  # noqa: BLE001
            grouped = False
        with torch.cuda.device(dev):
            if self._execs is None:
                self._execs = [self.net] + [self.net.clone() for _ in range(self.depth - 1)]
                self._streams = [torch.cuda.Stream(dev) for _ in self._execs]
                self._comm_stream = torch.cuda.Stream(dev)
            if grouped and self._members is None:
                # slot e: executor e itself + G - 1 clones of it (member j of a unit runs batch j of the unit)
                self._members = [[ex] + [ex.clone() for _ in range(G - 1)] for ex in self._execs]
                self._groups = [dict() for _ in self._execs]
        chans = [self.net.blobs[n].shape[1] for n in MAP_NAMES]
        nj, ctot = chans[0], sum(chans)
        half = self.half_maps if self.half_maps is not None else getattr(self.net, "dtype", "f32") == "f16"
        mdtype = torch.float16 if half else torch.float32
        mine = batches_of[rank]
        busy = [None] * self.depth
        poses, payload, recv, uploaded = {}, {}, {}, {}
        reqs = []
        sent = []          # (request, batch index) of this rank's sends: the payload is dropped once the send has completed
        uses_left = {}     # image-id tuple -> batches still to be enqueued with it: the uint8 stack is dropped after the last one
        for (_src, _s, _hw, chunk) in mine:
            ids = tuple(items[k][0] for k in chunk)
            uses_left[ids] = uses_left.get(ids, 0) + 1
        checked_shapes = set()

        def msize(b):
            return len(b[3]) * ctot * (b[2][0] // 8) * (b[2][1] // 8)

        def finish(e):
            if busy[e] is None:
                return
            self._streams[e].synchronize()
            for chunk, pose_t, _img_t in busy[e]:
                host = pose_t.cpu().numpy()
                for j, k in enumerate(chunk):
                    poses[k] = host[j]
            busy[e] = None

        def exchange_round(k, done_event):
            """Round k of the map exchange on the side stream: this rank's k-th batch goes to rank 0 (if it has one),
            rank 0 receives the k-th batch of every rank that has one."""
            ops = []
            with torch.cuda.stream(self._comm_stream):
                if done_event is not None:
                    self._comm_stream.wait_event(done_event)
                if rank != 0 and k < len(mine):
                    buf = payload[k]
                    if str(comm_dev) == "cpu":
                        self._comm_stream.synchronize()
                        buf = buf.cpu()
                        payload[k] = buf  # keep the sent tensor alive until the request completes
                    ops.append(dist.P2POp(dist.isend, buf, 0, self.group))
                if rank == 0:
                    for r in range(1, world):
                        if k < len(batches_of[r]):
                            recv[(r, k)] = torch.empty(msize(batches_of[r][k]), dtype=mdtype, device=comm_dev)
                            ops.append(dist.P2POp(dist.irecv, recv[(r, k)], r, self.group))
                if ops:
                    rq = dist.batch_isend_irecv(ops)
                    reqs.extend(rq)
                    if rank != 0 and k < len(mine):
                        sent.append((rq[0], k))
            # device memory must not grow with the data set: a payload that has been delivered is released
            for rq0, kk in list(sent):
                if rq0.is_completed():
                    payload.pop(kk, None)
                    sent.remove((rq0, kk))

        rounds = max([len(b) for b in batches_of]) if want_maps and world > 1 else 0
        # a unit = the batches that run as one launch sequence: G consecutive batches grouped, else one batch
        units = group_units(mine, G if grouped else 1)
        for ui, unit in enumerate(units):
            e = ui % self.depth
            finish(e)
            key = tuple((len(mine[bi][3]), mine[bi][2]) for bi in unit)
            if key not in self._seen:
                # first meeting of a shape (set): its tiles are timed on the device inside this call — with the GPU to itself
                # (timings taken under the other executors' kernels are noise, and every executor shares the choice)
                for q in range(self.depth):
                    finish(q)
                self._seen.add(key)
            st = self._streams[e]
            held = []  # per batch of the unit: (chunk, pose tensor, image stack) — alive until the unit has finished
            mbufs = []
            for bi in unit:
                src_hw, s, in_hw, chunk = mine[bi]
                ids = tuple(items[k][0] for k in chunk)
                with torch.cuda.device(dev), torch.cuda.stream(st):
                    # the scales of a pyramid forward the SAME images: their uint8 pixels are stacked and uploaded once
                    if ids not in uploaded:
                        # image by image straight into the device tensor: `np.stack` first would copy the 1.2 MB images once more
                        # on the host (a millisecond per 8 images of a 12-ms pyramid batch)
                        first = np.asarray(images[ids[0]])
                        up = torch.empty((len(ids),) + first.shape, dtype=torch.uint8, device=dev)
                        for j, i in enumerate(ids):
                            up[j].copy_(torch.from_numpy(np.ascontiguousarray(images[i])), non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(st)
                        uploaded[ids] = (up, ev)
                    img_t, up_ev = uploaded[ids]
                    st.wait_event(up_ev)  # the upload may have been enqueued on another executor's stream
                    uses_left[ids] -= 1
                    if uses_left[ids] == 0:
                        del uploaded[ids]  # last scale of these images: `held` keeps the tensor alive until its forward is done
                    pose_t = torch.empty((len(chunk), 5, nj), dtype=torch.float64, device=dev)
                    mbufs.append(torch.empty(msize(mine[bi]), dtype=mdtype, device=dev) if want_maps else None)
                assert img_t.device == dev and pose_t.device == dev
                held.append((chunk, pose_t, img_t))
            if grouped and len(unit) > 1:
                members = self._members[e][:len(unit)]
                grp = self._groups[e].get(len(unit))
                if grp is None:
                    grp = self._groups[e][len(unit)] = caffe.NetGroup(members)
                grp.forward_images_device([h[2].data_ptr() for h in held], [(len(mine[bi][3]),) + tuple(mine[bi][0]) for bi in unit],
                                          [mine[bi][1] for bi in unit], pose_ptrs=[h[1].data_ptr() for h in held], stream=st.cuda_stream)
            else:
                members = [self._execs[e]]
                src_hw, s, _in_hw, chunk = mine[unit[0]]
                members[0].forward_images_device(held[0][2].data_ptr(), len(chunk), src_hw[0], src_hw[1], s, pose_ptr=held[0][1].data_ptr(),
                                                 stream=st.cuda_stream)
            done = None
            if want_maps:
                for ex, bi, mbuf in zip(members, unit, mbufs):
                    _src, _s, in_hw, chunk = mine[bi]
                    skey = (len(chunk), in_hw)
                    if skey not in checked_shapes:
                        # the payload layout assumes stride-8 maps (msize): check it against the net once per shape instead of
                        # letting emit_maps write past the buffer for a model with another stride
                        checked_shapes.add(skey)
                        for name, c in zip(MAP_NAMES, chans):
                            got = tuple(ex.blobs[name].shape)
                            want = (len(chunk), c, in_hw[0] // 8, in_hw[1] // 8)
                            if got != want:
                                raise ValueError("map %r has shape %s, the exchange expects %s (stride-8 maps)" % (name, got, want))
                    hw8 = (in_hw[0] // 8) * (in_hw[1] // 8) * len(chunk)
                    o1, o2 = chans[0] * hw8, (chans[0] + chans[1]) * hw8
                    ex.emit_maps_device(mbuf[:o1].data_ptr(), mbuf[o1:o2].data_ptr(), mbuf[o2:].data_ptr(), half=half,
                                        stream=st.cuda_stream)
                    payload[bi] = mbuf
                done = torch.cuda.Event()
                done.record(st)
            busy[e] = held
            for bi in unit:
                if bi < rounds:
                    exchange_round(bi, done)
        for e in range(self.depth):
            finish(e)
        for k in range(len(mine), rounds):  # rounds in which this rank has nothing to send (rank 0 may still receive)
            exchange_round(k, None)
        for q in reqs:
            q.wait()
        if reqs:
            self._comm_stream.synchronize()
        maps = {}
        if want_maps and rank == 0:
            # All maps of a run land in ONE pinned float32 host buffer and are handed out as views of it; the buffer is taken
            # again by a later run once nobody references those views any more (else a new one is made).  A device-to-host copy
            # straight into fresh pageable memory faults every page inside the driver's pinning path — the 39-ms host entry of
            # round 3; here 237 MB per run of configs[4]: 320-520 crops/s depending on the box's page luck.
            todo = [(payload[bi], b) for bi, b in enumerate(mine)] + [(buf, batches_of[r][k]) for (r, k), buf in recv.items()]
            total = sum(int(buf.numel()) for buf, _b in todo)
            stage_t, stage_a = self._result_stage(total)
            off = 0
            for buf, b in todo:
                n, (h, w) = len(b[3]), (b[2][0] // 8, b[2][1] // 8)
                m = int(buf.numel())
                stage_t[off:off + m].copy_(buf)  # (converts float16 payloads on the way)
                p = off
                parts = []
                for c in chans:
                    parts.append(stage_a[p:p + n * c * h * w].reshape(n, c, h, w))
                    p += n * c * h * w
                for j, k in enumerate(b[3]):
                    maps[k] = {name: parts[q][j] for q, name in enumerate(MAP_NAMES)}
                off += m
        return poses, maps

    def _result_stage(self, numel):
        """(pinned float32 host tensor of `numel` elements, numpy view of it) for the maps of one run: the memory of an earlier
        run is taken again when nothing refers to the array handed out for it any more — the pool keeps the tensor (the memory) and
        only a WEAK reference to the hand-out; views keep their base array alive, so the weak reference dies with the last of
        them (no reference-count arithmetic: that is an implementation detail of the interpreter) —, else a new one is allocated; at
        most four are kept."""
        import weakref

        import torch

        pool = self.__dict__.setdefault("_stage_pool", [])
        for e in pool:
            if e[0].numel() >= numel and (e[1] is None or e[1]() is None):
                a = e[0].numpy()
                e[1] = weakref.ref(a)
                return e[0][:numel], a[:numel]
        try:
            t = torch.empty(max(numel, 1), dtype=torch.float32, pin_memory=True)
        except RuntimeError:
            t = torch.empty(max(numel, 1), dtype=torch.float32)
        a = t.numpy()
        pool.append([t, weakref.ref(a)])
        if len(pool) > 4:
            pool.pop(0)
        return t[:numel], a[:numel]

    # ------------------------------------------------------------------------------------------------------------
    def _run_host(self, images, items, mine, want_maps):
        """Host pipeline of this rank: pre-processing in NumPy (or the net's synchronous image entry), one batch at a time."""
        by_shape, by_source = {}, {}
        for k in mine:
            if self.image_entry:
                by_source.setdefault((images[items[k][0]].shape[:2], items[k][1]), []).append(k)
            else:
                by_shape.setdefault(items[k][2], []).append(k)
        poses, maps = {}, {}
        for key in sorted(by_source):
            ks, s = by_source[key], key[1]
            for b0 in range(0, len(ks), self.max_batch):
                chunk = ks[b0:b0 + self.max_batch]
                out = self.net.forward_images(np.stack([images[items[k][0]] for k in chunk]), s,
                                              want=MAP_NAMES if want_maps else (), pose=True)
                for j, k in enumerate(chunk):
                    poses[k] = out["pose"][j]
                    if want_maps:
                        maps[k] = {name: out[name][j].copy() for name in MAP_NAMES}
        for hw in sorted(by_shape):
            ks = by_shape[hw]
            for b0 in range(0, len(ks), self.max_batch):
                chunk = ks[b0:b0 + self.max_batch]
                batch = np.stack([self.preprocess(images[items[k][0]], items[k][1]).transpose(2, 0, 1) for k in chunk])
                out = self.net.forward_batch(np.ascontiguousarray(batch, np.float32), want=MAP_NAMES if want_maps else ())
                # one decode per distinct scale in the chunk (the division by the scale is part of the decode)
                for s in sorted(set(items[k][1] for k in chunk)):
                    dec = self.net.decode_pose(s)
                    for j, k in enumerate(chunk):
                        if items[k][1] == s:
                            poses[k] = dec[j]
                if want_maps:
                    for j, k in enumerate(chunk):
                        maps[k] = {name: out[name][j].copy() for name in out}
        return poses, maps

    def _use_device_pipeline(self):
        if not (self.image_entry and hasattr(self.net, "forward_images_device") and hasattr(self.net, "emit_maps_device")):
            return False
        try:
            import torch

            return torch.cuda.is_available()
        except Exception:
            return False

    def _select_outputs(self, want_maps):
        """DC_OPT_OUTPUTS on every executor of this runner: poses need `prob` and `loc_pred` only (python/pose/estimate_pose.py:231-241),
        so without the maps the 364-channel `next_pred` head — 23.3 of the 241 GFLOP of a 544x736 forward — is not computed at all."""
        if not hasattr(self.net, "set_outputs"):
            return
        names = None if want_maps else [n for n in self.net.outputs if n in ("prob", "loc_pred")]
        if names is not None and len(names) != 2:
            names = None
        nets = [self.net] + [m for grp in (self._members or []) for m in grp] + list(self._execs or [])
        seen = set()
        for n in nets:
            if id(n) in seen:
                continue
            seen.add(id(n))
            want = sorted(n.outputs) if names is None else sorted(names)
            if sorted(n.wanted_outputs) != want:
                n.set_outputs(names)

    def run(self, images, scales, want_maps=False):
        """images: list of HxWx3 BGR uint8 (the same list on every rank).  Returns on rank 0 a dict
        {"poses": [5xJ or None per image], "best_scale": [...], "items": [...], "item_poses": array,
         "maps": {item index: dict} if want_maps}; None on the other ranks."""
        import torch

        dist = _dist()
        world = dist.get_world_size(self.group) if dist else 1
        rank = dist.get_rank(self.group) if dist else 0
        shapes = [im.shape[:2] for im in images]
        items, shards = plan_work(shapes, scales, world)
        mine = shards[rank]
        self._select_outputs(want_maps)
        on_device = self._use_device_pipeline()
        if on_device:
            batches_of = [rank_batches(shapes, items, shards[r], self.max_batch, interleave=self.group_size > 1) for r in range(world)]
            poses, maps = self._run_device(images, items, batches_of, rank, world, want_maps, dist)
        else:
            poses, maps = self._run_host(images, items, mine, want_maps)
        if poses:
            nj = next(iter(poses.values())).shape[1]
        elif hasattr(self.net, "blobs"):
            nj = self.net.blobs["prob"].shape[1]  # a rank without work items still sizes the table from the model
        else:
            nj = 0  # a rank without work items and without a model to ask: its table has no rows, the width is moot
        ncol = 2 + 5 * nj  # item index, channels of next_pred (0 when maps are not kept), the pose
        local = np.zeros((len(mine), ncol), np.float64)
        for row, k in enumerate(mine):
            local[row, 0] = k
            local[row, 1] = maps[k]["next_pred"].shape[0] if (want_maps and k in maps) else 0
            local[row, 2:] = poses[k].reshape(-1)
        if world > 1:
            dev = self._comm_device(dist, self._torch_device() if on_device else None)
            sizes = [len(shards[r]) * ncol for r in range(world)]  # known from the schedule: no header exchange
            got = gather_maps_known(torch.from_numpy(local).reshape(-1).to(dev), sizes, 0, self.group)
            map_bufs = None
            if want_maps and not on_device:  # host pipeline: the maps follow in one gather at the end
                flat = np.concatenate([np.concatenate([maps[k][n].reshape(-1) for n in MAP_NAMES])
                                       for k in mine]) if mine else np.zeros(0, np.float32)
                map_bufs = gather_maps(torch.from_numpy(flat.astype(np.float32)).to(dev), dst=0, group=self.group)
            if rank != 0:
                return None
            rows = [g.cpu().numpy().reshape(-1, ncol) for g in got if g.numel()]
            table = np.concatenate(rows, axis=0) if rows else np.zeros((0, ncol))
            if map_bufs is not None:
                next_ch = {int(row[0]): int(row[1]) for row in table}
                maps = {}
                for r in range(world):
                    buf = map_bufs[r].cpu().numpy()
                    p = 0
                    for k in shards[r]:  # same order as the sender packed them
                        h, w = items[k][2][0] // 8, items[k][2][1] // 8
                        d = {}
                        for name, c in (("prob", nj), ("loc_pred", 2 * nj), ("next_pred", next_ch[k])):
                            d[name] = buf[p:p + c * h * w].reshape(c, h, w)
                            p += c * h * w
                        maps[k] = d
        else:
            table = local
        item_poses = np.zeros((len(items), 5, nj))
        for row in table:
            item_poses[int(row[0])] = row[2:].reshape(5, nj)
        best = [None] * len(images)
        best_conf = [0.0] * len(images)
        best_scale = [None] * len(images)
        for k, (i, s, _hw) in enumerate(items):  # items are ordered by (image, scale): the reference's loop order
            c = float(item_poses[k][2].min())
            if c > best_conf[i]:
                best[i], best_conf[i], best_scale[i] = item_poses[k], c, s
        res = {"poses": best, "best_scale": best_scale, "items": items, "item_poses": item_poses}
        if want_maps:
            res["maps"] = maps
        return res
