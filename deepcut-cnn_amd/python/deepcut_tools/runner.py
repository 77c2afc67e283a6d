"""Multi-scale / multi-crop pose estimation sharded over the GPUs of a node (BASELINE configs 3-5).

Every (image, scale) pair is an independent forward, so the work is dealt out statically:
  1. work items = (image index, scale), cost ~ pixels of the net input;
  2. longest-processing-time-first assignment to ranks (`lpt_shards`, identical on every rank — no
     scheduling traffic);
  3. on each rank, items with the same net-input shape are forwarded as ONE batch
     (`Net.forward_batch`), the pose is decoded on the device (`Net.decode_pose`: arg-max + location
     refinement, estimate_pose.py:131-143), optionally the three maps are kept (multi-person consumers
     need `next_pred`).  With the default pre-processing and a net that has `forward_images`, items of the
     same source size and scale go through the image entry instead: uint8 pixels up, pre-processing
     (estimate_pose.py:83-103, bit-exact) + forward + decode on the device, 70 doubles per item back;
  4. ONE exchange: poses (70 doubles per item) and, if asked for, the maps are gathered to rank 0
     (`gather_maps`: grouped send/recv, variable sizes);
  5. rank 0 keeps, per image, the scale whose minimum joint confidence is highest (estimate_pose.py:119-126).
Single-process use (no process group) is the world-size-1 case of the same code.
"""
import numpy as np

from .shard import gather_maps, lpt_shards


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


class ShardedPoseRunner(object):
    """`net`: a caffe.Net of this package (or anything with forward_batch(images)->dict and
    decode_pose(scale)->[n,5,J]).  `preprocess(image, scale) -> HxWx3 float32` defaults to pose.estimate_pose's."""

    def __init__(self, net, preprocess=None, group=None, max_batch=16, device=None, device_preprocess=True, depth=1):
        """depth > 1 (image entry, poses only): that many batches are kept in flight on this GPU, each on its own
        executor (`net.clone()`: shared weights) and stream — upload, pre-processing, forward and decode of one batch
        overlap the others'; results are identical to depth 1."""
        self.net = net
        self.group = group
        self.max_batch = max_batch
        self.device = device
        self.depth = max(1, int(depth))
        self._execs = None
        self.image_entry = bool(device_preprocess and preprocess is None and hasattr(net, "forward_images"))
        if preprocess is None:
            from pose.estimate_pose import preprocess as _pp

            preprocess = _pp
        self.preprocess = preprocess

    def _run_in_flight(self, images, items, by_source, poses):
        """Poses of the local work items with `depth` batches in flight (device buffers through torch)."""
        import torch

        if self._execs is None:
            self._execs = [self.net] + [self.net.clone() for _ in range(self.depth - 1)]
            self._streams = [torch.cuda.Stream() for _ in self._execs]
        nj = self.net.blobs["prob"].shape[1]
        busy = [None] * self.depth

        def finish(e):
            if busy[e] is None:
                return
            chunk, pose_t, _img_t = busy[e]
            self._streams[e].synchronize()
            host = pose_t.cpu().numpy()
            for j, k in enumerate(chunk):
                poses[k] = host[j]
            busy[e] = None

        slot = 0
        for key in sorted(by_source):
            ks, s = by_source[key], key[1]
            h, w = key[0]
            for b0 in range(0, len(ks), self.max_batch):
                chunk = ks[b0:b0 + self.max_batch]
                e = slot % self.depth
                slot += 1
                finish(e)
                st = self._streams[e]
                with torch.cuda.stream(st):
                    img_t = torch.from_numpy(np.stack([images[items[k][0]] for k in chunk])).cuda(non_blocking=True)
                    pose_t = torch.empty((len(chunk), 5, nj), dtype=torch.float64, device="cuda")
                self._execs[e].forward_images_device(img_t.data_ptr(), len(chunk), h, w, s, pose_ptr=pose_t.data_ptr(),
                                                     stream=st.cuda_stream)
                busy[e] = (chunk, pose_t, img_t)
        for e in range(self.depth):
            finish(e)

    def run(self, images, scales, want_maps=False):
        """images: list of HxWx3 BGR uint8 (the same list on every rank).  Returns on rank 0 a dict
        {"poses": [5xJ or None per image], "best_scale": [...], "items": [...], "item_poses": array,
         "maps": {item index: dict} if want_maps}; None on the other ranks."""
        import torch

        dist = _dist()
        world = dist.get_world_size(self.group) if dist else 1
        rank = dist.get_rank(self.group) if dist else 0
        items, shards = plan_work([im.shape[:2] for im in images], scales, world)
        mine = shards[rank]
        by_shape = {}
        by_source = {}
        for k in mine:
            if self.image_entry:
                by_source.setdefault((images[items[k][0]].shape[:2], items[k][1]), []).append(k)
            else:
                by_shape.setdefault(items[k][2], []).append(k)
        poses = {}
        maps = {}
        if by_source and self.depth > 1 and not want_maps and hasattr(self.net, "forward_images_device"):
            self._run_in_flight(images, items, by_source, poses)
            by_source = {}
        for key in sorted(by_source):
            ks, s = by_source[key], key[1]
            for b0 in range(0, len(ks), self.max_batch):
                chunk = ks[b0:b0 + self.max_batch]
                out = self.net.forward_images(np.stack([images[items[k][0]] for k in chunk]), s,
                                              want=("prob", "loc_pred", "next_pred") if want_maps else (), pose=True)
                for j, k in enumerate(chunk):
                    poses[k] = out["pose"][j]
                    if want_maps:
                        maps[k] = {name: out[name][j].copy() for name in ("prob", "loc_pred", "next_pred")}
        for hw in sorted(by_shape):
            ks = by_shape[hw]
            for b0 in range(0, len(ks), self.max_batch):
                chunk = ks[b0:b0 + self.max_batch]
                batch = np.stack([self.preprocess(images[items[k][0]], items[k][1]).transpose(2, 0, 1) for k in chunk])
                out = self.net.forward_batch(np.ascontiguousarray(batch, np.float32),
                                             want=("prob", "loc_pred", "next_pred") if want_maps else ())
                # one decode per distinct scale in the chunk (the division by the scale is part of the decode)
                for s in sorted(set(items[k][1] for k in chunk)):
                    dec = self.net.decode_pose(s)
                    for j, k in enumerate(chunk):
                        if items[k][1] == s:
                            poses[k] = dec[j]
                if want_maps:
                    for j, k in enumerate(chunk):
                        maps[k] = {name: out[name][j].copy() for name in out}
        nj = next(iter(poses.values())).shape[1] if poses else 14
        ncol = 2 + 5 * nj  # item index, channels of next_pred (0 when maps are not kept), the pose
        local = np.zeros((len(mine), ncol), np.float64)
        for row, k in enumerate(mine):
            local[row, 0] = k
            local[row, 1] = maps[k]["next_pred"].shape[0] if want_maps else 0
            local[row, 2:] = poses[k].reshape(-1)
        if world > 1:
            dev = self.device if self.device is not None else "cpu"
            got = gather_maps(torch.from_numpy(local).to(dev), dst=0, group=self.group)
            map_bufs = None
            if want_maps:
                flat = np.concatenate([np.concatenate([maps[k][n].reshape(-1) for n in ("prob", "loc_pred", "next_pred")])
                                       for k in mine]) if mine else np.zeros(0, np.float32)
                map_bufs = gather_maps(torch.from_numpy(flat.astype(np.float32)).to(dev), dst=0, group=self.group)
            if rank != 0:
                return None
            table = np.concatenate([g.cpu().numpy().reshape(-1, ncol) for g in got if g.numel()], axis=0)
            if want_maps:
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
