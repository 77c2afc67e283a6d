"""Image sharding across the GPUs of one node and the gather of score maps.

The reference has no inference-time multi-GPU path (its only multi-GPU code is the training-time
P2PSync tree, src/caffe/parallel.cpp); every forward (image, scale, crop) is independent, so the MI355X
design is: one process per GPU, weights replicated, work items dealt out by longest-processing-time-
first over H*W, no data-path collective except ONE exchange — the gather of the output maps to rank 0
(RCCL send/recv over the xGMI mesh: 7 peers -> 7 distinct links into the root).
Backend-agnostic (`nccl` = RCCL on ROCm; `gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


def lpt_shards(costs, world):
    """Longest-processing-time-first assignment: returns a list (per rank) of item indices.  Items of
    equal cost degrade to round-robin, which is what BASELINE config 4 (64 equal images) needs."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    for o in out:
        o.sort()
    return out


def gather_maps(local, dst=0, group=None):
    """Gather one flat float32 tensor per rank to `dst`.  Shapes may differ per rank (multi-scale
    shards): sizes are exchanged first with a tiny all_gather, then the payload moves with one
    grouped send/recv per peer (root receives from each peer on its own link).  Returns the list of
    per-rank tensors on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    flat = local.contiguous().view(-1)
    n = torch.tensor([flat.numel()], dtype=torch.int64, device=flat.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    return gather_maps_known(flat, sizes, dst, group)


def gather_maps_known(flat, sizes, dst=0, group=None, out=None, async_op=False):
    """Same, with per-rank sizes known on every rank from the deterministic schedule (no header
    exchange).  `out` (dst only): preallocated list of receive tensors to reuse across steps."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [flat]
    if rank == dst:
        bufs = out if out is not None else [torch.empty(s, dtype=flat.dtype, device=flat.device) for s in sizes]
        # a rank with nothing to send posts no send (sizes are known on both sides), so no receive is posted for it
        ops = [dist.P2POp(dist.irecv, bufs[r], r, group) for r in range(world) if r != dst and sizes[r] > 0]
        reqs = dist.batch_isend_irecv(ops) if ops else []
        bufs[dst] = flat
        if async_op:
            return bufs, reqs
        for q in reqs:
            q.wait()
        return bufs
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, dst, group)]) if sizes[rank] > 0 else []
    if async_op:
        return None, reqs
    for q in reqs:
        q.wait()
    return None
