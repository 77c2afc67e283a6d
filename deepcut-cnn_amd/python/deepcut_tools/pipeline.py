"""Several independent forwards in flight on one GPU, optionally coalesced into small batches.

A batch-1 layer of the DeeperCut net is a 5-25 us problem that fills ~3/4 of the 256 CUs, a third of it fixed cost, so ONE
forward at a time reaches ~48 % of the fp32-MFMA roof.  `Pipeline` keeps `depth` executors busy (a Net and its clones: own
activations, own HIP stream, own hipGraph, SHARED parameters, packed weights and tile choices); requests go to them
round-robin and the kernels of one fill the CUs and gaps another leaves idle (3 executors: x1.4).  With `coalesce` = k > 1,
k consecutive same-shape requests are first merged into ONE batch-k forward (cross-request batching,
dc_net_forward_requests): the launches fill the chip and pay their fixed cost once per k images.
Device-resident interface: the caller owns NCHW float32 device buffers (e.g. torch CUDA tensors) and keeps them alive
until the request's tag comes back from wait_one() / drain().  Three executors is the useful maximum: a HIP process has
four hardware queues, and more concurrent kernels only evict each other's tiles from the 4 MB L2s.
"""
import collections


class Pipeline(object):
    def __init__(self, net, depth=3, coalesce=1):
        self.nets = [net] + [net.clone() for _ in range(max(1, depth) - 1)]
        self.coalesce = max(1, int(coalesce))
        self._next = 0
        self._pending = collections.deque()  # (executor, [tags])
        self._held = []                      # requests waiting for their batch to fill: (in, h, w, prob, loc, next, tag)
        self._done = collections.deque()     # tags of finished requests not yet handed back

    @property
    def depth(self):
        return len(self.nets)

    def _launch(self, reqs):
        if len(self._pending) >= len(self.nets):
            self._wait_group()
        k = self._next
        self._next = (k + 1) % len(self.nets)
        h, w = reqs[0][1], reqs[0][2]
        if len(reqs) == 1 and self.coalesce == 1:
            r = reqs[0]
            self.nets[k].forward_device(r[0], r[7], h, w, r[3], r[4], r[5], stream="own")
        else:
            self.nets[k].forward_requests([r[0] for r in reqs], h, w, [r[3] for r in reqs], [r[4] for r in reqs],
                                          [r[5] for r in reqs], stream="own")
        self._pending.append((k, [r[6] for r in reqs]))
        return k

    def submit(self, in_ptr, n, h, w, prob_ptr=None, loc_ptr=None, next_ptr=None, tag=None):
        """Enqueue one forward (asynchronous).  If every executor is busy, waits for the oldest group first.  With
        coalesce > 1 (single-image requests only) the request is held until `coalesce` same-shape requests are there (or
        flush() / drain() is called)."""
        req = (in_ptr, h, w, prob_ptr, loc_ptr, next_ptr, tag, n)
        if self.coalesce == 1:
            return self._launch([req])
        if n != 1:
            raise ValueError("coalescing takes single-image requests")
        if self._held and (self._held[0][1], self._held[0][2]) != (h, w):
            self.flush()
        self._held.append(req)
        if len(self._held) >= self.coalesce:
            self.flush()
        return None

    def flush(self):
        """Launch the held requests as they are (a partial batch)."""
        if self._held:
            held, self._held = self._held, []
            self._launch(held)

    def _wait_group(self):
        k, tags = self._pending.popleft()
        self.nets[k].synchronize()
        self._done.extend(tags)

    def wait_one(self):
        """Block until the oldest in-flight request has finished; returns its tag."""
        if not self._done:
            self.flush()
            self._wait_group()
        return self._done.popleft()

    def drain(self):
        self.flush()
        while self._pending:
            self._wait_group()
        tags = list(self._done)
        self._done.clear()
        return tags

    def tune(self, requests, rounds=4, **kw):
        """Re-tune the tiles of the requests' shape for THIS pipeline's load (depth executors, this coalescing): the
        autotuner chose them for one forward alone.  `requests`: a list of submit() argument tuples (in_ptr, n, h, w,
        prob_ptr, loc_ptr, next_ptr) of one shape, at least depth * coalesce of them, whose buffers may be overwritten; the
        burst is replayed `rounds` times per measurement.  Returns what deepcut_tools.tune_in_flight returns."""
        import time

        from .tuning import tune_in_flight

        if self._pending or self._held:
            raise RuntimeError("tune() wants an idle pipeline")

        def load():
            t0 = time.perf_counter()
            for _ in range(rounds):
                for r in requests:
                    self.submit(*r)
            self.drain()
            return time.perf_counter() - t0

        load()  # every executor has lowered (and tuned, for latency) the shape
        return tune_in_flight(self.nets, load, **kw)
