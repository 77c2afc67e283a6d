"""Several independent forwards in flight on one GPU, coalesced into small batches as they queue up.

A batch-1 layer of the DeeperCut net is a 5-25 us problem that fills ~3/4 of the 256 CUs, a third of it fixed cost, so ONE
forward at a time reaches ~50 % of the fp32-MFMA roof.  `Pipeline` keeps `depth` executors busy (a Net and its clones: own
activations, own HIP stream, own hipGraph, SHARED parameters, packed weights and tile choices).  Requests are single images
(or whole batches with coalescing off); by default they are coalesced OPPORTUNISTICALLY: a request goes out at once, alone, if
an executor is free; while all executors are busy requests queue up, and the executor that frees next takes whatever is queued
— up to `max_batch` same-shape requests — as ONE batch forward (cross-request batching, dc_net_forward_requests: the launches
fill the chip and pay their fixed cost once per batch).  Light load: batch-1 latency; heavy load: batch-`max_batch` throughput.
`coalesce=k` (an int) is the fixed policy of rounds 2-3: hold requests until k are there.
Device-resident interface: the caller owns NCHW float32 device buffers (e.g. torch CUDA tensors) and keeps them alive
until the request's tag comes back from wait_one() / drain().  Three executors is the useful maximum: a HIP process has
four hardware queues, and more concurrent kernels only evict each other's tiles from the 4 MB L2s.
Host-resident interface (round 5): `submit_host` takes HOST arrays in and out (pinned ones from `caffe.pinned_empty` are
copied by the DMA engines): the upload, the forward and the downloads of a request sit on its executor's own stream, so with
`depth` executors the copies of one request run beside the kernels of the others — the reference's blocking SyncedMemory copies
(src/caffe/syncedmem.cpp:25-77) one request at a time reach 293 images/s on this path, the bus needs 7.3 GB/s for 490.
WHICH streams the executors run on is chosen by the library the first time a shape is met (`caffe.choose_streams`: the real
forwards timed on a process-wide pool of candidate streams; hardware-queue sharing is worth 380 ... 490 images/s at depth 4).
Return values (changed in round 4): submit() returns None — the request may be queued, no executor is known yet — and
wait_one() / drain() return tags in the order requests FINISH, which across executors is not the order of submission.
The reference forwards one image at a time (src/caffe/layers/conv_layer.cpp:31): nothing to mirror.
"""
import collections
import os
import time


class Pipeline(object):
    LATENCY_WINDOW = 1 << 16

    def __init__(self, net, depth=3, coalesce=None, max_batch=4, max_queue=None, choose_streams=True, auto_tune=None):
        """auto_tune: None = on exactly when DC_TUNE_CACHE names a file (and DC_PIPELINE_AUTOTUNE is not 0): the first time this
        process meets a device-resident request shape that the cache's side-car (`<cache>.inflight`) does not list for this
        depth / batching, the tiles are re-tuned under THIS pipeline's load on scratch buffers (`tune`, a few seconds, untimed
        set-up like the autotuner's) and the overrides go into the cache file — a later process starts on them.  bench.py runs
        the same descent for its `value`; without this a caller of the product ran the latency tiles, 1-2 % slower in flight."""
        self.nets = [net] + [net.clone() for _ in range(max(1, depth) - 1)]
        cache = os.environ.get("DC_TUNE_CACHE")
        self._auto_tune = (bool(cache) and os.environ.get("DC_PIPELINE_AUTOTUNE", "1") != "0") if auto_tune is None else bool(auto_tune)
        self._auto_tuned = set()   # shapes seen by this pipeline
        self.auto_tune_report = {}  # shape key -> what tune() returned
        self._choose_streams = bool(choose_streams) and os.environ.get("DC_STREAM_CHOICE", "1") != "0"
        self.stream_choice = None  # what caffe.choose_streams measured, once it ran
        self.opportunistic = coalesce is None
        self.coalesce = 1 if coalesce is None else max(1, int(coalesce))
        self.max_batch = max(1, int(max_batch)) if self.opportunistic else self.coalesce
        # requests that may wait for an executor before submit() blocks: one full batch per executor
        self.max_queue = int(max_queue) if max_queue is not None else self.max_batch * len(self.nets)
        self._next = 0
        self._pending = collections.deque()  # (executor, [(tag, submit time)]) in launch order
        self._held = collections.deque()     # queued requests: (in, h, w, prob, loc, next, tag, n, submit time)
        self._done = collections.deque()     # tags of finished requests not yet handed back
        # seconds from submit() to the moment the request was seen finished: the last LATENCY_WINDOW requests (a service runs for days)
        self.latencies = collections.deque(maxlen=self.LATENCY_WINDOW)
        self.batch_sizes = collections.Counter()

    @property
    def depth(self):
        return len(self.nets)

    # ---- launching ---------------------------------------------------------------------------------------------------
    def _free_executor(self):
        busy = set(k for k, _ in self._pending)
        for i in range(len(self.nets)):
            k = (self._next + i) % len(self.nets)
            if k not in busy:
                return k
        return None

    def _launch(self, reqs, k=None):
        if k is None:
            if len(self._pending) >= len(self.nets):
                self._wait_group()
            k = self._free_executor()
        self._next = (k + 1) % len(self.nets)
        h, w = reqs[0][1], reqs[0][2]
        if self._choose_streams and len(self.nets) > 1 and not self._pending:
            # first launch: every executor lowers / allocates / tunes the shape, then the library picks their streams by timing
            # them together (untimed set-up, like the autotuner's)
            import caffe

            self._choose_streams = False
            for e in self.nets:
                e.reserve(reqs[0][7] if len(reqs) == 1 else len(reqs), h, w)
            self.stream_choice = caffe.choose_streams(self.nets)
        if self._auto_tune and not self._pending and not (len(reqs[0]) > 9 and reqs[0][9] is not None):
            self._maybe_auto_tune(reqs[0][7] if len(reqs) == 1 else 1, h, w)
        if len(reqs[0]) > 9 and reqs[0][9] is not None:  # host request: (x, prob, loc_pred, next_pred) arrays
            self.nets[k].forward_host_async(*reqs[0][9])
        elif len(reqs) == 1 and (reqs[0][7] != 1 or self.max_batch == 1):
            r = reqs[0]
            self.nets[k].forward_device(r[0], r[7], h, w, r[3], r[4], r[5], stream="own")
        else:
            self.nets[k].forward_requests([r[0] for r in reqs], h, w, [r[3] for r in reqs], [r[4] for r in reqs],
                                          [r[5] for r in reqs], stream="own")
        self.batch_sizes[len(reqs)] += 1
        self._pending.append((k, [(r[6], r[8]) for r in reqs]))
        return k

    def _reap(self):
        """Collect every group that has finished, without blocking (the streams are independent: any of them may be first)."""
        now = None
        keep = collections.deque()
        while self._pending:
            k, tags = self._pending.popleft()
            if self.nets[k].busy():
                keep.append((k, tags))
                continue
            now = now or time.perf_counter()
            for tag, t0 in tags:
                self._done.append(tag)
                self.latencies.append(now - t0)
        self._pending = keep

    def _take_batch(self):
        first = self._held.popleft()
        reqs = [first]
        host = lambda r: len(r) > 9 and r[9] is not None  # noqa: E731  (host requests are batches of their own)
        while self._held and len(reqs) < self.max_batch and first[7] == 1 and self._held[0][7] == 1 and not host(first) and \
                not host(self._held[0]) and (self._held[0][1], self._held[0][2]) == (first[1], first[2]):
            reqs.append(self._held.popleft())
        return reqs

    def _pump(self):
        self._reap()
        while self._held:
            k = self._free_executor() if len(self._pending) < len(self.nets) else None
            if k is None:
                if len(self._held) <= self.max_queue:
                    return
                self._wait_group()  # the queue is full: wait for the oldest group, then hand it more work
                continue
            self._launch(self._take_batch(), k)

    def submit(self, in_ptr, n, h, w, prob_ptr=None, loc_ptr=None, next_ptr=None, tag=None):
        """Enqueue one forward (asynchronous).  Opportunistic mode: launched at once if an executor is free, else queued (and
        coalesced with its neighbours when one frees); blocks only when `max_queue` requests are waiting.  Fixed mode
        (coalesce=k, single-image requests only): held until k same-shape requests are there (or flush() / drain())."""
        req = (in_ptr, h, w, prob_ptr, loc_ptr, next_ptr, tag, n, time.perf_counter())
        if self.opportunistic:
            self._held.append(req)
            self._pump()
            return None
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

    def submit_host(self, x, prob=None, loc_pred=None, next_pred=None, tag=None):
        """Enqueue one host-in / host-out forward (asynchronous): x float32 [n,3,H,W], the outputs C-contiguous float32 arrays of
        the maps' shapes or None.  Every array must stay alive and untouched until the request's tag comes back from wait_one() /
        drain(); arrays from caffe.pinned_empty() are moved by the DMA engines while other executors compute."""
        n, _c, h, w = x.shape
        req = (None, h, w, None, None, None, tag, n, time.perf_counter(), (x, prob, loc_pred, next_pred))
        if self.opportunistic:
            self._held.append(req)
            self._pump()
            return None
        return self._launch([req])

    def flush(self):
        """Launch whatever is queued as it is (partial batches included)."""
        while self._held:
            self._launch(self._take_batch())

    def _wait_group(self):
        k, tags = self._pending.popleft()
        self.nets[k].synchronize()
        now = time.perf_counter()
        for tag, t0 in tags:
            self._done.append(tag)
            self.latencies.append(now - t0)

    def wait_one(self):
        """Block until a request has finished; returns its tag (requests of one executor finish in order)."""
        if not (self._done or self._pending or self._held):
            raise RuntimeError("wait_one() with nothing submitted")
        if not self._done:
            if self.opportunistic:
                self._pump()
                if not self._done and not self._pending:
                    self.flush()
            else:
                self.flush()
            if not self._done:
                self._wait_group()
            if self.opportunistic:
                self._pump()  # an executor is free now: give it the queued requests before returning to the caller
        return self._done.popleft()

    def drain(self):
        self.flush()
        while self._pending:
            self._wait_group()
        tags = list(self._done)
        self._done.clear()
        return tags

    def latency_percentiles(self, qs=(50, 99)):
        """Request latency (submit -> seen finished) in milliseconds at the given percentiles, over everything since the last
        reset_stats()."""
        v = sorted(self.latencies)
        if not v:
            return {}
        return {q: v[min(len(v) - 1, int(round(q / 100.0 * (len(v) - 1))))] * 1e3 for q in qs}

    def reset_stats(self):
        self.latencies = collections.deque(maxlen=self.LATENCY_WINDOW)
        self.batch_sizes = collections.Counter()

    def _maybe_auto_tune(self, n, h, w):
        """first sight of a device-resident request shape with the pipeline idle: tune the tiles for this pipeline's load unless the
        side-car of the tune cache says an earlier process already did (see __init__)"""
        import json

        key = "%s n%d %dx%d depth%d batch%d" % (getattr(self.nets[0], "dtype", "f32"), n, h, w, len(self.nets), self.max_batch)
        if key in self._auto_tuned:
            return
        self._auto_tuned.add(key)
        side = (os.environ.get("DC_TUNE_CACHE") or "") + ".inflight"
        done = []
        try:
            done = json.load(open(side))
        except Exception:
            pass
        if key in done or self._held:
            return
        try:
            import torch
        except Exception:
            return
        dev = torch.device("cuda", int(getattr(self.nets[0], "device", 0) or 0))
        c = self.nets[0].blobs["data"].channels
        per = self.max_batch * len(self.nets)
        self.nets[0].reserve(n, h, w)
        shp = [tuple(self.nets[0].blobs[k].shape) for k in ("prob", "loc_pred", "next_pred")]
        x = torch.randn(n, c, h, w, device=dev) * 50
        outs = [[torch.empty(s, device=dev) for s in shp] for _ in range(per)]
        reqs = [(x.data_ptr(), n, h, w) + tuple(o.data_ptr() for o in outs[i]) for i in range(per)]
        auto, self._auto_tune = self._auto_tune, False  # (tune() launches through this pipeline's own path)
        try:
            self.auto_tune_report[key] = self.tune(reqs, rounds=2)
            torch.cuda.synchronize(dev)
        finally:
            self._auto_tune = auto
        if os.environ.get("DC_TUNE_CACHE"):
            try:
                done.append(key)
                tmp = side + ".tmp.%d" % os.getpid()
                json.dump(sorted(set(done)), open(tmp, "w"))
                os.replace(tmp, side)
            except OSError:
                pass

    def tune(self, requests, rounds=4, **kw):
        """Re-tune the tiles of the requests' shape for THIS pipeline's load (depth executors, this coalescing): the
        autotuner chose them for one forward alone.  `requests`: a list of submit() argument tuples (in_ptr, n, h, w,
        prob_ptr, loc_ptr, next_ptr) of one shape whose buffers may be overwritten — a multiple of the batch size times the
        depth, so that every executor meets the same batch shape (a partial batch would put one executor on another plan);
        the burst is replayed `rounds` times per measurement.  Returns what deepcut_tools.tune_in_flight returns."""
        from .tuning import tune_in_flight

        if self._pending or self._held:
            raise RuntimeError("tune() wants an idle pipeline")
        per = self.max_batch * len(self.nets)
        if len(requests) == 0 or len(requests) % per:
            raise ValueError("tune() wants a multiple of batch size x depth = %d requests (got %d)" % (per, len(requests)))
        fixed = Pipeline.__new__(Pipeline)  # the same executors under the fixed policy: every batch full, every run alike
        fixed.__dict__.update(self.__dict__)
        fixed.opportunistic, fixed.coalesce = False, self.max_batch
        fixed._pending, fixed._held, fixed._done = collections.deque(), collections.deque(), collections.deque()
        fixed.latencies, fixed.batch_sizes = collections.deque(maxlen=self.LATENCY_WINDOW), collections.Counter()

        def load():
            t0 = time.perf_counter()
            for _ in range(rounds):
                for r in requests:
                    fixed.submit(*r)
            fixed.drain()
            return time.perf_counter() - t0

        load()  # every executor has lowered (and tuned, for latency) the shape
        return tune_in_flight(self.nets, load, **kw)
