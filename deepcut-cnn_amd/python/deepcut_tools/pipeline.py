"""Several independent forwards in flight on one GPU.

A batch-1 layer of the DeeperCut net is a 5-25 us problem that fills ~3/4 of the 256 CUs and leaves a
~2 us gap before its dependent successor, so ONE forward at a time reaches ~42 % of the fp32-MFMA roof.
`Pipeline` keeps `depth` requests in flight: `depth` executors (a Net and its clones: own activations,
own HIP stream, own hipGraph, SHARED parameters and packed weights) take requests round-robin; the
kernels of request i+1 fill the CUs and gaps request i leaves idle (~60 % of the roof at depth 3).
Device-resident interface: the caller owns NCHW float32 device buffers (e.g. torch CUDA tensors).
"""
import collections


class Pipeline(object):
    def __init__(self, net, depth=3):
        self.nets = [net] + [net.clone() for _ in range(max(1, depth) - 1)]
        self._next = 0
        self._pending = collections.deque()

    @property
    def depth(self):
        return len(self.nets)

    def submit(self, in_ptr, n, h, w, prob_ptr=None, loc_ptr=None, next_ptr=None, tag=None):
        """Enqueue one forward (asynchronous).  If every executor is busy, waits for the oldest request
        first.  Returns the executor index."""
        if len(self._pending) >= len(self.nets):
            self.wait_one()
        k = self._next
        self._next = (k + 1) % len(self.nets)
        self.nets[k].forward_device(in_ptr, n, h, w, prob_ptr, loc_ptr, next_ptr, stream="own")
        self._pending.append((k, tag))
        return k

    def wait_one(self):
        """Block until the oldest in-flight request has finished; returns its tag."""
        k, tag = self._pending.popleft()
        self.nets[k].synchronize()
        return tag

    def drain(self):
        tags = []
        while self._pending:
            tags.append(self.wait_one())
        return tags
