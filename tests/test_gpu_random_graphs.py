"""-m gpu: randomly generated small networks of the path's layer types (seeded, reproducible) through the
lowering at every fusion level against the CPU oracle.  Exercises what the fixed DeeperCut graph cannot:
in-place vs out-of-place chains, BatchNorm/Scale/ReLU that do NOT follow a convolution, blobs with several
consumers (Split insertion), residual adds whose operands come in either order, heads with and without
sigmoid, pooling in the middle, odd sizes, batch > 1.  All Caffe-visible blobs are compared at DC_OPT_FUSE 0,
the net outputs at levels 1 and 2."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


class Gen(object):
    def __init__(self, seed, chans=(32, 64, 96)):
        self.chans = chans
        self.rs = np.random.RandomState(seed)
        self.layers = []
        self.shapes = {}
        self.n = 0

    def name(self, p):
        self.n += 1
        return "%s%d" % (p, self.n)

    def add(self, text):
        self.layers.append(text)

    def conv(self, bot, cout, k, s, d, bias, typ="Convolution", top=None):
        top = top or self.name("c")
        p = d * (k // 2) if typ == "Convolution" else 0
        self.add('layer { name: "%s" type: "%s" bottom: "%s" top: "%s" convolution_param { num_output: %d kernel_size: %d '
                 "stride: %d pad: %d dilation: %d bias_term: %s } }" % (top, typ, bot, top, cout, k, s, p, d, "true" if bias else "false"))
        n, c, h, w = self.shapes[bot]
        if typ == "Convolution":
            oh = (h + 2 * p - (d * (k - 1) + 1)) // s + 1
            ow = (w + 2 * p - (d * (k - 1) + 1)) // s + 1
        else:
            oh, ow = s * (h - 1) + k, s * (w - 1) + k
        self.shapes[top] = (n, cout, oh, ow)
        return top

    def unary(self, typ, bot, inplace, extra=""):
        top = bot if inplace else self.name(typ[0].lower())
        self.add('layer { name: "%s" type: "%s" bottom: "%s" top: "%s" %s }' % (self.name("u"), typ, bot, top, extra))
        self.shapes[top] = self.shapes[bot]
        return top

    def chain(self, blob):
        """random BatchNorm / Scale / ReLU tail, in place or not"""
        rs = self.rs
        if rs.rand() < 0.7:
            blob = self.unary("BatchNorm", blob, rs.rand() < 0.8, "batch_norm_param { use_global_stats: true }")
        if rs.rand() < 0.7:
            blob = self.unary("Scale", blob, rs.rand() < 0.8, "scale_param { bias_term: %s }" % ("true" if rs.rand() < 0.8 else "false"))
        if rs.rand() < 0.7:
            blob = self.unary("ReLU", blob, rs.rand() < 0.8)
        return blob

    def build(self):
        rs = self.rs
        n = int(rs.choice([1, 2]))
        c0 = int(rs.choice(self.chans[:2]))
        h, w = int(rs.randint(9, 22)), int(rs.randint(9, 22))
        self.shapes["x"] = (n, c0, h, w)
        cur = "x"
        if rs.rand() < 0.3:  # element-wise layers straight on the input (not behind a convolution)
            cur = self.chain(cur)
        hi = None
        for _ in range(int(rs.randint(2, 5))):
            kind = rs.choice(["conv", "res", "pool", "branch"], p=[0.4, 0.3, 0.1, 0.2])
            c = self.shapes[cur][1]
            if kind == "conv":
                k = int(rs.choice([1, 3]))
                s = int(rs.choice([1, 1, 2]))
                d = int(rs.choice([1, 2])) if k == 3 else 1
                cur = self.chain(self.conv(cur, int(rs.choice(self.chans)), k, s, d, rs.rand() < 0.3))
            elif kind == "res":
                a = self.chain(self.conv(cur, c, int(rs.choice([1, 3])), 1, 1, False))
                b = self.conv(a, c, 1, 1, 1, rs.rand() < 0.3)
                if rs.rand() < 0.7:
                    b = self.unary("BatchNorm", b, True, "batch_norm_param { use_global_stats: true }")
                ops = [cur, b] if rs.rand() < 0.6 else [b, cur]
                top = self.name("e")
                self.add('layer { name: "%s" type: "Eltwise" bottom: "%s" bottom: "%s" top: "%s" }' % (top, ops[0], ops[1], top))
                self.shapes[top] = self.shapes[cur]
                cur = top
                if rs.rand() < 0.7:
                    cur = self.unary("ReLU", cur, rs.rand() < 0.8)
            elif kind == "pool" and min(self.shapes[cur][2:]) >= 5:
                top = self.name("p")
                self.add('layer { name: "%s" type: "Pooling" bottom: "%s" top: "%s" pooling_param { pool: MAX kernel_size: 3 stride: 2 } }' % (top, cur, top))
                nn, cc, hh, ww = self.shapes[cur]
                self.shapes[top] = (nn, cc, int(np.ceil((hh - 3) / 2.0)) + 1, int(np.ceil((ww - 3) / 2.0)) + 1)
                cur = top
            else:  # a side branch that stays a net output: `cur` gets two consumers
                side = self.conv(cur, self.chans[0], 1, 1, 1, True)
                if rs.rand() < 0.5:
                    self.unary("Sigmoid", side, rs.rand() < 0.5)
            if hi is None and min(self.shapes[cur][2:]) >= 4 and rs.rand() < 0.5:
                hi = cur
        if hi is not None and self.shapes[hi][1] % 32 == 0:  # DeeperCut-style heads: deconv(low) cropped onto conv1x1(hi)
            low = self.chain(self.conv(hi, int(rs.choice([64, 128])), 1, 2, 1, False))
            for hd in range(int(rs.randint(1, 4))):
                co = int(rs.choice([5, 14, 28, 33]))
                up = self.conv(low, co, 3, 2, 1, True, typ="Deconvolution")
                skip = self.conv(hi, co, 1, 1, 1, True)
                cr = self.name("k")
                self.add('layer { name: "%s" type: "Crop" bottom: "%s" bottom: "%s" top: "%s" }' % (cr, up, skip, cr))
                self.shapes[cr] = self.shapes[skip]
                top = self.name("h")
                ops = [skip, cr] if rs.rand() < 0.7 else [cr, skip]
                self.add('layer { name: "%s" type: "Eltwise" bottom: "%s" bottom: "%s" top: "%s" }' % (top, ops[0], ops[1], top))
                self.shapes[top] = self.shapes[skip]
                if hd == 0 and rs.rand() < 0.6:
                    self.unary("Sigmoid", top, False)
        n, c, h, w = self.shapes["x"]
        return 'input: "x" input_dim: %d input_dim: %d input_dim: %d input_dim: %d\n' % (n, c, h, w) + "\n".join(self.layers)


def _fill(net, rs):
    weights = []
    for name in net.params:
        blobs = []
        typ = net.layer_types[net._layer_names.index(name)]
        for i, p in enumerate(net.params[name]):
            if typ == "BatchNorm":
                val = [rs.randn(*p.shape) * 0.2, 0.5 + rs.rand(*p.shape), np.array([rs.choice([1.0, 3.0])])][i]
            elif typ == "Scale":
                val = [1 + 0.2 * rs.randn(*p.shape), 0.2 * rs.randn(*p.shape)][i]
            elif len(p.shape) == 4:
                val = rs.randn(*p.shape) / np.sqrt(np.prod(p.shape[1:]) if typ == "Convolution" else p.shape[0] * 2.25)
            else:
                val = 0.3 * rs.randn(*p.shape)
            val = np.asarray(val, np.float32)
            p.data[...] = val
            blobs.append(val)
        weights.append((name, typ, blobs))
    return weights


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DC_FUZZ_SEEDS", "24"))))
def test_random_graph(gpu_caffe, seed):
    text = Gen(seed).build()
    rs = np.random.RandomState(1000 + seed)
    ref = None
    for fuse in (0, 1, 2):
        net = gpu_caffe.Net(text, gpu_caffe.TEST, from_text=True, fuse=fuse)
        weights = _fill(net, np.random.RandomState(2000 + seed))
        if ref is None:
            x = rs.randn(*net.blobs["x"].shape).astype(np.float32)
            ref = O.OracleNet(text, weights).forward(x=x)
        net.blobs["x"].data[...] = x
        out = net.forward()
        assert sorted(out) == sorted(net.outputs)
        names = list(ref) if fuse == 0 else list(out)
        for name in names:
            r = ref[name]
            got = net.blobs[name].data
            assert got.shape == r.shape, (seed, fuse, name)
            err = float(np.abs(got - r).max()) / max(1.0, float(np.abs(r).max()))
            assert err <= 1e-4, (seed, fuse, name, err, text)


@pytest.mark.parametrize("seed", range(8))
def test_random_graph_fp16(gpu_caffe, seed):
    """Same generator with channel counts that are multiples of 64 (one 128-byte line of halves), float16 device
    images; tolerance 1e-2 of each blob's range (a handful of layers, each rounding its output to 11 bits)."""
    text = Gen(500 + seed, chans=(64, 128, 192)).build()
    rs = np.random.RandomState(3000 + seed)
    ref = None
    for fuse in (0, 2):
        net = gpu_caffe.Net(text, gpu_caffe.TEST, from_text=True, fuse=fuse, dtype="f16")
        weights = _fill(net, np.random.RandomState(4000 + seed))
        if ref is None:
            x = rs.randn(*net.blobs["x"].shape).astype(np.float32)
            ref = O.OracleNet(text, weights).forward(x=x)
        net.blobs["x"].data[...] = x
        out = net.forward()
        for name in (list(ref) if fuse == 0 else list(out)):
            r = ref[name]
            err = float(np.abs(net.blobs[name].data - r).max()) / max(1.0, float(np.abs(r).max()))
            assert err <= 1e-2, (seed, fuse, name, err)
