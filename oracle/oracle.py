"""oracle/oracle.py — layer-by-layer CPU executor of a Caffe prototxt over oracle/liboracle.so.

TEST INFRASTRUCTURE ONLY (see caffe_cpu.c header).  It is deliberately independent of the product:
its own prototxt reader (tiny recursive-descent parser below), its own .caffemodel-free weight
injection (a dict name -> [ndarray]), NCHW float32 blobs in a dict, one reference layer = one call,
executed serially in file order exactly as Net::ForwardFromTo does (src/caffe/net.cpp:565-581) with
in-place tops overwriting their blob (net.cpp:394-400).  No fusion, no layout change.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "caffe_cpu.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        fp, ci, cf, sz = C.POINTER(C.c_float), C.c_int, C.c_float, C.c_size_t
        L.oracle_conv_forward.argtypes = [fp, ci, ci, ci, ci, fp, fp, ci] + [ci] * 8 + [fp]
        L.oracle_deconv_forward.argtypes = [fp, ci, ci, ci, ci, fp, fp, ci] + [ci] * 8 + [fp]
        L.oracle_batchnorm_forward.argtypes = [fp, ci, ci, ci, fp, fp, fp, cf]
        L.oracle_scale_forward.argtypes = [fp, ci, ci, ci, fp, fp]
        L.oracle_relu_forward.argtypes = [fp, sz, cf]
        L.oracle_sigmoid_forward.argtypes = [fp, fp, sz]
        L.oracle_eltwise_sum.argtypes = [fp, fp, fp, sz]
        L.oracle_pool_out.argtypes = [ci] * 4
        L.oracle_pool_out.restype = ci
        L.oracle_maxpool_forward.argtypes = [fp, ci, ci, ci, ci, ci, ci, ci, fp]
        L.oracle_maxpool_forward_rect.argtypes = [fp] + [ci] * 10 + [fp]
        L.oracle_crop_forward.argtypes = [fp] + [ci] * 8 + [fp]
        L.oracle_sgemm.argtypes = [ci, ci, ci, ci, fp, fp, cf, fp]
        L.oracle_im2col.argtypes = [fp] + [ci] * 11 + [fp]
        L.oracle_col2im.argtypes = [fp] + [ci] * 11 + [fp]
        L.oracle_set_threads.argtypes = [ci]
        L.oracle_set_double_acc.argtypes = [ci]
        L.oracle_max_threads.restype = ci
        _lib = L
    return _lib


def set_threads(n):
    lib().oracle_set_threads(int(n))


def set_double_acc(on):
    lib().oracle_set_double_acc(1 if on else 0)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# single-layer entry points (NCHW float32 in / out)
# ---------------------------------------------------------------------------------------------
def _hw(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def conv_forward(x, w, bias=None, stride=1, pad=0, dilation=1):
    """stride / pad / dilation: int or (h, w)."""
    x, w = _f32(x), _f32(w)
    n, c, h, wd = x.shape
    co, ci, kh, kw = w.shape
    assert ci == c
    (sh, sw), (ph, pw), (dh, dw) = _hw(stride), _hw(pad), _hw(dilation)
    oh = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    ow = (wd + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    y = np.empty((n, co, oh, ow), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().oracle_conv_forward(_p(x), n, c, h, wd, _p(w), _p(b), co, kh, kw, ph, pw, sh, sw, dh, dw, _p(y))
    return y


def deconv_forward(x, w, bias=None, stride=1, pad=0, dilation=1):
    x, w = _f32(x), _f32(w)
    n, c, h, wd = x.shape
    ci, co, kh, kw = w.shape
    assert ci == c
    oh = stride * (h - 1) + dilation * (kh - 1) + 1 - 2 * pad
    ow = stride * (wd - 1) + dilation * (kw - 1) + 1 - 2 * pad
    y = np.empty((n, co, oh, ow), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().oracle_deconv_forward(_p(x), n, c, h, wd, _p(w), _p(b), co, kh, kw, pad, pad, stride, stride, dilation,
                                dilation, _p(y))
    return y


def batchnorm_forward(x, mean, var, sf, eps=1e-5):
    y = _f32(x).copy()
    n, c = y.shape[:2]
    s = int(np.prod(y.shape[2:]))
    lib().oracle_batchnorm_forward(_p(y), n, c, s, _p(_f32(mean)), _p(_f32(var)), _p(_f32(np.reshape(sf, (1,)))), eps)
    return y


def scale_forward(x, gamma, beta=None):
    y = _f32(x).copy()
    n, c = y.shape[:2]
    s = int(np.prod(y.shape[2:]))
    lib().oracle_scale_forward(_p(y), n, c, s, _p(_f32(gamma)), _p(_f32(beta)) if beta is not None else None)
    return y


def relu_forward(x, slope=0.0):
    y = _f32(x).copy()
    lib().oracle_relu_forward(_p(y), y.size, slope)
    return y


def sigmoid_forward(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().oracle_sigmoid_forward(_p(x), _p(y), x.size)
    return y


def eltwise_sum(a, b):
    a, b = _f32(a), _f32(b)
    assert a.shape == b.shape
    y = np.empty_like(a)
    lib().oracle_eltwise_sum(_p(a), _p(b), _p(y), a.size)
    return y


def maxpool_forward(x, k, stride, pad=0):
    """k / stride / pad: int or (h, w)."""
    x = _f32(x)
    n, c, h, w = x.shape
    (kh, kw), (sh, sw), (ph, pw) = _hw(k), _hw(stride), _hw(pad)
    oh, ow = lib().oracle_pool_out(h, kh, ph, sh), lib().oracle_pool_out(w, kw, pw, sw)
    y = np.empty((n, c, oh, ow), np.float32)
    if (kh, sh, ph) == (kw, sw, pw):
        lib().oracle_maxpool_forward(_p(x), n, c, h, w, kh, sh, ph, _p(y))
    else:
        lib().oracle_maxpool_forward_rect(_p(x), n, c, h, w, kh, kw, sh, sw, ph, pw, _p(y))
    return y


def crop_forward(x, ref, oh=0, ow=0):
    x = _f32(x)
    n, c, h, w = x.shape
    h1, w1 = ref.shape[2], ref.shape[3]
    assert h - oh > h1 and w - ow > w1, "invalid offset (crop_layer.cpp:30-32)"
    y = np.empty((n, c, h1, w1), np.float32)
    lib().oracle_crop_forward(_p(x), n, c, h, w, oh, ow, h1, w1, _p(y))
    return y


# ---------------------------------------------------------------------------------------------
# prototxt reader (independent of csrc/formats.cpp)
# ---------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"""\s*(?:\#[^\n]*\n)*\s*(?:(?P<brace>[{}])|(?P<str>"(?:[^"\\]|\\.)*"|'(?:[^'\\]|\\.)*')|(?P<tok>[^\s{}:#"']+)|(?P<colon>:))""")


def parse_prototxt(text):
    """-> nested list of (key, value) pairs; value is str or a nested list."""
    toks = []
    pos = 0
    text = text + "\n"
    while True:
        m = _TOKEN.match(text, pos)
        if not m or m.end() == pos:
            break
        pos = m.end()
        if m.group("brace"):
            toks.append(m.group("brace"))
        elif m.group("str") is not None:
            toks.append(("S", m.group("str")[1:-1]))
        elif m.group("tok"):
            toks.append(("T", m.group("tok")))
        elif m.group("colon"):
            toks.append(":")
    if text[pos:].strip() and not text[pos:].strip().startswith("#"):
        raise ValueError("prototxt: cannot tokenise near %r" % text[pos:pos + 40])
    i = [0]

    def msg(depth):
        out = []
        while i[0] < len(toks):
            t = toks[i[0]]
            if t == "}":
                if depth == 0:
                    raise ValueError("unbalanced }")
                i[0] += 1
                return out
            key = t[1]
            i[0] += 1
            if toks[i[0]] == ":":
                i[0] += 1
            if toks[i[0]] == "{":
                i[0] += 1
                out.append((key, msg(depth + 1)))
            else:
                out.append((key, toks[i[0]][1]))
                i[0] += 1
        if depth:
            raise ValueError("missing }")
        return out

    return msg(0)


def _get(m, key, default=None):
    for k, v in m:
        if k == key:
            return v
    return default


def _all(m, key):
    return [v for k, v in m if k == key]


class OracleNet(object):
    """Serial CPU forward of a prototxt: blobs is an OrderedDict-like dict name -> NCHW float32 ndarray."""

    def __init__(self, prototxt_text, weights):
        root = parse_prototxt(prototxt_text)
        self.inputs = _all(root, "input")
        self.layers = _all(root, "layer")
        self.weights = {name: [np.ascontiguousarray(b, np.float32) for b in blobs] for name, _t, blobs in weights}
        self.blobs = {}

    def forward(self, **inputs):
        B = self.blobs
        for k, v in inputs.items():
            B[k] = _f32(v)
        for L in self.layers:
            name, typ = _get(L, "name"), _get(L, "type")
            bot, top = _all(L, "bottom"), _all(L, "top")
            W = self.weights.get(name, [])
            x = B[bot[0]]
            if typ in ("Convolution", "Deconvolution"):
                p = _get(L, "convolution_param")
                k = int(_get(p, "kernel_size"))
                s = int(_get(p, "stride", 1))
                pad = int(_get(p, "pad", 0))
                d = int(_get(p, "dilation", 1))
                bias = W[1] if _get(p, "bias_term", "true") == "true" else None
                f = conv_forward if typ == "Convolution" else deconv_forward
                y = f(x, W[0], bias, s, pad, d)
            elif typ == "BatchNorm":
                bp = _get(L, "batch_norm_param", [])
                assert _get(bp, "use_global_stats", "true") == "true"
                y = batchnorm_forward(x, W[0], W[1], W[2], float(_get(bp, "eps", 1e-5)))
            elif typ == "Scale":
                sp = _get(L, "scale_param", [])
                y = scale_forward(x, W[0], W[1] if _get(sp, "bias_term", "false") == "true" else None)
            elif typ == "ReLU":
                y = relu_forward(x)
            elif typ == "Sigmoid":
                y = sigmoid_forward(x)
            elif typ == "Pooling":
                pp = _get(L, "pooling_param")
                assert _get(pp, "pool", "MAX") == "MAX"
                y = maxpool_forward(x, int(_get(pp, "kernel_size")), int(_get(pp, "stride", 1)), int(_get(pp, "pad", 0)))
            elif typ == "Eltwise":
                y = eltwise_sum(x, B[bot[1]])
            elif typ == "Crop":
                cp = _get(L, "crop_param", [])
                y = crop_forward(x, B[bot[1]], int(_get(cp, "offset_height", 0)), int(_get(cp, "offset_width", 0)))
            else:
                raise NotImplementedError(typ)
            B[top[0]] = y
        return B
