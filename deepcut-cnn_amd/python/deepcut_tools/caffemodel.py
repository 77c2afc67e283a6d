"""Minimal protobuf wire-format writer/reader for .caffemodel files (NetParameter, caffe.proto:64-96;
LayerParameter :311-334; BlobProto :10-22).  Pure numpy; no protobuf dependency.  The C++ library has
its own independent reader/writer (csrc/formats.cpp); tests cross-check the two."""
import struct

import numpy as np


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _blob(arr):
    arr = np.ascontiguousarray(arr, dtype="<f4")
    dims = b"".join(_varint(int(d)) for d in arr.shape)
    shape = _ld(7, _ld(1, dims))
    data = _varint((5 << 3) | 2) + _varint(arr.size * 4)
    return shape, data, arr


def write_caffemodel(path, net_name, layers):
    """layers: iterable of (name, type, [ndarray, ...])."""
    with open(path, "wb") as f:
        f.write(_ld(1, net_name.encode()))
        for name, typ, blobs in layers:
            parts = [_ld(1, name.encode()), _ld(2, typ.encode())]
            for b in blobs:
                shape, dhdr, arr = _blob(b)
                payload_len = len(shape) + len(dhdr) + arr.size * 4
                parts.append(_varint((7 << 3) | 2) + _varint(payload_len) + shape + dhdr)
                parts.append(arr)
            body_len = sum(p.nbytes if isinstance(p, np.ndarray) else len(p) for p in parts)
            f.write(_varint((100 << 3) | 2) + _varint(body_len))
            for p in parts:
                f.write(p.tobytes() if isinstance(p, np.ndarray) else p)


def _read_varint(buf, p):
    v = 0
    sh = 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << sh
        if not b & 0x80:
            return v, p
        sh += 7


def _fields(buf, p, end):
    while p < end:
        tag, p = _read_varint(buf, p)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = _read_varint(buf, p)
            yield fn, wt, v
        elif wt == 2:
            n, p = _read_varint(buf, p)
            yield fn, wt, (p, p + n)
            p += n
        elif wt == 5:
            yield fn, wt, struct.unpack_from("<f", buf, p)[0]
            p += 4
        elif wt == 1:
            yield fn, wt, struct.unpack_from("<d", buf, p)[0]
            p += 8
        else:
            raise ValueError("unsupported wire type %d" % wt)


def read_caffemodel(path):
    """-> (net_name, [(name, type, [ndarray,...]), ...])"""
    buf = memoryview(open(path, "rb").read())
    name = ""
    layers = []
    for fn, wt, v in _fields(buf, 0, len(buf)):
        if fn == 1 and wt == 2:
            name = bytes(buf[v[0]:v[1]]).decode()
        elif fn == 100 and wt == 2:
            lname = ltype = ""
            blobs = []
            for f2, w2, v2 in _fields(buf, v[0], v[1]):
                if f2 == 1 and w2 == 2:
                    lname = bytes(buf[v2[0]:v2[1]]).decode()
                elif f2 == 2 and w2 == 2:
                    ltype = bytes(buf[v2[0]:v2[1]]).decode()
                elif f2 == 7 and w2 == 2:
                    shape, legacy, data = [], [0, 0, 0, 0], None
                    has_legacy = False
                    for f3, w3, v3 in _fields(buf, v2[0], v2[1]):
                        if f3 == 7 and w3 == 2:
                            for f4, w4, v4 in _fields(buf, v3[0], v3[1]):
                                if f4 == 1 and w4 == 2:
                                    q = v4[0]
                                    while q < v4[1]:
                                        d, q = _read_varint(buf, q)
                                        shape.append(d)
                                elif f4 == 1 and w4 == 0:
                                    shape.append(v4)
                        elif f3 == 5 and w3 == 2:
                            data = np.frombuffer(buf[v3[0]:v3[1]], dtype="<f4").copy()
                        elif f3 == 8 and w3 == 2:
                            data = np.frombuffer(buf[v3[0]:v3[1]], dtype="<f8").astype(np.float32)
                        elif 1 <= f3 <= 4 and w3 == 0:
                            legacy[f3 - 1] = v3
                            has_legacy = True
                    if not shape and has_legacy:
                        shape = legacy
                    if data is None:
                        data = np.zeros(int(np.prod(shape)) if shape else 0, np.float32)
                    blobs.append(data.reshape(shape))
            layers.append((lname, ltype, blobs))
    return name, layers
