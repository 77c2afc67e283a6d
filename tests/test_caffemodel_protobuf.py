"""CPU: the C++ .caffemodel reader / writer (csrc/formats.cpp) pinned against an INDEPENDENT encoder — the real protobuf
runtime (google.protobuf, in the image) driven by a descriptor built at test time from the field numbers of the
reference's src/caffe/proto/caffe.proto:6-22 (BlobShape, BlobProto), :64-96 (NetParameter), :311-334 (LayerParameter)
and :1205-1260 (V1LayerParameter).  Until now the reader had only ever been fed by this repository's own writer
(deepcut_tools/caffemodel.py).  Covers packed and unpacked `data`, `double_data`, the legacy num/channels/height/width
shape, unknown fields, and the deprecated V1 `layers` form; and the reverse direction: a file written by dc_net_save is
parsed by protobuf."""
import numpy as np
import pytest

descriptor_pb2 = pytest.importorskip("google.protobuf.descriptor_pb2")
from google.protobuf import descriptor_pool, message_factory  # noqa: E402

import caffe  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _messages():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "caffe_subset.proto"
    fd.package = "caffe_subset"
    fd.syntax = "proto2"

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, number, typ, label=F.LABEL_OPTIONAL, type_name=None, packed=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, number, typ, label
        if type_name:
            f.type_name = ".caffe_subset." + type_name
        if packed is not None:
            f.options.packed = packed
        return f

    R = F.LABEL_REPEATED
    bs = msg("BlobShape")                                          # caffe.proto:6-8
    field(bs, "dim", 1, F.TYPE_INT64, R, packed=True)
    for name, packed in (("BlobProto", True), ("BlobProtoUnpacked", False)):  # caffe.proto:10-22
        bp = msg(name)
        field(bp, "shape", 7, F.TYPE_MESSAGE, type_name="BlobShape")
        field(bp, "data", 5, F.TYPE_FLOAT, R, packed=packed)
        field(bp, "diff", 6, F.TYPE_FLOAT, R, packed=packed)
        field(bp, "double_data", 8, F.TYPE_DOUBLE, R, packed=packed)
        field(bp, "num", 1, F.TYPE_INT32)
        field(bp, "channels", 2, F.TYPE_INT32)
        field(bp, "height", 3, F.TYPE_INT32)
        field(bp, "width", 4, F.TYPE_INT32)
    for name, blob in (("LayerParameter", "BlobProto"), ("LayerParameterUnpacked", "BlobProtoUnpacked")):  # caffe.proto:311-334
        lp = msg(name)
        field(lp, "name", 1, F.TYPE_STRING)
        field(lp, "type", 2, F.TYPE_STRING)
        field(lp, "bottom", 3, F.TYPE_STRING, R)
        field(lp, "top", 4, F.TYPE_STRING, R)
        field(lp, "phase", 10, F.TYPE_INT32)
        field(lp, "loss_weight", 5, F.TYPE_FLOAT, R)
        field(lp, "blobs", 7, F.TYPE_MESSAGE, R, type_name=blob)
    v1 = msg("V1LayerParameter")                                   # caffe.proto:1205-1260
    field(v1, "bottom", 2, F.TYPE_STRING, R)
    field(v1, "top", 3, F.TYPE_STRING, R)
    field(v1, "name", 4, F.TYPE_STRING)
    field(v1, "type", 5, F.TYPE_INT32)                             # enum LayerType on the wire = varint
    field(v1, "blobs", 6, F.TYPE_MESSAGE, R, type_name="BlobProto")
    field(v1, "blobs_lr", 7, F.TYPE_FLOAT, R)
    for name, layer in (("NetParameter", "LayerParameter"), ("NetParameterUnpacked", "LayerParameterUnpacked")):  # caffe.proto:64-96
        np_ = msg(name)
        field(np_, "name", 1, F.TYPE_STRING)
        field(np_, "input", 3, F.TYPE_STRING, R)
        field(np_, "input_shape", 8, F.TYPE_MESSAGE, R, type_name="BlobShape")
        field(np_, "input_dim", 4, F.TYPE_INT32, R)
        field(np_, "force_backward", 5, F.TYPE_BOOL)
        field(np_, "debug_info", 7, F.TYPE_BOOL)
        field(np_, "layer", 100, F.TYPE_MESSAGE, R, type_name=layer)
        field(np_, "layers", 2, F.TYPE_MESSAGE, R, type_name="V1LayerParameter")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    if get is None:
        fac = message_factory.MessageFactory(pool)
        get = fac.GetPrototype
    return {n: get(pool.FindMessageTypeByName("caffe_subset." + n)) for n in
            ("NetParameter", "NetParameterUnpacked", "LayerParameter", "BlobProto", "V1LayerParameter")}


PROTO = '''name: "tiny"
input: "data" input_dim: 1 input_dim: 3 input_dim: 16 input_dim: 16
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 4 kernel_size: 3 pad: 1 } }
layer { name: "bn1" type: "BatchNorm" bottom: "c1" top: "c1" batch_norm_param { use_global_stats: true } }
layer { name: "s1" type: "Scale" bottom: "c1" top: "c1" scale_param { bias_term: true } }
layer { name: "r1" type: "ReLU" bottom: "c1" top: "c1" }
layer { name: "c2" type: "Convolution" bottom: "c1" top: "c2" convolution_param { num_output: 2 kernel_size: 1 bias_term: false } }
'''


def _weights(seed=0):
    rs = np.random.RandomState(seed)
    return {"c1": [rs.randn(4, 3, 3, 3).astype(np.float32), rs.randn(4).astype(np.float32)],
            "bn1": [rs.randn(4).astype(np.float32), (1 + rs.rand(4)).astype(np.float32), np.array([999.5], np.float32)],
            "s1": [rs.randn(4).astype(np.float32), rs.randn(4).astype(np.float32)],
            "c2": [rs.randn(2, 4, 1, 1).astype(np.float32)]}


def _fill_blob(b, arr, style):
    if style == "legacy" and arr.ndim == 4:  # deprecated 4-D dimensions (caffe.proto:17-21)
        b.num, b.channels, b.height, b.width = arr.shape
    elif style == "legacy":
        b.num, b.channels, b.height, b.width = (1,) * (4 - arr.ndim) + arr.shape
    else:
        b.shape.dim.extend(arr.shape)
    if style == "double":
        b.double_data.extend(arr.astype(np.float64).ravel().tolist())
    else:
        b.data.extend(arr.ravel().tolist())


def _check_loaded(net, w):
    for lname, blobs in w.items():
        assert len(net.params[lname]) == len(blobs)
        for p, b in zip(net.params[lname], blobs):
            assert p.data.shape == b.shape
            assert np.array_equal(p.data, b), lname


@pytest.mark.parametrize("style", ["packed", "unpacked", "double", "legacy"])
def test_reader_takes_files_encoded_by_the_protobuf_runtime(tmp_path, style):
    M = _messages()
    w = _weights(1)
    net_msg = M["NetParameterUnpacked" if style == "unpacked" else "NetParameter"]()
    net_msg.name = "tiny"
    net_msg.force_backward = True  # fields the loader has no use for must be skipped, not choked on
    net_msg.input.append("data")
    extra = net_msg.layer.add()     # a source layer the net does not have: ignored (net.cpp:815-818)
    extra.name, extra.type = "not_in_the_net", "InnerProduct"
    _fill_blob(extra.blobs.add(), np.ones((2, 2), np.float32), "packed")
    for lname, blobs in w.items():
        l = net_msg.layer.add()
        l.name, l.type = lname, {"c": "Convolution", "b": "BatchNorm", "s": "Scale"}[lname[0]]
        l.bottom.append("x")
        l.top.append("y")
        l.phase = 1
        l.loss_weight.append(0.5)
        for b in blobs:
            _fill_blob(l.blobs.add(), b, style)
    path = str(tmp_path / ("pb_%s.caffemodel" % style))
    raw = net_msg.SerializeToString()
    open(path, "wb").write(raw)
    if style == "unpacked":  # really the one-tag-per-element encoding: 5 bytes per float instead of 4 + header
        assert len(raw) > len(_roundtrip_packed(M, w)) + 100
    net = caffe.Net(PROTO, path, caffe.TEST, from_text=True)
    _check_loaded(net, w)


def _roundtrip_packed(M, w):
    m = M["NetParameter"]()
    for lname, blobs in w.items():
        l = m.layer.add()
        l.name = lname
        for b in blobs:
            _fill_blob(l.blobs.add(), b, "packed")
    return m.SerializeToString()


def test_v1_layers_encoded_by_protobuf(tmp_path):
    """The deprecated `layers` (field 2, V1LayerParameter: name = 4, type enum = 5, blobs = 6) is upgraded on load
    (upgrade_proto.cpp:19-78)."""
    M = _messages()
    w = _weights(2)
    m = M["NetParameter"]()
    m.name = "tiny_v1"
    v1type = {"c": 4, "b": 0, "s": 0}  # CONVOLUTION = 4; types that did not exist in V1 carry NONE
    for lname, blobs in w.items():
        l = m.layers.add()
        l.name, l.type = lname, v1type[lname[0]]
        l.blobs_lr.extend([1.0, 2.0])
        for b in blobs:
            _fill_blob(l.blobs.add(), b, "legacy")
    path = str(tmp_path / "v1.caffemodel")
    open(path, "wb").write(m.SerializeToString())
    net = caffe.Net(PROTO, path, caffe.TEST, from_text=True)
    _check_loaded(net, w)


def test_files_written_by_dc_net_save_parse_with_protobuf(tmp_path):
    M = _messages()
    w = _weights(3)
    net = caffe.Net(PROTO, caffe.TEST, from_text=True)
    for lname, blobs in w.items():
        for p, b in zip(net.params[lname], blobs):
            p.data[...] = b
    path = str(tmp_path / "saved.caffemodel")
    net.save(path)
    m = M["NetParameter"]()
    m.ParseFromString(open(path, "rb").read())
    assert m.name == "tiny"
    by_name = {l.name: l for l in m.layer}
    assert [l.name for l in m.layer] == ["c1", "bn1", "s1", "r1", "c2"]  # Net::ToProto writes every layer (net.cpp:910-925)
    assert by_name["c1"].type == "Convolution" and list(by_name["c1"].bottom) == ["data"] and list(by_name["c1"].top) == ["c1"]
    for lname, blobs in w.items():
        assert len(by_name[lname].blobs) == len(blobs)
        for pb, b in zip(by_name[lname].blobs, blobs):
            assert list(pb.shape.dim) == list(b.shape)
            assert np.array_equal(np.array(pb.data, np.float32).reshape(b.shape), b)
    # and the Python writer of this repository agrees with protobuf too
    from deepcut_tools import write_caffemodel

    p2 = str(tmp_path / "py.caffemodel")
    write_caffemodel(p2, "tiny", [(k, "X", v) for k, v in w.items()])
    m2 = M["NetParameter"]()
    m2.ParseFromString(open(p2, "rb").read())
    assert [l.name for l in m2.layer] == list(w) and np.array_equal(np.array(m2.layer[0].blobs[0].data, np.float32), w["c1"][0].ravel())


def test_truncated_and_corrupt_files_are_refused(tmp_path):
    M = _messages()
    raw = _roundtrip_packed(M, _weights(4))
    for cut in (len(raw) - 3, len(raw) // 2, 7):
        path = str(tmp_path / ("cut%d.caffemodel" % cut))
        open(path, "wb").write(raw[:cut])
        with pytest.raises(caffe.DeepcutError):
            caffe.Net(PROTO, path, caffe.TEST, from_text=True)
