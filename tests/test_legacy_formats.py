"""Deprecated weight / definition formats: the reference upgrades V0 / V1 files on load
(src/caffe/util/upgrade_proto.cpp:19-78, UpgradeV1Net :647-850, type names :852-940; field numbers caffe.proto:95,
1205-1296, 1299-1341).  The files below are assembled byte by byte from those field numbers."""
import numpy as np
import pytest

import caffe
from deepcut_tools.caffemodel import _ld, _varint

NEW = '''name: "n" input: "data" input_dim: 1 input_dim: 4 input_dim: 9 input_dim: 9
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 32 kernel_size: 3 pad: 1 } }
layer { name: "r1" type: "ReLU" bottom: "c1" top: "c1" }
layer { name: "up" type: "Deconvolution" bottom: "c1" top: "up" convolution_param { num_output: 2 kernel_size: 3 stride: 2 } }
layer { name: "p" type: "Sigmoid" bottom: "up" top: "p" }
'''
V1 = '''name: "n" input: "data" input_dim: 1 input_dim: 4 input_dim: 9 input_dim: 9
layers { name: "c1" type: CONVOLUTION bottom: "data" top: "c1" blobs_lr: 1 blobs_lr: 2 weight_decay: 1 weight_decay: 0
         convolution_param { num_output: 32 kernel_size: 3 pad: 1 } }
layers { name: "r1" type: RELU bottom: "c1" top: "c1" }
layers { name: "drop" type: DROPOUT bottom: "c1" top: "c1" include { phase: TRAIN } }
layers { name: "up" type: DECONVOLUTION bottom: "c1" top: "up" convolution_param { num_output: 2 kernel_size: 3 stride: 2 } }
layers { name: "p" type: SIGMOID bottom: "up" top: "p" }
'''


def _blob_legacy(arr):
    """BlobProto with the legacy num/channels/height/width fields (1-4) and packed float data (5)."""
    a = np.ascontiguousarray(arr, "<f4")
    dims = (1,) * (4 - a.ndim) + a.shape
    out = b"".join(_varint((i + 1) << 3) + _varint(int(d)) for i, d in enumerate(dims))
    return out + _ld(5, a.tobytes())


def _v1_layer(name, type_enum, blobs, v0=False):
    if v0:  # V1 entry wrapping a V0LayerParameter: name 1, type 2 (string), blobs 50
        inner = _ld(1, name.encode()) + _ld(2, b"conv") + b"".join(_ld(50, _blob_legacy(b)) for b in blobs)
        return _ld(1, inner) + _ld(2, b"data")
    return (_ld(2, b"data") + _ld(3, name.encode()) + _ld(4, name.encode()) + _varint(5 << 3) + _varint(type_enum)
            + b"".join(_ld(6, _blob_legacy(b)) for b in blobs))


def _weights():
    rs = np.random.RandomState(0)
    return {"c1": [rs.randn(32, 4, 3, 3).astype(np.float32), rs.randn(32).astype(np.float32)],
            "up": [rs.randn(32, 2, 3, 3).astype(np.float32), rs.randn(2).astype(np.float32)]}


@pytest.mark.parametrize("v0", [False, True])
def test_v1_and_v0_caffemodels_load_by_name(tmp_path, v0):
    w = _weights()
    body = _ld(1, b"legacy")
    body += _ld(2, _v1_layer("c1", 4, w["c1"], v0))           # CONVOLUTION = 4
    body += _ld(2, _v1_layer("ignored_relu", 18, [], False))  # RELU = 18, no blobs
    body += _ld(2, _v1_layer("up", 39, w["up"], v0))          # DECONVOLUTION = 39
    body += _ld(2, _v1_layer("not_in_net", 14, [np.ones((1, 1, 3, 5), np.float32)], False))
    path = tmp_path / "legacy.caffemodel"
    path.write_bytes(body)
    net = caffe.Net(NEW, str(path), caffe.TEST, from_text=True)
    for name, blobs in w.items():
        for got, want in zip(net.params[name], blobs):
            # legacy 4-D blob shapes (1,1,1,32) are accepted for the 1-D bias (Blob::ShapeEquals, blob.cpp:413-434)
            assert np.array_equal(got.data.reshape(-1), want.reshape(-1)), name


def test_v1_layers_win_over_layer_entries_in_the_same_file(tmp_path):
    w = _weights()
    new_style = _ld(100, _ld(1, b"c1") + _ld(2, b"Convolution") + _ld(7, _blob_legacy(np.zeros((32, 4, 3, 3), np.float32)))
                    + _ld(7, _blob_legacy(np.zeros(32, np.float32))))
    path = tmp_path / "both.caffemodel"
    path.write_bytes(_ld(1, b"x") + new_style + _ld(2, _v1_layer("c1", 4, w["c1"])))
    net = caffe.Net(NEW, str(path), caffe.TEST, from_text=True)
    assert np.array_equal(net.params["c1"][0].data, w["c1"][0])


def test_v1_blob_shape_mismatch_is_an_error(tmp_path):
    path = tmp_path / "bad.caffemodel"
    path.write_bytes(_ld(2, _v1_layer("c1", 4, [np.zeros((32, 4, 3, 2), np.float32), np.zeros(32, np.float32)])))
    with pytest.raises(caffe.DeepcutError) as e:
        caffe.Net(NEW, str(path), caffe.TEST, from_text=True)
    assert "shape" in str(e.value).lower()


def test_v1_prototxt_is_upgraded_on_load():
    new, old = caffe.Net(NEW, caffe.TEST, from_text=True), caffe.Net(V1, caffe.TEST, from_text=True)
    assert old._layer_names == new._layer_names == ["c1", "r1", "up", "p"]
    assert [l.type for l in old.layers] == [l.type for l in new.layers] == ["Convolution", "ReLU", "Deconvolution", "Sigmoid"]
    assert list(old.blobs) == list(new.blobs) and old.outputs == new.outputs == ["p"]
    for k in new.blobs:
        assert old.blobs[k].shape == new.blobs[k].shape
    assert [p.shape for p in old.params["up"]] == [(32, 2, 3, 3), (2,)]
    assert old.plan_text().split("\n", 1)[1] == new.plan_text().split("\n", 1)[1]


def test_v1_prototxt_errors():
    head = 'input: "data" input_dim: 1 input_dim: 4 input_dim: 9 input_dim: 9\n'
    with pytest.raises(caffe.DeepcutError) as e:
        caffe.Net(head + 'layers { name: "x" type: FANCY bottom: "data" top: "x" }', caffe.TEST, from_text=True)
    assert e.value.code == -4 and "FANCY" in str(e.value)
    with pytest.raises(caffe.DeepcutError):
        caffe.Net(head + 'layers { name: "a" type: RELU bottom: "data" top: "a" } '
                         'layer { name: "b" type: "ReLU" bottom: "a" top: "b" }', caffe.TEST, from_text=True)
    with pytest.raises(caffe.DeepcutError) as e:
        caffe.Net(head + 'layers { layer { name: "a" type: "relu" } bottom: "data" top: "a" }', caffe.TEST, from_text=True)
    assert e.value.code == -4
    with pytest.raises(caffe.DeepcutError) as e:  # known V1 type, but not a layer of this forward path
        caffe.Net(head + 'layers { name: "l" type: LRN bottom: "data" top: "o" }', caffe.TEST, from_text=True)
    assert "outside the DeeperCut forward path" in str(e.value)
