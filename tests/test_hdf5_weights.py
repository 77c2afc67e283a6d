"""HDF5 weight files (reference Net::CopyTrainedLayersFromHDF5 net.cpp:861-909, chosen for names ending
in ".h5", net.cpp:843-850; written by Net::ToHDF5 net.cpp:926-975 through H5LTmake_dataset_float).

csrc/hdf5_reader.cpp decodes the file format itself (no HDF5 library in the product).  It is pinned against files made
by the REAL library: the committed tests/golden/weights_golden.h5 (tests/golden/make_hdf5_golden.py) and, when
libhdf5 can be loaded (it can in this image), larger files written on the fly — hundreds of groups, so that the group
B-trees have several levels and many symbol-table nodes."""
import os
import sys

import numpy as np
import pytest

import caffe

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_hdf5_golden as H  # noqa: E402

NET = '''name: "n" input: "data" input_dim: 1 input_dim: 4 input_dim: 9 input_dim: 9
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 32 kernel_size: 3 pad: 1 } }
layer { name: "bn_c1" type: "BatchNorm" bottom: "c1" top: "c1" batch_norm_param { use_global_stats: true } }
layer { name: "scale_c1" type: "Scale" bottom: "c1" top: "c1" scale_param { bias_term: true } }
layer { name: "up" type: "Deconvolution" bottom: "c1" top: "up" convolution_param { num_output: 2 kernel_size: 3 stride: 2 } }
'''
needs_lib = pytest.mark.skipif(H.libs() is None, reason="libhdf5 not loadable: only the committed fixture is checked")


def test_committed_fixture_loads_by_name():
    net = caffe.Net(NET, os.path.join(HERE, "golden", "weights_golden.h5"), caffe.TEST, from_text=True)
    seen = 0
    for name, blobs in H.golden_layers():
        if name not in net.params:
            assert name == "not_in_the_net"  # "Ignoring source layer"
            continue
        for p, b in zip(net.params[name], blobs):
            want = b.astype(np.float64).astype(np.float32) if name == "scale_c1" else b  # stored as double in the fixture
            assert np.array_equal(p.data, want), name
            seen += 1
    assert seen == 9


def test_suffix_selects_the_reader(tmp_path):
    # the reference decides by the file NAME (net.cpp:843-850): an HDF5 file not called *.h5 goes to the protobuf reader
    import shutil

    other = tmp_path / "weights.caffemodel"
    shutil.copy(os.path.join(HERE, "golden", "weights_golden.h5"), str(other))
    net = caffe.Net(NET, caffe.TEST, from_text=True)
    with pytest.raises(caffe.DeepcutError):
        net.copy_from(str(other))
    bogus = tmp_path / "bogus.h5"
    bogus.write_bytes(b"\x0a\x03abc" * 100)
    with pytest.raises(caffe.DeepcutError) as e:
        net.copy_from(str(bogus))
    assert "not an HDF5 file" in str(e.value)
    with pytest.raises(caffe.DeepcutError) as e:
        net.copy_from(str(tmp_path / "missing.h5"))
    assert "Could not open file" in str(e.value)


def test_truncated_file_is_an_error_not_a_crash(tmp_path):
    raw = open(os.path.join(HERE, "golden", "weights_golden.h5"), "rb").read()
    net = caffe.Net(NET, caffe.TEST, from_text=True)
    for cut in (40, 700, 2000, len(raw) // 2, len(raw) - 64):
        p = tmp_path / ("cut%d.h5" % cut)
        p.write_bytes(raw[:cut])
        with pytest.raises(caffe.DeepcutError):
            net.copy_from(str(p))


def _chain(n, c=32):
    txt = ['name: "chain" input: "data" input_dim: 1 input_dim: %d input_dim: 4 input_dim: 4' % c]
    prev = "data"
    for i in range(n):
        name = "layer_with_a_rather_long_name_%04d/branch2b" % i
        txt.append('layer { name: "%s" type: "Convolution" bottom: "%s" top: "t%d" convolution_param { num_output: %d kernel_size: 1 } }'
                   % (name, prev, i, c))
        prev = "t%d" % i
    return "\n".join(txt) + "\n"


@needs_lib
@pytest.mark.parametrize("n", [1, 9, 70, 600])
def test_many_layers_written_by_the_real_library(tmp_path, n):
    rs = np.random.RandomState(n)
    names = ["layer_with_a_rather_long_name_%04d/branch2b" % i for i in range(n)]
    # '/' in a Caffe layer name would nest HDF5 groups; the reference has the same limitation, so the file uses '_'
    file_names = [s.replace("/", "_") for s in names]
    layers = [(fn, [rs.randn(32, 32, 1, 1).astype(np.float32), rs.randn(32).astype(np.float32)]) for fn in file_names]
    path = str(tmp_path / "chain.h5")
    H.write_caffe_h5(path, layers)
    net = caffe.Net(_chain(n).replace("/branch2b", "_branch2b"), path, caffe.TEST, from_text=True)
    for fn, blobs in layers:
        assert np.array_equal(net.params[fn][0].data, blobs[0]) and np.array_equal(net.params[fn][1].data, blobs[1]), fn


@needs_lib
def test_hdf5_and_binaryproto_give_the_same_net(tmp_path):
    from deepcut_tools import write_caffemodel

    layers = [(n, b) for n, b in H.golden_layers() if n != "not_in_the_net"]
    H.write_caffe_h5(str(tmp_path / "w.h5"), layers)
    types = {"c1": "Convolution", "bn_c1": "BatchNorm", "scale_c1": "Scale", "up": "Deconvolution"}
    write_caffemodel(str(tmp_path / "w.caffemodel"), "n", [(n, types[n], b) for n, b in layers])
    a = caffe.Net(NET, str(tmp_path / "w.h5"), caffe.TEST, from_text=True)
    b = caffe.Net(NET, str(tmp_path / "w.caffemodel"), caffe.TEST, from_text=True)
    for name in a.params:
        for pa, pb in zip(a.params[name], b.params[name]):
            assert np.array_equal(pa.data, pb.data)


@needs_lib
def test_shape_and_count_mismatches(tmp_path):
    rs = np.random.RandomState(0)
    net = caffe.Net(NET, caffe.TEST, from_text=True)
    H.write_caffe_h5(str(tmp_path / "shape.h5"), [("c1", [rs.randn(32, 4, 3, 2).astype(np.float32), rs.randn(32).astype(np.float32)])])
    with pytest.raises(caffe.DeepcutError) as e:
        net.copy_from(str(tmp_path / "shape.h5"))
    assert "shape mismatch" in str(e.value)
    H.write_caffe_h5(str(tmp_path / "count.h5"), [("c1", [rs.randn(32, 4, 3, 3).astype(np.float32)])])
    with pytest.raises(caffe.DeepcutError) as e:
        net.copy_from(str(tmp_path / "count.h5"))
    assert "Incompatible number of blobs" in str(e.value)


def test_corrupted_bytes_never_crash_the_reader(tmp_path):
    """Every structure offset read from the file is bounds-checked: flipping bytes gives an error or (when the byte was
    padding or payload) a load, never a fault."""
    raw = bytearray(open(os.path.join(HERE, "golden", "weights_golden.h5"), "rb").read())
    rs = np.random.RandomState(7)
    net = caffe.Net(NET, caffe.TEST, from_text=True)
    outcomes = {"ok": 0, "refused": 0}
    p = tmp_path / "fuzz.h5"
    for trial in range(120):
        b = bytearray(raw)
        for _ in range(int(rs.randint(1, 6))):
            # bias towards the metadata at the front of the file, where the structures live
            pos = int(rs.randint(0, 4096)) if trial % 3 else int(rs.randint(0, len(b)))
            b[pos] = int(rs.randint(0, 256))
        p.write_bytes(bytes(b))
        try:
            net.copy_from(str(p))
            outcomes["ok"] += 1
        except caffe.DeepcutError:
            outcomes["refused"] += 1
    assert outcomes["refused"] > 10 and outcomes["ok"] > 10, outcomes
