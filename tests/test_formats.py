"""CPU: the two independent prototxt readers (csrc/formats.cpp used by the product, oracle/oracle.py used
by the checker) agree on the model; the generated model equals the reference file when it is present."""
import os

import pytest

import caffe
from deepcut_tools import deepercut_layer_table, deepercut_prototxt
from oracle import oracle as O

REF = "/root/reference/models/deepercut/ResNet-152.prototxt"


def _layers_from_oracle_parser(text):
    root = O.parse_prototxt(text)
    out = []
    for L in O._all(root, "layer"):
        out.append((O._get(L, "name"), O._get(L, "type"), O._all(L, "bottom"), O._all(L, "top")))
    return out


def test_both_parsers_agree_on_the_generated_model():
    text = deepercut_prototxt(152, 240, 320)
    a = _layers_from_oracle_parser(text)
    net = caffe.Net(text, caffe.TEST, from_text=True)
    names = [n for n, t in zip(net._layer_names, net.layer_types) if t != "Split"]
    types = [t for t in net.layer_types if t != "Split"]
    assert names == [x[0] for x in a]
    assert types == [x[1] for x in a]
    table = deepercut_layer_table(152)
    assert [(l["name"], l["type"], l["bottoms"], l["tops"]) for l in table] == a


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
def test_generated_model_is_equivalent_to_the_reference_prototxt():
    ref_text = open(REF).read()
    ref = O.parse_prototxt(ref_text)
    gen = O.parse_prototxt(deepercut_prototxt(152, 688, 688))
    assert O._get(ref, "name") == O._get(gen, "name")
    assert O._all(ref, "input") == O._all(gen, "input") and O._all(ref, "input_dim") == O._all(gen, "input_dim")
    rl, gl = O._all(ref, "layer"), O._all(gen, "layer")
    assert len(rl) == len(gl) == 680

    def norm(L):
        d = {}
        for k, v in L:
            if k == "param":
                continue
            if isinstance(v, list):
                v = tuple(sorted((kk, str(vv)) for kk, vv in v if not (kk == "pool" and vv == "MAX")))
            d.setdefault(k, []).append(v)
        return {k: (sorted(v) if k not in ("bottom", "top") else v) for k, v in d.items()}

    for a, b in zip(rl, gl):
        assert norm(a) == norm(b), (O._get(a, "name"), norm(a), norm(b))
    # and the product's parser loads the reference file itself, with the same graph
    n_ref = caffe.Net(REF, caffe.TEST)
    n_gen = caffe.Net(deepercut_prototxt(152, 688, 688), caffe.TEST, from_text=True)
    assert n_ref._layer_names == n_gen._layer_names
    assert list(n_ref.blobs) == list(n_gen.blobs)
    assert [n_ref.blobs[k].shape for k in n_ref.blobs] == [n_gen.blobs[k].shape for k in n_gen.blobs]
    assert abs(n_ref.flops() / 1e9 - 285.02) < 0.01
