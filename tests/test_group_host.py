"""CPU: the host side of dc_group_* (creation rules, error behaviour without a device) — the grouped forward itself is
tests/test_gpu_group.py."""
import numpy as np
import pytest

import caffe
from deepcut_tools import deepercut_prototxt


def _net(h=64, w=64):
    return caffe.Net(deepercut_prototxt(101, h, w), caffe.TEST, from_text=True)


def test_a_group_is_a_net_and_its_clones():
    a = _net()
    g = caffe.NetGroup([a, a.clone(), a.clone()])
    assert len(g) == 3
    with pytest.raises(caffe.DeepcutError):
        caffe.NetGroup([a, _net()])  # another model: nothing to share
    with pytest.raises(caffe.DeepcutError):
        caffe.NetGroup([a, a])       # the same executor twice
    b = a.clone()
    b.set_option(3, 1)               # float16 member beside a float32 one
    with pytest.raises(caffe.DeepcutError):
        caffe.NetGroup([a, b])
    with pytest.raises(caffe.DeepcutError):
        g.plan_text()                # nothing has run


def test_group_forward_needs_the_gpu_path():
    a = _net()
    g = caffe.NetGroup([a, a.clone()])
    caffe.set_mode_cpu()
    x = np.zeros((1, 3, 64, 64), np.float32)
    with pytest.raises(caffe.DeepcutError) as e:
        g.forward_batch([x, x])
    assert "CPU mode" in str(e.value) or "no HIP device" in str(e.value)
    with pytest.raises(ValueError):
        g.forward_batch([x])  # one batch per member


def test_a_group_may_be_destroyed_after_its_nets():
    """A garbage collector finalises a reference cycle in any order (the demo mirror keeps its scale groups ON the net): destroying
    the group after one of its members must not touch the dead net."""
    import ctypes as C

    from caffe import pycaffe as P

    L = P._lib
    text = deepercut_prototxt(101, 64, 64).encode()
    a, b, g = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert L.dc_net_create_from_text(text, None, caffe.TEST, C.byref(a)) == 0
    assert L.dc_net_clone(a, C.byref(b)) == 0
    arr = (C.c_void_p * 2)(a, b)
    assert L.dc_group_create(arr, 2, C.byref(g)) == 0 and L.dc_group_size(g) == 2
    assert L.dc_net_destroy(b) == 0 and L.dc_net_destroy(a) == 0
    assert L.dc_group_destroy(g) == 0
