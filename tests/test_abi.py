"""CPU: the C-ABI library loads without a GPU and exports every symbol include/deepcut_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "deepcut_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dc_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    import caffe

    lib = ctypes.CDLL(caffe.lib_path())
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_the_header():
    import caffe.pycaffe as pc

    assert sorted(pc.EXPORTED_SYMBOLS) == _declared()


def test_no_torch_or_cxx_types_in_signatures():
    text = open(os.path.join(ROOT, "include", "deepcut_hip.h")).read()
    assert "std::" not in text and "torch" not in text and "at::" not in text
    assert 'extern "C"' in text


def test_context_is_per_thread_and_defaults_to_cpu():
    import threading

    import caffe
    import caffe.pycaffe as pc

    caffe.set_mode_gpu()
    seen = []
    t = threading.Thread(target=lambda: seen.append(pc._lib.dc_get_mode()))
    t.start()
    t.join()
    assert seen == [0]  # Caffe::Get() is thread local, default CPU (common.cpp:13-20,55)
    assert pc._lib.dc_get_mode() == 1
    caffe.set_mode_cpu()


def test_library_has_no_cpu_forward_and_does_not_touch_the_oracle():
    import subprocess

    import caffe

    out = subprocess.run(["nm", "-D", caffe.lib_path()], capture_output=True, text=True).stdout
    assert "oracle_" not in out
    # the product sources never include, link or call anything of oracle/: the word may only appear in comments and in
    # error strings that tell the user where the CPU restatement lives
    for src in os.listdir(os.path.join(ROOT, "deepcut-cnn_amd", "csrc")):
        for ln in open(os.path.join(ROOT, "deepcut-cnn_amd", "csrc", src)).read().splitlines():
            code = ln.split("//")[0]
            if "oracle" in code.lower():
                assert "#include" not in code and "oracle_" not in code and '"' in code, (src, ln)
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "deepcut-cnn_amd", "python")):
        for f in files:
            if f.endswith(".py"):
                body = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in body and "from oracle" not in body, f


def test_new_entry_points_validate_their_arguments_without_a_gpu():
    """Round-2 entries: argument checks and host-side behaviour that need no device."""
    import ctypes as C

    import caffe
    import caffe.pycaffe as pc
    from deepcut_tools import deepercut_prototxt

    L = pc._lib
    net = caffe.Net(deepercut_prototxt(152, 64, 64), caffe.TEST, from_text=True)
    st = net.stats()
    assert set(st) == set(caffe.Net.STAT_NAMES) and all(v == 0 for k, v in st.items())
    assert net.device == -1 and net.dtype == "f32" and net.get_option(1) == 2
    with __import__("pytest").raises(caffe.DeepcutError):
        net.get_option(99)
    # stand-alone blobs: create / shape / host side / copy / destroy; a net's blob cannot be destroyed through this door
    dims = (C.c_int * 4)(2, 3, 4, 5)
    a, b = C.c_void_p(), C.c_void_p()
    assert L.dc_blob_create(4, dims, C.byref(a)) == 0 and L.dc_blob_create(4, dims, C.byref(b)) == 0
    pa = C.POINTER(C.c_float)()
    assert L.dc_blob_mutable_cpu_data(a, C.byref(pa)) == 0
    pa[7] = 2.5
    assert L.dc_blob_copy_from(b, a, 0) == 0 and L.dc_blob_head(b) == 1
    pb = C.POINTER(C.c_float)()
    assert L.dc_blob_cpu_data(b, C.byref(pb)) == 0 and pb[7] == 2.5 and pb[8] == 0.0
    small = C.c_void_p()
    d2 = (C.c_int * 1)(3)
    assert L.dc_blob_create(1, d2, C.byref(small)) == 0
    assert L.dc_blob_copy_from(small, a, 0) != 0 and b"different sizes" in L.dc_last_error()
    assert L.dc_blob_copy_from(small, a, 1) == 0 and L.dc_blob_count(small) == 120  # reshape=1 follows the source
    nb = C.c_void_p()
    assert L.dc_net_blob(net._h, b"data", C.byref(nb)) == 0
    assert L.dc_blob_destroy(nb) != 0 and b"belongs to a net" in L.dc_last_error()
    for h in (a, b, small):
        assert L.dc_blob_destroy(h) == 0
    # layer nets: bottoms become inputs; wrong bottom count / unknown type are refused with the layer named
    x = C.c_void_p()
    dx = (C.c_int * 4)(1, 8, 6, 7)
    assert L.dc_blob_create(4, dx, C.byref(x)) == 0
    ln = C.c_void_p()
    text = b'name: "p" type: "Pooling" bottom: "x" top: "y" pooling_param { pool: MAX kernel_size: 3 stride: 2 }'
    arr = (C.c_void_p * 1)(x)
    assert L.dc_net_create_for_layer(text, 1, 1, arr, C.byref(ln)) == 0
    assert L.dc_net_num_inputs(ln) == 1 and L.dc_net_input_name(ln, 0) == b"x" and L.dc_net_output_name(ln, 0) == b"y"
    yb = C.c_void_p()
    assert L.dc_net_blob(ln, b"y", C.byref(yb)) == 0
    n, dd = C.c_int(), (C.c_int * 8)()
    assert L.dc_blob_shape(yb, C.byref(n), dd) == 0 and list(dd[:4]) == [1, 8, 3, 3]  # ceil-mode pooling
    L.dc_net_destroy(ln)
    assert L.dc_net_create_for_layer(text, 1, 0, None, C.byref(ln)) != 0 and b"declares 1 bottom" in L.dc_last_error()
    assert L.dc_net_create_for_layer(b'name: "q" type: "LRN" bottom: "x" top: "y"', 1, 1, arr, C.byref(ln)) != 0 and b"LRN" in L.dc_last_error()
    L.dc_blob_destroy(x)
    # device-only entries fail loudly (no CPU path), never silently
    assert L.dc_net_reserve(net._h, 1, 64, 64) != 0
    assert L.dc_net_emit_maps(net._h, None, None, None, 0, 0, None) != 0
    assert L.dc_net_forward_requests(net._h, 0, None, 64, 64, None, None, None, None) != 0
