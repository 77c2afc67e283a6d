"""Generate tests/golden/weights_golden.h5 with the real HDF5 library, calling exactly what the reference's
Net::ToHDF5 calls (src/caffe/net.cpp:926-975, src/caffe/util/hdf5.cpp:85-125):

    H5Fcreate(TRUNC) ; H5Gcreate2(file, "data") ; H5Gcreate2(file, "diff")
    per layer with parameters:  H5Gcreate2(data, <layer name>) ; H5LTmake_dataset_float(layer, "<i>", ndims, dims, ptr)

The image has no h5py and no HDF5 headers, but /opt/conda/lib/libhdf5{,_hl}.so is there and is driven through
ctypes.  `write_caffe_h5()` is also what tests/test_hdf5_weights.py uses to produce larger files on the fly (hundreds
of layers: multi-level group B-trees) when the library can be loaded; the committed fixture is the small one below.

    python tests/golden/make_hdf5_golden.py
"""
import ctypes as C
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights_golden.h5")
_LIBS = None


def libs():
    """(libhdf5, libhdf5_hl) or None when they cannot be loaded."""
    global _LIBS
    if _LIBS is None:
        _LIBS = ()
        for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", ""):
            try:
                h = C.CDLL(os.path.join(d, "libhdf5.so"), mode=C.RTLD_GLOBAL)
                hl = C.CDLL(os.path.join(d, "libhdf5_hl.so"))
            except OSError:
                continue
            hid = C.c_int64
            h.H5open()
            h.H5Fcreate.restype = hid
            h.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid, hid]
            h.H5Gcreate2.restype = hid
            h.H5Gcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid]
            h.H5Gclose.argtypes = [hid]
            h.H5Fclose.argtypes = [hid]
            for fn in (hl.H5LTmake_dataset_float, hl.H5LTmake_dataset_double):
                fn.restype = C.c_int
                fn.argtypes = [hid, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]
            _LIBS = (h, hl)
            break
    return _LIBS or None


def write_caffe_h5(path, layers, double_layers=()):
    """layers: [(name, [ndarray, ...])] -> the file Net::ToHDF5 would write (write_diff = false)."""
    h, hl = libs()
    H5F_ACC_TRUNC, H5P_DEFAULT = 2, 0
    f = h.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    assert f >= 0
    data = h.H5Gcreate2(f, b"data", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
    diff = h.H5Gcreate2(f, b"diff", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)  # present but empty, as with write_diff = false
    for name, blobs in layers:
        g = h.H5Gcreate2(data, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        assert g >= 0, name
        for i, b in enumerate(blobs):
            dbl = name in double_layers
            a = np.ascontiguousarray(b, np.float64 if dbl else np.float32)
            dims = (C.c_uint64 * max(1, a.ndim))(*a.shape)
            fn = hl.H5LTmake_dataset_double if dbl else hl.H5LTmake_dataset_float
            assert fn(g, str(i).encode(), a.ndim, dims, a.ctypes.data_as(C.c_void_p)) >= 0
        h.H5Gclose(g)
    h.H5Gclose(diff)
    h.H5Gclose(data)
    h.H5Fclose(f)


def golden_layers():
    rs = np.random.RandomState(42)
    return [("c1", [rs.randn(32, 4, 3, 3).astype(np.float32), rs.randn(32).astype(np.float32)]),
            ("bn_c1", [rs.randn(32).astype(np.float32), rs.rand(32).astype(np.float32) + 0.5, np.array([999.98236], np.float32)]),
            ("scale_c1", [rs.rand(32).astype(np.float32) + 0.5, rs.randn(32).astype(np.float32)]),
            ("up", [rs.randn(32, 2, 3, 3).astype(np.float32), rs.randn(2).astype(np.float32)]),
            ("not_in_the_net", [rs.randn(3, 5).astype(np.float32)])]


def main():
    write_caffe_h5(OUT, golden_layers(), double_layers=("scale_c1",))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
