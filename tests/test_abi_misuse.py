"""Every entry point of the C ABI called the wrong way — null handles and null outputs everywhere, indices out of range, ranks
and dimensions that overflow — on the CPU, in a process of its own: an error code (or NULL) every time, never a crash.  The
reference CHECK-fails (aborts the process) on most of these (blob.cpp:23-43, net.cpp:405-450); `include/deepcut_hip.h` promises
"nothing aborts the process"."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NULLS = r'''
import ctypes as C, re, sys
hdr = open(sys.argv[1]).read()
hdr = re.sub(r'/\*.*?\*/', ' ', hdr, flags=re.S); hdr = re.sub(r'//[^\n]*', ' ', hdr)
protos = re.findall(r'\b(int|const char\*)\s+(dc_[a-z_0-9]+)\s*\(([^)]*)\)\s*;', hdr)
lib = C.CDLL(sys.argv[2])
assert len(protos) >= 79, len(protos)
for ret, name, params in protos:
    ps = [p.strip() for p in params.split(',')] if params.strip() not in ('', 'void') else []
    args = [C.c_void_p(0) if '*' in p else C.c_double(0.0) if p.startswith('double') else C.c_float(0.0) if p.startswith('float') else C.c_int(0)
            for p in ps]
    f = getattr(lib, name)
    f.restype = C.c_char_p if ret != 'int' else C.c_int
    print(name, flush=True)
    r = f(*args)
    takes_handle = any(('dc_net*' in p or 'dc_blob*' in p or 'dc_group*' in p) and '**' not in p for p in ps)
    if takes_handle and ret == 'int' and not name.endswith('_destroy'):
        assert r <= 0, (name, r)   # an error code, or a count of nothing
print('DONE')
'''

BAD = r'''
import ctypes as C, sys
lib = C.CDLL(sys.argv[2])
for n in ('dc_net_layer_name', 'dc_net_blob_name', 'dc_net_input_name', 'dc_net_output_name', 'dc_conv_variant_name', 'dc_last_error'):
    getattr(lib, n).restype = C.c_char_p
T = b"""name: "n" input: "data" input_dim: 1 input_dim: 3 input_dim: 9 input_dim: 9
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 4 kernel_size: 3 } }
"""
net = C.c_void_p()
assert lib.dc_net_create_from_text(T, None, 1, C.byref(net)) == 0
for i in (-1, 5, 10**9, -2**31):
    for f in (lib.dc_net_layer_name, lib.dc_net_blob_name, lib.dc_net_input_name, lib.dc_net_output_name):
        assert f(net, i) is None, (f, i)
for i in (-1, 10**9, -2**31):
    assert lib.dc_conv_variant_name(i) is None and lib.dc_conv_variant_esize(i) == 0
b = C.c_void_p()
assert lib.dc_net_blob(net, None, C.byref(b)) < 0 and lib.dc_net_blob(net, b"data", None) < 0
assert lib.dc_net_blob(net, b"data", C.byref(b)) == 0
for idx in (-1, 2, 99):
    assert lib.dc_net_param(net, b"c1", idx, C.byref(C.c_void_p())) < 0
assert lib.dc_net_param(net, None, 0, C.byref(C.c_void_p())) < 0
dims = (C.c_int * 40)(*([1] * 40))
assert lib.dc_blob_reshape(b, 40, dims) < 0 and lib.dc_blob_reshape(b, -1, dims) < 0 and lib.dc_blob_reshape(b, 4, None) < 0
assert lib.dc_blob_reshape(b, 4, (C.c_int * 4)(1, 3, -5, 9)) == -3
# Blob::Reshape: "blob size exceeds INT_MAX" (blob.cpp:31-34); 65536^4 wraps a 64-bit count to 0
big = (C.c_int * 4)(65536, 65536, 65536, 65536)
assert lib.dc_blob_reshape(b, 4, big) == -3 and b"INT_MAX" in lib.dc_last_error()
assert lib.dc_blob_reshape(b, 4, (C.c_int * 4)(2, 32768, 32768, 1)) == -3   # 2^31: one past INT_MAX
assert lib.dc_blob_create(4, big, C.byref(C.c_void_p())) < 0
assert lib.dc_blob_create(40, dims, C.byref(C.c_void_p())) < 0
assert lib.dc_blob_shape(b, None, None) < 0
nd, d8 = C.c_int(), (C.c_int * 8)()
assert lib.dc_blob_shape(b, C.byref(nd), d8) == 0 and list(d8)[:nd.value] == [1, 3, 9, 9]   # the refused reshapes left it alone
assert lib.dc_net_forward(net, 5, -3, None) < 0
assert lib.dc_net_set_option(net, 99, 1) < 0 and lib.dc_net_get_option(net, 99, C.byref(C.c_int())) < 0
assert lib.dc_net_reserve(net, -1, -1, -1) < 0
assert lib.dc_image_canvas_size(-5, 10, C.c_double(1e308), C.byref(C.c_int()), C.byref(C.c_int())) < 0
assert lib.dc_image_canvas_size(50, 10, C.c_double(float("nan")), C.byref(C.c_int()), C.byref(C.c_int())) < 0
assert lib.dc_group_create((C.c_void_p * 1)(net), -1, C.byref(C.c_void_p())) < 0
assert lib.dc_group_create((C.c_void_p * 2)(net, None), 2, C.byref(C.c_void_p())) < 0
assert lib.dc_net_set_tile(net, None, None) < 0
assert lib.dc_net_copy_from(net, b"/nonexistent") == -2 and lib.dc_net_save(net, b"/nonexistent_dir/x.caffemodel") == -2
assert lib.dc_net_destroy(net) == 0
print('DONE')
'''


def _run(code):
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code), os.path.join(ROOT, "include", "deepcut_hip.h"),
                        os.path.join(ROOT, "deepcut-cnn_amd", "lib", "libdeepcut_hip.so")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("DONE"), (r.stdout[-1500:] + r.stderr[-2500:])


def test_null_arguments_everywhere():
    _run(NULLS)


def test_out_of_range_indices_ranks_and_sizes():
    _run(BAD)
