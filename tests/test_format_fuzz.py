"""The library's three file parsers under AddressSanitizer + UBSan (tools/probes/fuzz_formats.cpp; CPU only): a model definition
in protobuf text format (ReadProtoFromTextFile, io.cpp:34-43), weight files in the current, V1 and V0 wire formats
(ReadProtoFromBinaryFile io.cpp:52-65 + UpgradeNetAsNeeded upgrade_proto.cpp:19-78) and an HDF5 weight file
(CopyTrainedLayersFromHDF5, net.cpp:861-975 — decoded here without libhdf5) are mutated a few thousand times each; every mutant
must parse or be refused with an exception: no crash, no out-of-bounds access, no undefined arithmetic, no allocation the file
cannot justify, no parse that takes seconds.  The reference leaves all of this to libprotobuf / libhdf5; a from-scratch
decoder of untrusted files owes the check itself."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from deepcut_tools import deepercut_prototxt
from deepcut_tools.caffemodel import _ld, write_caffemodel
from test_legacy_formats import V1, _v1_layer, _weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITER = int(os.environ.get("DC_FORMAT_FUZZ_ITER", "1500"))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("fuzz") / "fuzz_formats")
    src = [os.path.join(ROOT, p) for p in ("tools/probes/fuzz_formats.cpp", "deepcut-cnn_amd/csrc/formats.cpp", "deepcut-cnn_amd/csrc/hdf5_reader.cpp")]
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", os.path.join(ROOT, "include")] + src + ["-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr.lower() and "cannot find" in r.stderr.lower():
        pytest.skip("no sanitizer runtime in this image")
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(exe, kind, seed_path, rng):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:allocator_may_return_null=0:max_allocation_size_mb=2048")
    r = subprocess.run([exe, kind, str(seed_path), str(ITER), str(rng)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "mutants" in r.stdout
    return r.stdout


@pytest.mark.parametrize("which", ["current", "v1"])
def test_text_format_mutants(harness, tmp_path, which):
    text = deepercut_prototxt(50, 64, 80) if which == "current" else V1
    p = tmp_path / "seed.prototxt"
    p.write_text(text[:20000] if which == "current" else text)  # (the head of the definition: every construct, a tenth of the time)
    if which == "current":  # cut at a layer boundary so that the seed itself parses
        head = text[:20000]
        p.write_text(head[:head.rindex("\nlayer {")] + "\n")
    _run(harness, "text", p, 1)


@pytest.mark.parametrize("which", ["current", "v1", "v0"])
def test_caffemodel_mutants(harness, tmp_path, which):
    w = _weights()
    p = tmp_path / "seed.caffemodel"
    if which == "current":
        rs = np.random.RandomState(1)
        layers = [("c1", "Convolution", [w["c1"][0], w["c1"][1]]), ("bn", "BatchNorm", [rs.rand(32).astype(np.float32), rs.rand(32).astype(np.float32),
                  np.ones(1, np.float32)]), ("up", "Deconvolution", [w["up"][0], w["up"][1]])]
        write_caffemodel(str(p), "seed", layers)
    else:
        v0 = which == "v0"
        body = _ld(1, b"legacy") + _ld(2, _v1_layer("c1", 4, w["c1"], v0)) + _ld(2, _v1_layer("r", 18, [], False)) + _ld(2, _v1_layer("up", 39, w["up"], v0))
        p.write_bytes(body)
    _run(harness, "model", p, 2)


def test_hdf5_mutants(harness):
    out = _run(harness, "hdf5", os.path.join(ROOT, "tests", "golden", "weights_golden.h5"), 3)
    # most mutants of a 20-KB file hit dataset bytes and still parse; the structural ones are refused
    assert " parsed" in out and " refused" in out
