"""-m gpu: every tile variant of the gather-GEMM kernel, forced one at a time (DC_CONV_VARIANT) over the whole
ResNet-152 graph at 72x104, against the CPU oracle.  Autotuning only ever times the variants; this is where each
of them has to be right (4- and 8-wave workgroups, in-workgroup split-K 1/2/4/8, fp32 and fp16 operands)."""
import os

import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu
H, W = [int(v) for v in os.environ.get("DC_TEST_VARIANT_HW", "72,104").split(",")]  # override to sweep another input size
MAX_VARIANTS = 64  # parametrisation bound; indices past the library's table (caffe.conv_variants()) are skipped, a longer table fails


@pytest.fixture(scope="module")
def reference(synth152):
    from deepcut_tools import deepercut_prototxt
    from oracle import oracle as O

    path, layers = synth152
    O.set_threads(min(16, os.cpu_count() or 1))
    img = rand_image(9, H, W)
    return img, O.OracleNet(deepercut_prototxt(152, H, W), layers).forward(data=img)


def _run(gpu_caffe, synth152, v, dtype, img, monkeypatch):
    from deepcut_tools import deepercut_prototxt

    path, _ = synth152
    monkeypatch.setenv("DC_CONV_VARIANT", str(v))
    net = gpu_caffe.Net(deepercut_prototxt(152, H, W), path, gpu_caffe.TEST, from_text=True, dtype=dtype)
    net.blobs["data"].data[...] = img
    net.forward()
    used = set(ln.split("\t")[1] for ln in net.plan_text().splitlines() if "conv_gemm<" in ln)
    return net, used


@pytest.mark.parametrize("v", range(MAX_VARIANTS))
def test_forced_variant_matches_oracle(gpu_caffe, synth152, reference, monkeypatch, v):
    table = gpu_caffe.conv_variants()
    if v >= len(table):
        pytest.skip("the variant table has %d entries" % len(table))
    name, esize = table[v]
    img, ref = reference
    net, used = _run(gpu_caffe, synth152, v, "f16" if esize == 2 else "f32", img, monkeypatch)
    # layers whose K segments the forced variant cannot take fall back to the cost model's choice, so a few names may
    # appear beside it; the forced one must be there
    assert "conv_gemm<%s>" % name in used, (name, sorted(used))
    assert len(used) <= 4, "variant %d (%s) was not forced: %s" % (v, name, sorted(used))
    if esize == 2:
        assert float(np.abs(net.blobs["prob"].data - ref["prob"]).max()) <= 2.5e-3
        for k in ("loc_pred", "next_pred"):
            assert float(np.abs(net.blobs[k].data - ref[k]).max()) <= 4e-3 * max(1.0, float(np.abs(ref[k]).max()))
    else:
        for k in ("prob", "loc_pred", "next_pred"):
            assert float(np.abs(net.blobs[k].data - ref[k]).max()) <= 1e-3, (k, sorted(used))


def test_variant_table_is_covered(gpu_caffe, synth152, reference, monkeypatch):
    table = gpu_caffe.conv_variants()
    assert 0 < len(table) <= MAX_VARIANTS, "raise MAX_VARIANTS: the forced-variant test does not reach the end of the table"
    assert len(set(n for n, _ in table)) == len(table), "variant names are the tune-cache keys: they must be unique"
    img, _ = reference
    monkeypatch.setenv("DC_AUTOTUNE", "0")  # cost-model choice: what a forced index falls back to where it cannot apply
    _, base = _run(gpu_caffe, synth152, -1, "f32", img, monkeypatch)
    _, beyond = _run(gpu_caffe, synth152, len(table), "f32", img, monkeypatch)  # out of range: nothing is forced
    assert beyond == base
