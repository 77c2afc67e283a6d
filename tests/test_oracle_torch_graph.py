"""CPU: an INDEPENDENT whole-graph check of the oracle (SURVEY §7.1's torch-CPU cross-check).

`OracleNet` (oracle/oracle.py) is what every GPU parity test trusts for the graph-level semantics of the reference: file-order
execution with in-place tops (src/caffe/net.cpp:394-400,565-581), Eltwise operand order, the fork's Crop, ceil-mode pooling, the
deconvolution + crop + skip-add heads, and its own prototxt reader.  The per-op pins (tests/test_oracle_reference_vectors.py) cannot
see a mistake there, and tests/test_fullnet_golden.py pins the oracle to itself.  Here the DeeperCut graph is written a second time,
by hand, as a torch float64 functional graph that never reads a prototxt — wiring straight from the ResNet / DeeperCut definition
(He et al.; models/deepercut/ResNet-152.prototxt:67-7344 read as a paper figure, not parsed) — and every Caffe-visible blob of the
oracle is compared with it.  When the reference tree is present (build container) the oracle parses the reference's OWN
`models/deepercut/ResNet-152.prototxt`; otherwise (the GPU box's CPU suite) the generated text, which tests/test_formats.py holds
equivalent to it."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O

REF_PROTO = "/root/reference/models/deepercut/ResNet-152.prototxt"
_BLOCKS = {152: (3, 8, 36, 3), 101: (3, 4, 23, 3)}


def _torch_deepercut(x, W, depth):
    """float64 functional forward; W: layer name -> list of float64 tensors.  Returns name -> tensor for the blobs that are
    visible after a Caffe forward (in-place chains hold their final value)."""
    B = {}

    def bn_scale(t, tag):
        m, v, sf = W["bn" + tag]
        f = 0.0 if float(sf[0]) == 0 else 1.0 / float(sf[0])
        t = (t - (m * f).view(1, -1, 1, 1)) / torch.sqrt((v * f).view(1, -1, 1, 1) + 1e-5)
        g, b = W["scale" + tag]
        return t * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    def conv(t, name, stride=1, pad=0, dil=1):
        w = W[name]
        return F.conv2d(t, w[0], w[1] if len(w) > 1 else None, stride, pad, dil)

    t = torch.relu(bn_scale(conv(x, "conv1", 2, 3), "_conv1"))
    B["conv1"] = t
    t = F.max_pool2d(t, 3, 2, ceil_mode=True)
    B["pool1"] = t
    counts = _BLOCKS[depth]
    for si, stage in enumerate((2, 3, 4, 5)):
        for bi in range(counts[si]):
            if stage in (2, 5):
                tag = "%d%s" % (stage, "abc"[bi])
            else:
                tag = "%da" % stage if bi == 0 else "%db%d" % (stage, bi)
            stride = 2 if (bi == 0 and stage in (3, 4)) else 1
            dil = 2 if stage == 5 else 1
            short = t
            if bi == 0:
                short = bn_scale(conv(t, "res%s_branch1" % tag, stride), "%s_branch1" % tag)
                B["res%s_branch1" % tag] = short
            a = torch.relu(bn_scale(conv(t, "res%s_branch2a" % tag, stride), "%s_branch2a" % tag))
            b = torch.relu(bn_scale(conv(a, "res%s_branch2b" % tag, 1, dil, dil), "%s_branch2b" % tag))
            c = bn_scale(conv(b, "res%s_branch2c" % tag), "%s_branch2c" % tag)
            B["res%s_branch2a" % tag], B["res%s_branch2b" % tag], B["res%s_branch2c" % tag] = a, b, c
            t = torch.relu(short + c)
            B["res%s" % tag] = t
        if stage == 3:
            res3 = t
    for suffix, out in (("pose", "fc_pose"), ("locref", "loc_pred"), ("next", "next_pred")):
        wu = W["res5c_up_" + suffix]
        up = F.conv_transpose2d(t, wu[0], wu[1], 2)
        skip = conv(res3, "res3d_" + suffix)
        B["res5c_up_" + suffix], B["res3d_" + suffix] = up, skip
        assert up.shape[2] > skip.shape[2] and up.shape[3] > skip.shape[3]  # crop_layer.cpp:30-32
        B["res5c_up_%sc" % suffix] = up[:, :, : skip.shape[2], : skip.shape[3]]
        B[out] = skip + B["res5c_up_%sc" % suffix]
    B["prob"] = torch.sigmoid(B["fc_pose"])
    return B


@pytest.mark.parametrize("depth,hw,batch", [(152, (64, 64), 1), (152, (72, 104), 2), (101, (64, 80), 1)])
def test_oracle_graph_equals_independent_torch_float64_graph(depth, hw, batch):
    from deepcut_tools import deepercut_prototxt, synth_weights

    h, w = hw
    layers = synth_weights(depth, seed=3)
    if depth == 152 and os.path.exists(REF_PROTO):
        # the reference's own file, its input_dim lines rewritten to this test's shape (ResNet-152.prototxt:3-9: 1x3x688x688)
        text = open(REF_PROTO).read()
        import re

        dims = iter((batch, 3, h, w))
        text = re.sub(r"input_dim:\s*\d+", lambda m: "input_dim: %d" % next(dims), text, count=4)
    else:
        text = deepercut_prototxt(depth, h, w, batch)
    img = (np.random.RandomState(11).randn(batch, 3, h, w) * 50).astype(np.float32)
    O.set_threads(min(8, os.cpu_count() or 1))
    got = O.OracleNet(text, layers).forward(data=img)
    W = {name: [torch.from_numpy(np.asarray(b, np.float64)) for b in blobs] for name, _t, blobs in layers}
    with torch.no_grad():
        ref = _torch_deepercut(torch.from_numpy(img.astype(np.float64)), W, depth)
    assert set(ref) | {"data"} == set(got), sorted(set(got) ^ set(ref))[:8]
    worst = ("", 0.0)
    for name, r in ref.items():
        r = r.numpy()
        assert got[name].shape == r.shape, name
        err = float(np.abs(got[name] - r).max()) / max(1.0, float(np.abs(r).max()))
        if err > worst[1]:
            worst = (name, err)
        assert err <= 2e-5, "%s: %g" % (name, err)  # measured ~1.2e-6
    for k in ("prob", "loc_pred", "next_pred"):
        assert float(np.abs(got[k] - ref[k].numpy()).max()) <= 2e-5, k
    print("worst blob", worst)
