"""Conditioned synthetic weights for the DeeperCut net.

The trained .caffemodel is not part of the reference tree (models/deepercut/download_models.sh
curls it), so parity tests and bench.py run on synthetic weights written in the reference's
.caffemodel wire format.  Naive N(0,1) weights blow activations up to ~1e5 after 152 layers, which
makes the north-star's 1e-3 max-abs bound meaningless, so the generator conditions them: He-scaled
convolutions, BatchNorm statistics near (0,1) with a non-trivial moving-average factor, residual-
branch gammas ~0.2 so the trunk stays O(1), small head weights so loc_pred/next_pred stay O(1).
Everything is a pure function of `seed`, so fixtures store the seed, not 263 MB of weights.
"""
import numpy as np

from .caffemodel import write_caffemodel
from .model_zoo import deepercut_layer_table


def synth_weights(depth=152, seed=0, num_joints=14, num_pairs=None, table=None):
    """-> list of (layer_name, layer_type, [ndarray,...]) in layer order."""
    table = table or deepercut_layer_table(depth, num_joints, num_pairs)
    chans = {"data": 3}
    out = []
    for l in table:
        rs = np.random.RandomState((seed * 1000003 + _stable_hash(l["name"])) % (2 ** 31 - 1))
        t = l["type"]
        cin = chans[l["bottoms"][0]]
        if t in ("Convolution", "Deconvolution"):
            p = l["conv"]
            cout, k = p["num_output"], p["kernel_size"]
            is_head = p.get("bias_term", True)
            if t == "Convolution":
                fan_in = cin * k * k
                std = (0.05 if is_head else 1.0) * np.sqrt(2.0 / fan_in)
                if l["bottoms"][0] == "data":
                    std /= 50.0  # mean-subtracted 8-bit pixels have std ~50: bring conv1 to O(1)
                w = rs.randn(cout, cin, k, k).astype(np.float32) * np.float32(std)
            else:
                fan_in = cin * (k * k) / 4.0  # each output pixel sees ~k*k/s*s taps
                std = 0.05 * np.sqrt(2.0 / fan_in)
                w = rs.randn(cin, cout, k, k).astype(np.float32) * np.float32(std)
            blobs = [w]
            if is_head:
                blobs.append((rs.randn(cout) * 0.1).astype(np.float32))
            out.append((l["name"], t, blobs))
            chans[l["tops"][0]] = cout
        elif t == "BatchNorm":
            sf = np.float32(999.98236)  # a realistic moving-average normaliser (blob2[0])
            mean = (rs.randn(cin) * 0.05).astype(np.float32) * sf
            var = (1.0 + 0.2 * rs.rand(cin)).astype(np.float32) * sf
            out.append((l["name"], t, [mean, var, np.array([sf], np.float32)]))
            chans[l["tops"][0]] = cin
        elif t == "Scale":
            residual_end = l["name"].endswith("branch2c")
            g = (0.2 if residual_end else 1.0) * (1.0 + 0.1 * rs.randn(cin))
            b = 0.05 * rs.randn(cin)
            out.append((l["name"], t, [g.astype(np.float32), b.astype(np.float32)]))
            chans[l["tops"][0]] = cin
        else:
            chans[l["tops"][0]] = cin
    return out


def _stable_hash(s):
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def write_synth_caffemodel(path, depth=152, seed=0, num_joints=14, num_pairs=None, table=None):
    layers = synth_weights(depth, seed, num_joints, num_pairs, table)
    write_caffemodel(path, "ResNet-%d" % depth, layers)
    return layers
