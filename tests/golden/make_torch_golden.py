"""Generate tests/golden/torch64_*.npz: the three output maps (and a few trunk statistics) of the DeeperCut forward computed by the
INDEPENDENT torch float64 graph of tests/test_oracle_torch_graph.py (hand-written from the ResNet / DeeperCut definition, it never
reads a prototxt and shares no code with oracle/ or the library) on a seeded input with the seeded synthetic weights.  Unlike
fullnet_*.npz (written by the oracle: regression pins) these are a second opinion: the CPU suite holds the oracle to them, the GPU suite
the HIP path — without either having produced them.

    python tests/golden/make_torch_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

CASES = [("r152_64x64", 152, 64, 64, 1), ("r152_72x104_b2", 152, 72, 104, 2), ("r101_64x80", 101, 64, 80, 1), ("r152_60x76", 152, 60, 76, 1)]


def main():
    import torch

    from deepcut_tools import synth_weights
    from test_oracle_torch_graph import _torch_deepercut

    for tag, depth, h, w, n in CASES:
        layers = synth_weights(depth, seed=5)
        W = {name: [torch.from_numpy(np.asarray(b, np.float64)) for b in blobs] for name, _t, blobs in layers}
        img = (np.random.RandomState(21).randn(n, 3, h, w) * 50).astype(np.float32)
        with torch.no_grad():
            out = _torch_deepercut(torch.from_numpy(img.astype(np.float64)), W, depth)
        path = os.path.join(HERE, "torch64_%s.npz" % tag)
        np.savez_compressed(path, weight_seed=5, depth=depth, input_seed=21, input_scale=50.0, shape=(n, 3, h, w),
                            prob=out["prob"].numpy().astype(np.float32), loc_pred=out["loc_pred"].numpy().astype(np.float32),
                            next_pred=out["next_pred"].numpy().astype(np.float32),
                            res3_last_absmax=float(out["res3b7" if depth == 152 else "res3b3"].abs().max()),
                            res5c_absmax=float(out["res5c"].abs().max()))
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
