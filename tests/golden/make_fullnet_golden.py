"""Generate tests/golden/fullnet_*.npz: the three output maps of the DeeperCut ResNet-152 forward
computed by the CPU oracle (oracle/) on a seeded input with seeded synthetic weights.  The fixture
stores the seeds and generator spec, not the 263 MB of weights.  These pin the oracle + weight
generator against drift (CPU test) and give the GPU tests a reference that needs no oracle run.

    python tests/golden/make_fullnet_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepcut-cnn_amd", "python"))

CASES = [("64x64", 64, 64, 1), ("104x136", 104, 136, 1), ("72x200_b2", 72, 200, 2)]


def main():
    from deepcut_tools import deepercut_prototxt, synth_weights
    from oracle import oracle as O

    O.set_threads(os.cpu_count() or 1)
    layers = synth_weights(152, seed=0)
    for tag, h, w, n in CASES:
        img = (np.random.RandomState(7).randn(n, 3, h, w) * 50).astype(np.float32)
        out = O.OracleNet(deepercut_prototxt(152, h, w, n), layers).forward(data=img)
        path = os.path.join(HERE, "fullnet_%s.npz" % tag)
        np.savez_compressed(path, weight_seed=0, depth=152, input_seed=7, input_scale=50.0, shape=(n, 3, h, w),
                            prob=out["prob"], loc_pred=out["loc_pred"], next_pred=out["next_pred"],
                            res5c_absmax=np.abs(out["res5c"]).max())
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
