"""Generate tests/golden/pose_golden.npz by IMPORTING the reference's python/pose/estimate_pose.py
(build container only: /root/reference does not exist on the GPU box).  The reference module imports
`caffe` and `scipy.misc` at load time; pycaffe cannot be built here (Boost.Python), so an empty module
object named `caffe` is put in sys.modules for the import to succeed — none of the functions exercised
below touch it (they are pure NumPy).  Stored: inputs and the reference's outputs only.

    python tests/golden/make_pose_golden.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/python/pose"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_golden.npz")


def main():
    sys.modules.setdefault("caffe", types.ModuleType("caffe"))
    sys.path.insert(0, REF)
    import estimate_pose as ref

    rs = np.random.RandomState(1234)
    data = {}
    cases = [(15, 20, 1.0), (34, 46, 1.0), (13, 17, 0.75), (9, 25, 1.25), (1, 1, 0.5)]
    for i, (h, w, scale) in enumerate(cases):
        prob = rs.rand(14, h, w).astype(np.float32)
        if i == 1:  # ties: the first maximum in row-major order must win
            prob[3] = 0.5
            prob[5, 10:12, 7:9] = 2.0
        loc = rs.randn(28, h, w).astype(np.float32)
        # the reference's layout: scoremat (h,w,14), offmat (h,w,14,2) built as in _cnn_process_image /
        # _process_image_tiled (estimate_pose.py:231-242, 221)
        scoremat = prob.transpose(1, 2, 0)
        offmat = loc.reshape(14, 2, h, w).transpose(2, 3, 1, 0).transpose(0, 1, 3, 2)
        pose = ref._pose_from_mats(scoremat, offmat, scale)
        data["prob_%d" % i], data["loc_%d" % i] = prob, loc
        data["scale_%d" % i], data["pose_%d" % i] = np.float64(scale), pose
    lengths = np.array([1, 8, 699, 700, 701, 952, 953, 1204, 1205, 1456, 1457, 2000, 4000])
    data["tile_lengths"] = lengths
    data["tile_counts"] = np.array([ref._get_num_tiles(int(l), 700, 224) for l in lengths])
    data["mean"] = np.asarray(ref._MEAN, np.float64)
    data["locref_scale"] = np.float64(ref._LOCREF_SCALE_MUL)
    data["stride"] = np.float64(ref._STRIDE)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
