"""`caffe` — drop-in pycaffe surface for the DeeperCut forward path on MI355X.

Mirrors what python/pose/estimate_pose.py and pose_demo.py use of the reference's pycaffe
(python/caffe/__init__.py:1-4, python/caffe/pycaffe.py, python/caffe/_caffe.cpp): `Net`, `Blob`,
`TEST`/`TRAIN`, `set_mode_cpu`, `set_mode_gpu`, `set_device`.  It is a ctypes binding of
libdeepcut_hip.so (include/deepcut_hip.h); there is no compute in Python and no CPU fallback:
`forward()` in CPU mode raises.
"""
from .pycaffe import Net, NetGroup, Comm, Blob, Layer, choose_streams, pinned_empty, lpt_schedule, set_mode_cpu, set_mode_gpu, set_device, device_count, canvas_size, conv_variants, wino_half_pack, stream1x1_pack, stream1x1f_pack, stem7x7_pack, lib_path  # noqa: F401
from .pycaffe import TRAIN, TEST, DeepcutError  # noqa: F401

__version__ = "1.0.0-rc3+deepcut_hip"
