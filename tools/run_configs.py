#!/usr/bin/env python
"""BASELINE.json configs 3 and 5 (and config 4's per-GPU share) on ONE MI355X, end to end through
deepcut_tools.ShardedPoseRunner: uint8 images on the host -> device pre-processing (Pillow-exact) -> forward -> pose
decode on the device -> best scale per image.  Synthetic images / conditioned synthetic weights.

    python tools/run_configs.py > profiles/r01_configs.json

config 3: 8 images 736x544, scales 0.5/0.75/1.0/1.25, fp16 operands, batch 8 per scale        (823.8 GFLOP per pyramid)
config 4: 64 images 736x544 sharded 8-way = 8 images per GPU, fp32 and fp16 (this is ONE rank's share, no gather)
config 5: 32 crops 256x336, 4 scales, maps kept (next_pred on), fp32; all 128 forwards on this one GPU"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deepcut-cnn_amd", "python"))

import numpy as np  # noqa: E402


def main():
    gs = int(os.environ.get("DC_RUN_GROUP_SIZE", "4"))  # batches of a rank that run as one grouped launch sequence (1: none)
    import caffe
    from deepcut_tools import ShardedPoseRunner, deepercut_prototxt, synth_weights

    caffe.set_mode_gpu()
    caffe.set_device(0)
    layers = synth_weights(152, seed=0)

    def make_net(dtype):
        net = caffe.Net(deepercut_prototxt(152, 544, 736), caffe.TEST, from_text=True, dtype=dtype)
        for name, _t, blobs in layers:
            for p, b in zip(net.params[name], blobs):
                p.data[...] = b
        return net

    def timed(fn, reps):
        fn()  # shapes, plans, tuning
        fn()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t) / reps

    scales = [0.5, 0.75, 1.0, 1.25]
    out = {"device": "1 x MI355X", "group_size": gs, "data": "synthetic uint8 images (RandomState seeds as in SURVEY 8d), conditioned synthetic weights seed 0",
           "path": "ShardedPoseRunner: host uint8 -> dc_net_forward_images (pre-processing, forward, pose decode on the device)"}
    imgs8 = [np.random.RandomState(10 + i).randint(0, 256, (544, 736, 3)).astype(np.uint8) for i in range(8)]
    crops = [np.random.RandomState(200 + i).randint(0, 256, (336, 256, 3)).astype(np.uint8) for i in range(32)]

    half = make_net("f16")
    r = ShardedPoseRunner(half, max_batch=8, group_size=gs)
    dt = timed(lambda: r.run(imgs8, scales), 5)
    out["config3_pyramid_fp16_batch8"] = {"seconds_per_batch_of_8_pyramids": dt, "images_per_s": 8 / dt, "forwards_per_s": 32 / dt,
                                          "tflops": 8 * 823.8e9 / dt / 1e12}
    dt = timed(lambda: r.run(imgs8, [1.0]), 10)
    out["config4_share_8_images_fp16"] = {"seconds": dt, "images_per_s": 8 / dt}
    r2 = ShardedPoseRunner(half, max_batch=8, depth=2, group_size=gs)
    dt = timed(lambda: r2.run(imgs8, scales), 5)
    out["config3_pyramid_fp16_batch8_2_batches_in_flight"] = {"seconds_per_batch_of_8_pyramids": dt, "images_per_s": 8 / dt,
                                                              "forwards_per_s": 32 / dt, "tflops": 8 * 823.8e9 / dt / 1e12}
    del r, r2, half

    full = make_net("f32")
    r = ShardedPoseRunner(full, max_batch=8, group_size=gs)
    dt = timed(lambda: r.run(imgs8, [1.0]), 10)
    out["config4_share_8_images_fp32"] = {"seconds": dt, "images_per_s": 8 / dt}
    r = ShardedPoseRunner(full, max_batch=16, group_size=gs)
    dt = timed(lambda: r.run(crops, scales, want_maps=True), 3)
    out["config5_32_crops_4_scales_fp32_with_maps"] = {"seconds": dt, "crops_per_s": 32 / dt, "forwards_per_s": 128 / dt,
                                                       "tflops": 32 * 177.8e9 / dt / 1e12,
                                                       "note": "maps (prob, loc_pred, next_pred) copied to the host for every forward"}
    dt = timed(lambda: r.run(crops, scales), 3)
    out["config5_32_crops_4_scales_fp32_poses_only"] = {"seconds": dt, "crops_per_s": 32 / dt, "forwards_per_s": 128 / dt,
                                                        "tflops": 32 * 177.8e9 / dt / 1e12}
    r3 = ShardedPoseRunner(full, max_batch=16, depth=3, group_size=gs)
    dt = timed(lambda: r3.run(crops, scales), 3)
    out["config5_32_crops_4_scales_fp32_poses_only_3_batches_in_flight"] = {"seconds": dt, "crops_per_s": 32 / dt, "forwards_per_s": 128 / dt,
                                                                            "tflops": 32 * 177.8e9 / dt / 1e12}
    r3 = ShardedPoseRunner(full, max_batch=4, depth=3, group_size=gs)
    dt = timed(lambda: r3.run(imgs8, [1.0]), 10)
    out["config4_share_8_images_fp32_3_batches_of_4_in_flight"] = {"seconds": dt, "images_per_s": 8 / dt}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
