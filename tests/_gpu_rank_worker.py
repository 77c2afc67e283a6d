"""One rank of tests/test_gpu_configs.py::test_two_ranks_sharing_gpu0_equal_world_size_1 (launched by torch.distributed.run):
ShardedPoseRunner with the real Net over gloo, every rank on GPU 0.  Rank 0 writes the gathered result to argv[2]."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCALES = [0.75, 1.0, 1.25]


def worker_images(mode="mixed"):
    rs = np.random.RandomState(5)
    if mode == "c3":  # BASELINE configs[3] in small: 64 equal images, one scale, dealt 8 per rank over 8 ranks
        return [rs.randint(0, 256, (96, 128, 3)).astype(np.uint8) for _ in range(64)]
    return [rs.randint(0, 256, (h, w, 3)).astype(np.uint8) for (h, w) in [(96, 128)] * 5 + [(120, 88)] * 2]


def main():
    import torch.distributed as dist

    weights, out = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else "mixed"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    import caffe
    from deepcut_tools import ShardedPoseRunner, deepercut_prototxt

    caffe.set_mode_gpu()
    caffe.set_device(0)
    net = caffe.Net(deepercut_prototxt(152, 96, 128), weights, caffe.TEST, from_text=True, hipgraph=1)
    if mode == "c3":
        res = ShardedPoseRunner(net, max_batch=8, depth=2).run(worker_images("c3"), [1.0], want_maps=True)
    else:
        res = ShardedPoseRunner(net, max_batch=4, depth=2).run(worker_images(), SCALES, want_maps=True)
    if dist.get_rank() == 0:
        d = {"item_poses": res["item_poses"], "best_scale": np.array([s if s is not None else -1.0 for s in res["best_scale"]])}
        for k, m in res["maps"].items():
            for name, v in m.items():
                d["%s_%d" % (name, k)] = v
        np.savez(out, **d)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
