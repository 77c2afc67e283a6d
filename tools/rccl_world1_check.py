"""World-size-1 RCCL sanity check (ON THE GPU BOX): the process group bench.py / ShardedPoseRunner would create at N > 1 can be
initialised here, and barrier / all_reduce / the known-size gather run through it (with one rank the gather has no peers)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deepcut-cnn_amd", "python"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
t = torch.ones(4, device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
from deepcut_tools import gather_maps_known  # noqa: E402

out = gather_maps_known(torch.arange(8, device=dev, dtype=torch.float16), [8], 0, None)
torch.cuda.synchronize()
print("rccl world-1 ok:", dist.get_backend(), t.tolist(), out[0].tolist())
dist.destroy_process_group()
