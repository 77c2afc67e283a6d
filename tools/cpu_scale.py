"""Diagnostics: thread scaling of the CPU oracle on this host (prints the cgroup CPU quota it finds)."""
import sys, os, time
sys.path.insert(0,'deepcut-cnn_amd/python'); sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from deepcut_tools import deepercut_prototxt, synth_weights
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
for f in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)' ")
wts=synth_weights(152,0)
img=(np.random.RandomState(1).randn(1,3,240,320)*50).astype(np.float32)
for nt in (1,4,8,16,32,64,128):
    O.set_threads(nt)
    net=O.OracleNet(deepercut_prototxt(152,240,320),wts)
    net.forward(data=img)
    t=time.time(); net.forward(data=img); dt=time.time()-t
    print(nt,'threads %.2f s %.1f GFLOP/s'%(dt,46.24/dt), flush=True)
