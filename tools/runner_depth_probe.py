import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in (ROOT, ROOT+"/deepcut-cnn_amd", ROOT+"/deepcut-cnn_amd/python"): sys.path.insert(0,p)
import numpy as np
import caffe
from deepcut_tools import ShardedPoseRunner, deepercut_prototxt, synth_weights
caffe.set_mode_gpu(); caffe.set_device(0)
layers=synth_weights(152,seed=0)
net=caffe.Net(deepercut_prototxt(152,544,736),caffe.TEST,from_text=True,dtype="f16")
for name,_t,blobs in layers:
    for p,b in zip(net.params[name],blobs): p.data[...]=b
scales=[0.5,0.75,1.0,1.25]
imgs8=[np.random.RandomState(10+i).randint(0,256,(544,736,3)).astype(np.uint8) for i in range(8)]
def timed(fn,reps=5):
    fn(); fn()
    t=time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter()-t)/reps
for label,depth in (("depth2 first",2),("depth1 second",1),("depth2 third",2),("depth1 fourth",1)):
    r=ShardedPoseRunner(net,max_batch=8,depth=depth)
    dt=timed(lambda: r.run(imgs8,scales))
    print("%-16s %.1f image-pyramids/s" % (label, 8/dt), flush=True)
