import os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+"/deepcut-cnn_amd", ROOT+"/deepcut-cnn_amd/python"): sys.path.insert(0,p)
import numpy as np, torch
import caffe
from deepcut_tools import deepercut_prototxt, synth_weights
caffe.set_mode_gpu(); caffe.set_device(0)
H,W=544,736
layers=synth_weights(152,seed=0)
net=caffe.Net(deepercut_prototxt(152,H,W,1),caffe.TEST,from_text=True,hipgraph=1)
for name,_t,blobs in layers:
    for p,b in zip(net.params[name],blobs): p.data[...]=b
xh=(np.random.RandomState(0).randn(1,3,H,W)*50).astype(np.float32)
def full():
    net.blobs["data"].data[...]=xh; net.forward(); return [net.blobs[k].data for k in ("prob","loc_pred","next_pred")]
for _ in range(5): full()
def t(label,fn,n=30):
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); fn(); ts.append((time.perf_counter()-t0)*1e3)
    ts.sort(); print("%-50s med %.3f min %.3f ms"%(label,ts[len(ts)//2],ts[0]),flush=True)
t("full sequence", full)
d=net.blobs["data"]
t("blobs['data'].data property only", lambda: d.data)
def w(): d.data[...]=xh
t("data[...]=xh (host write 4.8 MB)", w)
nl=len(net._layer_names)
def f():
    d.data  # mark host authoritative
    net._forward(0,nl-1)
t("mark + C forward (H2D + compute + 3 maps delivered)", f)
def f2():
    net._forward(0,nl-1)
t("C forward, input already on device (+ delivery)", f2)
t("python forward() incl. building the dict", lambda: (d.data, net.forward()))
t("reading 3 outputs .data", lambda: [net.blobs[k].data for k in ("prob","loc_pred","next_pred")])
xd=torch.from_numpy(xh).cuda(); o=[torch.empty(net.blobs[k].shape,device="cuda") for k in ("prob","loc_pred","next_pred")]
def dev():
    net.forward_device(xd.data_ptr(),1,H,W,o[0].data_ptr(),o[1].data_ptr(),o[2].data_ptr()); torch.cuda.synchronize()
t("forward_device + sync", dev)
