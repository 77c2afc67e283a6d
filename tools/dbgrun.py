"""Diagnostics: run two 544x736 forwards of the synthetic-weight net (used with DC_DEBUG_TIMING=<launch index> to
print device-side phase timestamps of one launch)."""
import sys, os
sys.path.insert(0,'deepcut-cnn_amd/python'); sys.path.insert(0,'.')
import numpy as np, caffe
from deepcut_tools import deepercut_prototxt, synth_weights
caffe.set_mode_gpu()
net=caffe.Net(deepercut_prototxt(152,544,736),caffe.TEST,from_text=True)
for name,_t,blobs in synth_weights(152,0):
    for p,b in zip(net.params[name],blobs): p.data[...]=b
net.blobs['data'].data[...]=(np.random.RandomState(0).randn(1,3,544,736)*50).astype(np.float32)
net.forward(); net.forward()
