import sys
sys.path.insert(0, "deepcut-cnn_amd/python"); sys.path.insert(0, ".")
import torch
import caffe
from deepcut_tools import deepercut_prototxt, synth_weights
from bench import inject_weights
caffe.set_mode_gpu(); caffe.set_device(0)
H, W = 544, 736
net = caffe.Net(deepercut_prototxt(152, H, W, 1), caffe.TEST, from_text=True, hipgraph=1)
inject_weights(net, synth_weights(152, seed=0))
x = (torch.randn(1, 3, H, W) * 50).cuda()
net.forward_device(x.data_ptr(), 1, H, W, None, None, None, 0)
print([(r["signature"].split("/")[0:2], r["tile"]) for r in net.tune_report() if r["signature"].endswith("+w")])
