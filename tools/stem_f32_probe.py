import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import numpy as np
import caffe
caffe.set_mode_gpu(); caffe.set_device(0)
L = ['name: "s"', 'input: "data"'] + ["input_dim: %d" % d for d in (1, 3, 544, 736)]
L.append('layer { name: "conv1" type: "Convolution" bottom: "data" top: "conv1" convolution_param { num_output: 64 kernel_size: 7 pad: 3 stride: 2 bias_term: false } }')
L.append('layer { name: "bn" type: "BatchNorm" bottom: "conv1" top: "conv1" batch_norm_param { use_global_stats: true } }')
L.append('layer { name: "scale" type: "Scale" bottom: "conv1" top: "conv1" scale_param { bias_term: true } }')
L.append('layer { name: "relu" type: "ReLU" bottom: "conv1" top: "conv1" }')
net = caffe.Net("\n".join(L) + "\n", caffe.TEST, from_text=True)
rs = np.random.RandomState(0)
net.params["conv1"][0].data[...] = rs.randn(64, 3, 7, 7) / 12.0
net.params["bn"][1].data[...] = 1.0; net.params["bn"][2].data[...] = 1.0; net.params["scale"][0].data[...] = 1.0
net.blobs["data"].data[...] = rs.randn(1, 3, 544, 736) * 50
net.forward()
for e in net.tune_report():
    print(e["signature"], "chosen", e["tile"], " ".join("%s %.2f" % t for t in sorted(e["timed"], key=lambda t: t[1])[:6]))
if os.environ.get("DC_DEBUG_TIMING") is None:
    print(net.plan_text())
