"""How much do independent streams help ONE layer type?  Three single-layer nets (clones) launched round-robin on three
streams vs one stream: aggregate launches per second (tools/inflight_probe.py; diagnostics for DESIGN §7b)."""
import sys, time
sys.path.insert(0, "deepcut-cnn_amd/python")
import numpy as np, torch, caffe

caffe.set_mode_gpu()


def layer_net(cin, cout, k, h, w):
    pad = k // 2
    txt = ('name: "l" input: "data" input_dim: 1 input_dim: %d input_dim: %d input_dim: %d\n' % (cin, h, w) +
           'layer { name: "c" type: "Convolution" bottom: "data" top: "prob" convolution_param { num_output: %d kernel_size: %d pad: %d bias_term: false } }\n'
           % (cout, k, pad) + 'layer { name: "r" type: "ReLU" bottom: "prob" top: "prob" }\n')
    net = caffe.Net(txt, caffe.TEST, from_text=True, hipgraph=0)
    net.params["c"][0].data[...] = np.random.RandomState(0).randn(cout, cin, k, k).astype(np.float32) * 0.01
    return net


def rate(nets, xs, outs, streams, iters=300):
    for i in range(30):
        k = i % len(nets)
        nets[k].forward_device(xs[k].data_ptr(), 1, xs[k].shape[2], xs[k].shape[3], outs[k].data_ptr(), None, None, streams[k].cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(iters):
        k = i % len(nets)
        nets[k].forward_device(xs[k].data_ptr(), 1, xs[k].shape[2], xs[k].shape[3], outs[k].data_ptr(), None, None, streams[k].cuda_stream)
    torch.cuda.synchronize()
    return iters / (time.perf_counter() - t)


for name, (cin, cout, k) in {"res4 2a 1x1 1024->256": (1024, 256, 1), "res4 2b 3x3 256->256": (256, 256, 3), "res4 2c 1x1 256->1024": (256, 1024, 1)}.items():
    base = layer_net(cin, cout, k, 34, 46)
    nets = [base, base.clone(), base.clone()]
    xs = [torch.randn(1, cin, 34, 46, device="cuda") for _ in nets]
    outs = [torch.empty(1, cout, 34, 46, device="cuda") for _ in nets]
    streams = [torch.cuda.Stream() for _ in nets]
    r1 = rate(nets[:1], xs, outs, streams)
    r3 = rate(nets, xs, outs, streams)
    print("%-24s 1 stream %7.0f launches/s (%.1f us each incl. 2 layout kernels)   3 streams %7.0f /s   x%.2f" % (name, r1, 1e6 / r1, r3, r3 / r1))
