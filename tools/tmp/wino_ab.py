import sys, time
sys.path.insert(0, "deepcut-cnn_amd/python")
import numpy as np, torch
sys.path.insert(0, ".")
import caffe
from deepcut_tools import deepercut_prototxt, synth_weights
sys.path.insert(0, ".")
from bench import inject_weights
caffe.set_mode_gpu(); caffe.set_device(0)
H, W = 544, 736
layers = synth_weights(152, seed=0)
net = caffe.Net(deepercut_prototxt(152, H, W, 1), caffe.TEST, from_text=True, hipgraph=1)
inject_weights(net, layers)
x = (torch.randn(1, 3, H, W) * 50).cuda()
outs = [torch.empty(net.blobs[k].shape, device="cuda") for k in ("prob", "loc_pred", "next_pred")]
st = torch.cuda.Stream()
def fwd(n):
    for _ in range(n):
        net.forward_device(x.data_ptr(), 1, H, W, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
def timed(n=60):
    fwd(10)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); fwd(n); best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e3
fwd(3)
rep = net.tune_report()
wsig = [r for r in rep if r["signature"].endswith("+w")]
for r in wsig:
    print(r["signature"], r["tile"], r["launches"], [(t, round(us, 2)) for t, us in r["timed"][:5]])
base = timed()
print("as tuned: %.3f ms per forward (%.1f images/s)" % (base, 1e3 / base))
for form in ("wino_f23_w16", "wino_f23"):
    for r in wsig:
        net.set_tile(r["signature"], form)
    t = timed()
    print("every Winograd signature on %s: %.3f ms (%.1f images/s)" % (form, t, 1e3 / t))
# per signature: w16 only there
for r in wsig:
    for q in wsig:
        net.set_tile(q["signature"], "wino_f23")
    net.set_tile(r["signature"], "wino_f23_w16")
    t = timed()
    print("w16 only on %s (%d launches): %.3f ms (%.1f images/s)" % (r["signature"], r["launches"], t, 1e3 / t))
