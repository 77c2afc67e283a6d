"""Mutated model definitions through net construction on the CPU (no device needed: parsing, InsertSplits, shape inference,
lowering are host code): numbers replaced by extremes (0, -1, 2^31, 1e30, nan ...), bottoms / tops rewired, lines dropped or
doubled, layer types swapped — every mutant must either build (and survive a reshape to another input size) or be refused
with an error; the reference CHECK-fails (aborts) on such input (net.cpp:405,450; base_conv_layer.cpp:22-90), this library
must never crash, hang or silently continue.  usage: fuzz_net_definitions.py <rng seed> <iterations>   (run in a process of
its own by tests/test_net_fuzz.py: a crash is a finding, not a test error)"""
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deepcut-cnn_amd"), os.path.join(ROOT, "deepcut-cnn_amd", "python")):
    sys.path.insert(0, p)
import caffe  # noqa: E402
from deepcut_tools import deepercut_prototxt  # noqa: E402

seed, iters = int(sys.argv[1]), int(sys.argv[2])
rnd = random.Random(seed)
HEAD = deepercut_prototxt(50, 64, 80)[:16000]
HEAD = HEAD[:HEAD.rindex("\nlayer {")] + "\n"  # conv1 ... the first residual blocks of the real definition
SMALL = '''name: "n" input: "data" input_dim: 1 input_dim: 3 input_dim: 33 input_dim: 41
layer { name: "c1" type: "Convolution" bottom: "data" top: "c1" convolution_param { num_output: 16 kernel_size: 7 pad: 3 stride: 2 bias_term: false } }
layer { name: "bn1" type: "BatchNorm" bottom: "c1" top: "c1" batch_norm_param { use_global_stats: true } }
layer { name: "sc1" type: "Scale" bottom: "c1" top: "c1" scale_param { bias_term: true } }
layer { name: "r1" type: "ReLU" bottom: "c1" top: "c1" }
layer { name: "p1" type: "Pooling" bottom: "c1" top: "p1" pooling_param { kernel_size: 3 stride: 2 pool: MAX } }
layer { name: "c2" type: "Convolution" bottom: "p1" top: "c2" convolution_param { num_output: 16 kernel_size: 3 pad: 2 dilation: 2 } }
layer { name: "e" type: "Eltwise" bottom: "p1" bottom: "c2" top: "e" }
layer { name: "up" type: "Deconvolution" bottom: "e" top: "up" convolution_param { num_output: 4 kernel_size: 3 stride: 2 } }
layer { name: "sk" type: "Convolution" bottom: "c1" top: "sk" convolution_param { num_output: 4 kernel_size: 1 } }
layer { name: "cr" type: "Crop" bottom: "up" bottom: "sk" top: "cr" crop_param { axis: 2 } }
layer { name: "s" type: "Eltwise" bottom: "sk" bottom: "cr" top: "s" }
layer { name: "p" type: "Sigmoid" bottom: "s" top: "p" }
'''
EXTREMES = ["0", "-1", "1", "2", "3", "65536", "2147483647", "2147483648", "4294967296", "-2147483648", "1e30", "nan", "0.5", "99999999999999999999"]
TYPES = ["Convolution", "Pooling", "Eltwise", "Crop", "Deconvolution", "Scale", "BatchNorm", "Sigmoid", "ReLU", "Split", "Nope"]
NUMBER = re.compile(r"(?<=: )-?\d+")


def mutate(t):
    for _ in range(rnd.randint(1, 3)):
        k = rnd.random()
        nums = list(NUMBER.finditer(t))
        if k < 0.6 and nums:  # a number becomes an extreme
            m = rnd.choice(nums)
            t = t[:m.start()] + rnd.choice(EXTREMES) + t[m.end():]
        elif k < 0.75:  # a blob name is rewired
            names = re.findall(r'(?:bottom|top): "(\w+)"', t)
            a, b = rnd.choice(names), rnd.choice(names)
            t = t.replace('"%s"' % a, '"%s"' % b, 1)
        elif k < 0.9:  # a line dropped or doubled
            lines = t.split("\n")
            i = rnd.randrange(len(lines))
            if rnd.random() < 0.5:
                del lines[i]
            else:
                lines.insert(i, lines[rnd.randrange(len(lines))])
            t = "\n".join(lines)
        else:  # a layer type swapped
            t = t.replace(rnd.choice(TYPES[:7]), rnd.choice(TYPES), 1)
    return t


built = refused = 0
for i in range(iters):
    text = mutate(SMALL if rnd.random() < 0.7 else HEAD)
    try:
        net = caffe.Net(text, caffe.TEST, from_text=True)
        shapes = [tuple(b.shape) for b in net.blobs.values()]
        assert all(d >= 0 for s in shapes for d in s), shapes
        net.blobs[net.inputs[0]].reshape(1, 3, rnd.choice([1, 8, 33, 100]), rnd.choice([1, 9, 41, 100]))
        try:
            net.reshape()
        except caffe.DeepcutError:
            pass  # (the other input size does not fit the mutated geometry)
        built += 1
        del net
    except (caffe.DeepcutError, RuntimeError, ValueError, KeyError, IndexError):
        refused += 1
print("seed %d: %d built, %d refused" % (seed, built, refused))
