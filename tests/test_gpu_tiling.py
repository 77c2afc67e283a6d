"""-m gpu: the exact overlap-tile stitcher (pose.estimate_pose.forward_maps_tiled) through the
HIP forward path: on a DeeperCut-shaped network whose receptive field (~130 px) fits inside the 224-px tile
margin, stitched maps must equal the maps of one un-tiled forward."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _conv(name, bottom, top, nout, k, pad=0, stride=1, dilation=1, bias=False, typ="Convolution"):
    return ('layer { name: "%s" type: "%s" bottom: "%s" top: "%s" convolution_param { num_output: %d kernel_size: %d '
            'pad: %d stride: %d dilation: %d bias_term: %s } }'
            % (name, typ, bottom, top, nout, k, pad, stride, dilation, "true" if bias else "false"))


def _bn_relu(tag, blob, relu=True):
    out = ['layer { name: "bn%s" type: "BatchNorm" bottom: "%s" top: "%s" batch_norm_param { use_global_stats: true } }'
           % (tag, blob, blob),
           'layer { name: "scale%s" type: "Scale" bottom: "%s" top: "%s" scale_param { bias_term: true } }' % (tag, blob, blob)]
    if relu:
        out.append('layer { name: "relu%s" type: "ReLU" bottom: "%s" top: "%s" }' % (tag, blob, blob))
    return out


def local_fcn_prototxt(h, w):
    """stride-2 7x7 stem, ceil-mode pool, two stride-2 1x1 reductions (total stride 16), a dilated 3x3, and the
    DeeperCut head: stride-2 deconvolution, fork Crop, 1x1 skip from the stride-8 level, Eltwise, Sigmoid."""
    L = ['name: "local_fcn"', 'input: "data"'] + ["input_dim: %d" % d for d in (1, 3, h, w)]
    L.append(_conv("conv1", "data", "conv1", 16, 7, pad=3, stride=2))
    L += _bn_relu("_conv1", "conv1")
    L.append('layer { name: "pool1" type: "Pooling" bottom: "conv1" top: "pool1" pooling_param { pool: MAX kernel_size: 3 stride: 2 } }')
    L.append(_conv("c2", "pool1", "c2", 32, 3, pad=1))
    L += _bn_relu("_c2", "c2")
    L.append(_conv("c3", "c2", "c3", 32, 1, stride=2))  # stride 8
    L += _bn_relu("_c3", "c3")
    L.append(_conv("c4", "c3", "c4", 64, 1, stride=2))  # stride 16
    L += _bn_relu("_c4", "c4")
    L.append(_conv("c5", "c4", "c5", 64, 3, pad=2, dilation=2))
    L += _bn_relu("_c5", "c5")
    for suffix, nout, out in (("pose", 14, "fc_pose"), ("locref", 28, "loc_pred")):
        L.append(_conv("up_" + suffix, "c5", "up_" + suffix, nout, 3, stride=2, bias=True, typ="Deconvolution"))
        L.append(_conv("skip_" + suffix, "c3", "skip_" + suffix, nout, 1, bias=True))
        L.append('layer { name: "crop_%s" type: "Crop" bottom: "up_%s" bottom: "skip_%s" top: "crop_%s" }'
                 % (suffix, suffix, suffix, suffix))
        L.append('layer { name: "%s" type: "Eltwise" bottom: "skip_%s" bottom: "crop_%s" top: "%s" }' % (out, suffix, suffix, out))
    L.append('layer { name: "prob" type: "Sigmoid" bottom: "fc_pose" top: "prob" }')
    return "\n".join(L) + "\n"


def _fill(net, seed):
    rs = np.random.RandomState(seed)
    for name, blobs in net.params.items():
        if name.startswith("bn"):
            blobs[0].data[...] = rs.randn(*blobs[0].data.shape) * 0.1
            blobs[1].data[...] = rs.uniform(0.5, 1.5, blobs[1].data.shape)
            blobs[2].data[...] = 1.0
        elif name.startswith("scale"):
            blobs[0].data[...] = rs.uniform(0.5, 1.5, blobs[0].data.shape)
            blobs[1].data[...] = rs.randn(*blobs[1].data.shape) * 0.1
        else:
            w = blobs[0].data
            fan = float(np.prod(w.shape[1:])) if not name.startswith("up_") else float(w.shape[0] * 9 / 4.0)
            w[...] = rs.randn(*w.shape) / np.sqrt(fan)
            if len(blobs) > 1:
                blobs[1].data[...] = rs.randn(*blobs[1].data.shape) * 0.1


@pytest.mark.parametrize("hw", [(720, 960), (1176, 704)])
def test_exact_tiling_equals_one_forward(gpu_caffe, hw):
    from pose import estimate_pose as ep

    h, w = hw
    net = gpu_caffe.Net(local_fcn_prototxt(h, w), gpu_caffe.TEST, from_text=True)
    _fill(net, 3)
    canvas = (np.random.RandomState(11).randn(h, w, 3) * 50).astype(np.float32)
    whole_prob, whole_loc = ep.forward_maps(net, canvas)
    tiles = []

    def fwd(n, tile):
        tiles.append(tile.shape[:2])
        return ep.forward_maps(n, tile)

    prob, loc = ep.forward_maps_tiled(net, canvas, mode="exact", forward=fwd)
    assert len(tiles) == len(ep.tile_spans(h)) * len(ep.tile_spans(w)) > 1
    assert prob.shape == whole_prob.shape == (14, h // 8, w // 8) and loc.shape == whole_loc.shape
    assert float(np.abs(whole_loc).max()) > 0.5  # the comparison is not about zeros
    # same arithmetic per cell; only the tile variant (summation order) may differ between shapes
    assert float(np.abs(prob - whole_prob).max()) <= 1e-4
    assert float(np.abs(loc - whole_loc).max()) <= 1e-4 * max(1.0, float(np.abs(whole_loc).max()))
