"""Generator of the DeeperCut part-detector model definition.

`deepercut_prototxt(depth=152)` emits a prototxt that is layer-for-layer equivalent (same layer
names, types, connectivity and parameters, hence the same blob names and the same weight-file
compatibility) to the reference's models/deepercut/ResNet-152.prototxt: ResNet-152 trunk with the
MSRA stride placement (stride 2 on branch1/branch2a), conv5 at stride 1 with dilation-2 3x3
convolutions, and three deconvolution heads (part score maps, location refinement, pairwise
regression) tied to res3's last block through 1x1 skip convolutions.
tests/test_formats.py (test_generated_model_is_equivalent_to_the_reference_prototxt) checks the equivalence against the
reference file when it is present (the build container; the file does not travel to the GPU box).
`depth=101` gives the ResNet-101 variant BASELINE.json names (res3b1..b3 / res4b1..b22).
"""

_BLOCKS = {152: (3, 8, 36, 3), 101: (3, 4, 23, 3), 50: (3, 4, 6, 3)}


def _block_names(stage, count, letters):
    if letters:  # res2/res5 and ResNet-50 style: a, b, c, ...
        return ["%d%s" % (stage, chr(ord("a") + i)) for i in range(count)]
    return ["%da" % stage] + ["%db%d" % (stage, i) for i in range(1, count)]


def deepercut_layer_table(depth=152, num_joints=14, num_pairs=None):
    """Return the model as a list of dict(name,type,bottoms,tops,params) — the single source both the
    prototxt text and the tests' structural checks are generated from."""
    if depth not in _BLOCKS:
        raise ValueError("depth must be one of %s" % sorted(_BLOCKS))
    if num_pairs is None:
        num_pairs = num_joints * (num_joints - 1)  # ordered joint pairs (182 for 14 joints)
    L = []

    def conv(name, bottom, top, nout, k, pad, stride, bias=False, dilation=1, typ="Convolution"):
        p = {"num_output": nout, "kernel_size": k, "pad": pad, "stride": stride}
        if dilation != 1:
            p["dilation"] = dilation
        if not bias:
            p["bias_term"] = False
        L.append(dict(name=name, type=typ, bottoms=[bottom], tops=[top], conv=p))

    def bn_scale(tag, blob):
        L.append(dict(name="bn" + tag, type="BatchNorm", bottoms=[blob], tops=[blob], bn=True))
        L.append(dict(name="scale" + tag, type="Scale", bottoms=[blob], tops=[blob], scale=True))

    def relu(name, blob):
        L.append(dict(name=name, type="ReLU", bottoms=[blob], tops=[blob]))

    conv("conv1", "data", "conv1", 64, 7, 3, 2)
    bn_scale("_conv1", "conv1")
    relu("conv1_relu", "conv1")
    L.append(dict(name="pool1", type="Pooling", bottoms=["conv1"], tops=["pool1"], pool=dict(kernel_size=3, stride=2)))

    prev = "pool1"
    counts = _BLOCKS[depth]
    last_res3 = None
    for si, stage in enumerate((2, 3, 4, 5)):
        width = 64 << si
        letters = stage in (2, 5) or depth == 50
        for bi, tag in enumerate(_block_names(stage, counts[si], letters)):
            first = bi == 0
            stride = 2 if (first and stage in (3, 4)) else 1  # conv5 stays at 1/16 (atrous)
            dil = 2 if stage == 5 else 1
            shortcut = prev
            if first:
                conv("res%s_branch1" % tag, prev, "res%s_branch1" % tag, width * 4, 1, 0, stride)
                bn_scale("%s_branch1" % tag, "res%s_branch1" % tag)
                shortcut = "res%s_branch1" % tag
            a, b, c = ("res%s_branch2%s" % (tag, x) for x in "abc")
            conv(a, prev, a, width, 1, 0, stride)
            bn_scale("%s_branch2a" % tag, a)
            relu(a + "_relu", a)
            conv(b, a, b, width, 3, dil, 1, dilation=dil)
            bn_scale("%s_branch2b" % tag, b)
            relu(b + "_relu", b)
            conv(c, b, c, width * 4, 1, 0, 1)
            bn_scale("%s_branch2c" % tag, c)
            out = "res%s" % tag
            L.append(dict(name=out, type="Eltwise", bottoms=[shortcut, c], tops=[out]))
            relu(out + "_relu", out)
            prev = out
        if stage == 3:
            last_res3 = prev
    top = prev  # res5c

    def head(suffix, nout, crop_name, out_name, sigmoid_to=None):
        up = "%s_up_%s" % (top, suffix)
        skip = "res3d_%s" % suffix
        conv(up, top, up, nout, 3, 0, 2, bias=True, typ="Deconvolution")
        conv(skip, last_res3, skip, nout, 1, 0, 1, bias=True)
        L.append(dict(name=crop_name, type="Crop", bottoms=[up, skip], tops=[up + "c"]))
        L.append(dict(name=out_name, type="Eltwise", bottoms=[skip, up + "c"], tops=[out_name]))
        if sigmoid_to:
            L.append(dict(name=sigmoid_to, type="Sigmoid", bottoms=[out_name], tops=[sigmoid_to]))

    head("pose", num_joints, "crop1", "fc_pose", sigmoid_to="prob")
    head("locref", 2 * num_joints, "crop_locref", "loc_pred")
    head("next", 2 * num_pairs, "crop_next", "next_pred")
    return L


def deepercut_prototxt(depth=152, height=688, width=688, batch=1, num_joints=14, num_pairs=None):
    """Text prototxt of the DeeperCut net (TEST-phase deploy form: bare `input: "data"`)."""
    out = ['name: "ResNet-%d"' % depth, 'input: "data"']
    out += ["input_dim: %d" % d for d in (batch, 3, height, width)]
    for l in deepercut_layer_table(depth, num_joints, num_pairs):
        f = ['name: "%s"' % l["name"], 'type: "%s"' % l["type"]]
        f += ['bottom: "%s"' % b for b in l["bottoms"]]
        f += ['top: "%s"' % t for t in l["tops"]]
        if "conv" in l:
            p = l["conv"]
            body = " ".join("%s: %s" % (k, str(v).lower() if isinstance(v, bool) else v) for k, v in p.items())
            f.append("convolution_param { %s }" % body)
        if "bn" in l:
            f += ["param { lr_mult: 0 }"] * 3
            f.append("batch_norm_param { use_global_stats: true }")
        if "scale" in l:
            f.append("scale_param { bias_term: true }")
        if "pool" in l:
            f.append("pooling_param { pool: MAX kernel_size: %(kernel_size)d stride: %(stride)d }" % l["pool"])
        out.append("layer { %s }" % " ".join(f))
    return "\n".join(out) + "\n"
