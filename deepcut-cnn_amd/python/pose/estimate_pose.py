"""Single-person pose estimation on top of the MI355X DeeperCut forward path.

Same entry point and return layout as the reference's python/pose/estimate_pose.py
(`estimate_pose(image, model_def, model_bin, scales)` -> 5x14 array: x, y, confidence and the two
location-refinement components per joint), written against the `caffe` shim of this package
(libdeepcut_hip.so).  What is kept from the reference, with the lines it mirrors:

* pre-processing (estimate_pose.py:83-103): replicate the last row / column 64 px, rescale the uint8
  image bilinearly (scipy.misc.imresize == PIL resize to int(dim*scale); identity at scale 1), subtract
  the BGR mean [104, 117, 123], paste into a zero canvas whose sides are rounded up to the stride 8;
* decoding (:131-143): per joint the arg-max cell of the score map, position = cell*8 + 4 + the
  location-refinement vector at that cell * sqrt(53), everything divided by the scale; rows 3 and 4 hold
  the refinement vector in the reference's (row-offset, column-offset) order;
* scale selection (:119-126): the scale whose minimum joint confidence is highest (strict >, start 0).

Tiling (:146-221, 245-259).  The reference cuts inputs wider than 700 px into 700-px tiles overlapping by
2 x 224 px "to fit GPU memory", with a port of 1-based MATLAB indexing: tile 0 is trimmed on both sides,
tile 1 only at its end, the last tile is never recognised, rows are trimmed twice, the tile step (252 px)
is not a whole number of 8-px map cells, and `rf / stride` is a float on Python 3 (TypeError) — SURVEY F7.
Three behaviours are offered through `tiling=`:

* None (default): no tiling.  288 GB of HBM hold any image; the whole canvas is one forward.
* "exact": a correct overlap-tile stitcher (`tile_spans`): 688-px tiles (the largest multiple of the
  network's stride 16 below 700) every 240 px, 28 map cells dropped on each *interior* seam only, so the
  stitched map has exactly the un-tiled map's shape and, wherever the receptive field fits in the 224-px
  margin, its values.
* "reference": the reference's arithmetic as it ran under Python 2 (integer `rf / stride`), bugs
  included, for callers that must reproduce its output; pinned by tests/golden/tiling_golden.npz, which
  was produced by the reference's own `_process_image_tiled`.
"""
import logging as _logging

import numpy as _np

_LOGGER = _logging.getLogger(__name__)

MEAN_BGR = _np.array([104.0, 117.0, 123.0])
LOCREF_SCALE = _np.sqrt(53.0)
STRIDE = 8
PAD = 64
NUM_JOINTS = 14

_MODEL = {}


def num_tiles(length, max_size=700, rf=224):
    """Tiles the reference would cut a side of `length` px into (estimate_pose.py:146-156)."""
    if length <= max_size:
        return 1
    k = 0
    while (max_size - rf) * 2 + (max_size - 2 * rf) * k <= length:
        k += 1
    return 2 + k


def tile_spans(length, max_size=700, rf=224, stride=STRIDE):
    """Exact tiling of one side of `length` px (a multiple of `stride`): a list of
    (start_px, end_px, keep_lo, keep_hi) with keep_* in map cells relative to the tile.  Tiles are
    tile = max_size rounded down to a multiple of 16 px long (the network's deepest stride, so every
    tile samples the same pixel grid as the whole image) and start every tile - 2*rf px; rf/stride cells
    are dropped next to interior seams only, and the kept cell ranges partition [0, length/stride)."""
    if rf % 16 or length % stride:
        raise ValueError("rf must be a multiple of 16 and length a multiple of %d" % stride)
    if length <= max_size:
        return [(0, length, 0, length // stride)]
    tile = max_size // 16 * 16
    step = tile - 2 * rf
    if step <= 0:
        raise ValueError("max_size %d leaves no interior between two %d-px margins" % (max_size, rf))
    cut = rf // stride
    spans, start = [], 0
    while True:
        end = min(start + tile, length)
        last = end == length
        cells = (end - start) // stride
        spans.append((start, end, cut if start else 0, cells if last else cells - cut))
        if last:
            return spans
        start += step


def _reference_cut(n_tiles, idx, cut):
    """The slice `_cutoff_tile` (estimate_pose.py:245-259) applies for tile `idx` of `n_tiles`: the
    comparisons are 1-based while the caller counts from 0, so tile 0 loses both margins, tile 1 its
    trailing margin and `idx == n_tiles` never happens."""
    if n_tiles == 1:
        return slice(None)
    if idx == 1:
        return slice(None, -cut)
    return slice(cut, -cut)


def forward_maps_tiled(net, net_input, max_size=700, rf=224, mode="exact", forward=None):
    """HxWx3 float32 canvas -> (prob [14,h,w], loc_pred [28,h,w]) from per-tile forwards.
    mode "exact": see `tile_spans`; mode "reference": the reference's stitching (module docstring).
    `forward(net, tile)` defaults to `forward_maps`."""
    forward = forward or forward_maps
    h, w = net_input.shape[:2]
    if mode == "exact":
        rows = []
        for y0, y1, ky0, ky1 in tile_spans(h, max_size, rf):
            line = []
            for x0, x1, kx0, kx1 in tile_spans(w, max_size, rf):
                prob, loc = forward(net, net_input[y0:y1, x0:x1])
                line.append((prob[:, ky0:ky1, kx0:kx1], loc[:, ky0:ky1, kx0:kx1]))
            rows.append((_np.concatenate([t[0] for t in line], axis=2),
                         _np.concatenate([t[1] for t in line], axis=2)))
        return (_np.concatenate([r[0] for r in rows], axis=1), _np.concatenate([r[1] for r in rows], axis=1))
    if mode != "reference":
        raise ValueError("tiling mode must be 'exact' or 'reference', not %r" % (mode,))
    cut = rf // STRIDE  # Python 2's `rf / stride`
    nx, ny = num_tiles(w, max_size, rf), num_tiles(h, max_size, rf)
    step = max_size - 2 * rf
    rows = []
    for j in range(ny):
        sy = _reference_cut(ny, j, cut)
        line = []
        for i in range(nx):
            sx = _reference_cut(nx, i, cut)
            prob, loc = forward(net, net_input[j * step:j * step + max_size, i * step:i * step + max_size])
            line.append((prob[:, :, sx][:, sy], loc[:, :, sx][:, sy]))
        # the reference trims the assembled line in y a second time (:208-211)
        rows.append((_np.concatenate([t[0] for t in line], axis=2)[:, sy],
                     _np.concatenate([t[1] for t in line], axis=2)[:, sy]))
    return (_np.concatenate([r[0] for r in rows], axis=1), _np.concatenate([r[1] for r in rows], axis=1))


def _resize_bilinear_u8(image, scale):
    """scipy.misc.imresize(image, scale, interp='bilinear') for a uint8 HxWx3 image: PIL resize to
    (int(W*scale), int(H*scale)); the identity when scale == 1."""
    if scale == 1.0:
        return image
    from PIL import Image

    h, w = image.shape[:2]
    size = (int(w * scale), int(h * scale))
    return _np.asarray(Image.fromarray(image).resize(size, Image.BILINEAR))


def preprocess(image, scale):
    """HxWx3 BGR uint8 -> float32 net input HxWx3 (mean-subtracted, zero-padded to a multiple of 8)."""
    image = _np.asarray(image)
    h, w = image.shape[:2]
    out_h = int(_np.ceil(float(h) * scale / STRIDE) * STRIDE)
    out_w = int(_np.ceil(float(w) * scale / STRIDE) * STRIDE)
    padded = _np.pad(image, ((0, PAD), (0, PAD), (0, 0)), mode="edge")
    scaled = _resize_bilinear_u8(padded, scale).astype(_np.float32) - MEAN_BGR.astype(_np.float32)
    canvas = _np.zeros((out_h, out_w, 3), _np.float32)
    hh, ww = min(out_h, scaled.shape[0]), min(out_w, scaled.shape[1])
    canvas[:hh, :ww] = scaled[:hh, :ww]
    return canvas


def pose_from_maps(prob, loc_pred, scale=1.0):
    """prob [14,h,w], loc_pred [28,h,w] (channel 2j / 2j+1 = x / y refinement of joint j in units of
    sqrt(53) px) -> 5x14 float64, exactly the reference's `_pose_from_mats` arithmetic."""
    prob = _np.asarray(prob)
    loc = _np.asarray(loc_pred)
    j, h, w = prob.shape
    flat = prob.reshape(j, -1)
    idx = flat.argmax(axis=1)  # first maximum in row-major order, as numpy.argmax on the 2-D map
    rows, cols = _np.divmod(idx, w)
    jj = _np.arange(j)
    conf = flat[jj, idx]
    off_x = loc[2 * jj, rows, cols].astype(_np.float64)
    off_y = loc[2 * jj + 1, rows, cols].astype(_np.float64)
    pose = _np.empty((5, j), _np.float64)
    pose[0] = (cols.astype(_np.float64) * STRIDE + 0.5 * STRIDE + off_x * LOCREF_SCALE) / scale
    pose[1] = (rows.astype(_np.float64) * STRIDE + 0.5 * STRIDE + off_y * LOCREF_SCALE) / scale
    pose[2] = conf
    pose[3] = off_y * LOCREF_SCALE / scale
    pose[4] = off_x * LOCREF_SCALE / scale
    return pose


def pose_cells(pose, scale=1.0):
    """(rows, cols) of the arg-max cells a 5xJ pose was decoded from (the inverse of `pose_from_maps`' position rule)."""
    pose = _np.asarray(pose, _np.float64)
    cols = _np.rint(((pose[0] - pose[4]) * scale - 0.5 * STRIDE) / STRIDE).astype(int)
    rows = _np.rint(((pose[1] - pose[3]) * scale - 0.5 * STRIDE) / STRIDE).astype(int)
    return rows, cols


def select_best(poses):
    """Keep the pose whose minimum joint confidence is highest (strict >, initial 0): None if no scale
    reaches a positive minimum, like the reference."""
    best, best_conf = None, 0.0
    for p in poses:
        c = float(p[2].min())
        if c > best_conf:
            best, best_conf = p, c
    return best


def _get_model(model_def, model_bin):
    import caffe as _caffe

    key = (model_def, model_bin)
    if key not in _MODEL:
        _LOGGER.info("Loading pose model...")
        _MODEL[key] = _caffe.Net(model_def, model_bin, _caffe.TEST)
    return _MODEL[key]


def forward_maps(net, net_input):
    """HxWx3 float32 -> (prob [14,h,w], loc_pred [28,h,w]) through net.forward()."""
    chw = _np.ascontiguousarray(net_input.transpose((2, 0, 1)), dtype=_np.float32)
    net.blobs["data"].reshape(1, 3, chw.shape[1], chw.shape[2])
    net.blobs["data"].data[0, ...] = chw
    net.forward()
    return net.blobs["prob"].data[0].copy(), net.blobs["loc_pred"].data[0].copy()


def _scale_group(net, n):
    """The grouped executor of an n-scale pyramid for `net`: the net itself plus n - 1 clones.  The clones are kept with
    the net and shared by the groups of every n (a 2-scale and a 4-scale call use the same first clone), and at most
    `_MAX_SCALE_GROUPS` groups are kept per net (least recently used first out), so neither grows with the number of
    distinct len(scales) a caller goes through."""
    import caffe as _caffe

    groups = net.__dict__.setdefault("_scale_groups", {})
    grp = groups.pop(n, None)
    if grp is None:
        clones = net.__dict__.setdefault("_scale_clones", [])
        while len(clones) < n - 1:
            clones.append(net.clone())
        grp = _caffe.NetGroup([net] + clones[: n - 1])
        while len(groups) >= _MAX_SCALE_GROUPS:
            groups.pop(next(iter(groups)))
    groups[n] = grp  # most recently used last
    return grp


_MAX_SCALE_GROUPS = 4


def _read_outputs_only(net):
    """The demo reads `prob` and `loc_pred` (estimate_pose.py:231-241 of the reference) and never `next_pred`: tell the net once
    (DC_OPT_OUTPUTS) — without the 364-channel pairwise head the merged heads shrink from 406 to 42 channels, 23.3 GFLOP less per
    544x736 forward; the two maps equal the full forward's (bit for bit under the same head tile).  `net.set_outputs(None)` brings every output back."""
    for n in [net] + list(getattr(net, "__dict__", {}).get("_scale_clones", [])):  # (the scale-group clones follow their net)
        if hasattr(n, "set_outputs") and "next_pred" in n.wanted_outputs and {"prob", "loc_pred"} <= set(n.outputs):
            n.set_outputs(["loc_pred", "prob"])


def estimate_pose(image, model_def, model_bin, scales=None, net=None, tiling=None, on_device=True, grouped=None, all_outputs=False):
    """image: HxWx3 BGR uint8.  Returns the 5x14 pose of the best scale (see module docstring).
    tiling: None (one forward per scale), "exact" or "reference" (see `forward_maps_tiled`).
    on_device: without tiling, pre-process and decode on the GPU (`Net.forward_images`: the same canvas bit
    for bit, the same forward, 70 doubles back instead of the maps); False keeps every step where the
    reference has it (Pillow + NumPy on the host around `net.forward()`).
    grouped: None = several scales on the device run as ONE grouped forward (`caffe.NetGroup`); False = the
    reference's loop, one forward per scale (bit-identical to the host route's forwards).
    all_outputs: False (default) = the net computes only what is read here, `prob` and `loc_pred` (see `_read_outputs_only`); True
    leaves the net's output selection alone (a caller that also reads `net.blobs['next_pred']` afterwards)."""
    if scales is None:
        scales = [1.0]
    if net is None:
        net = _get_model(model_def, model_bin)
    if not all_outputs:
        _read_outputs_only(net)
    poses = []
    if (grouped is None or grouped) and tiling is None and on_device and len(scales) > 1 and hasattr(net, "clone") \
            and _np.asarray(image).dtype == _np.uint8:
        # the scale loop of the reference (:81-128) as ONE grouped forward: a member per scale (the net and clones of it, kept
        # with the net), every layer a single launch over all the scales (caffe.NetGroup / dc_group_forward_images)
        import caffe as _caffe

        if hasattr(_caffe, "NetGroup"):
            outs = _scale_group(net, len(scales)).forward_images(_np.asarray(image), list(scales), want=(), pose=True)
            return select_best([o["pose"][0] for o in outs])
    for s in scales:
        if tiling is None and on_device and hasattr(net, "forward_images") and _np.asarray(image).dtype == _np.uint8:
            poses.append(net.forward_images(_np.asarray(image), s, want=(), pose=True)["pose"][0])
            continue
        if tiling is None:
            prob, loc = forward_maps(net, preprocess(image, s))
        else:
            prob, loc = forward_maps_tiled(net, preprocess(image, s), mode=tiling)
        poses.append(pose_from_maps(prob, loc, s))
    return select_best(poses)
