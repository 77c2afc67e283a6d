"""CPU ORACLE (test infrastructure, never on the product path) for the multi-person consumers of the maps
(SURVEY §8f row 2).

PARITY UNPINNED: the reference repository has no consumer of `next_pred` and no part-candidate extraction (it stops at
the maps) — there is nothing to compare with.  What is restated here is the INVERSE of the label encoding of
the reference's training layer, src/caffe/layers/pose_data_layer.cpp:

  :686   pt = (i*stride + half_stride, j*stride + half_stride) / scale          the image point a map cell (j, i) stands for
  :752-765 loc target k of joint = (joint - pt)[k] * scale / sqrt(53)            -> joint = pt + loc * sqrt(53) / scale
  :768-802 next target (l, k)    = ((next - pt)[k] * scale - mean[l][k]) / std[l][k]
                                                                                 -> next = pt + (v * std + mean) / scale

`nms_candidates` is a plain definition (local maximum of a (2r+1)^2 window, threshold, deterministic tie rule), not a
reference algorithm.  Pure numpy / Python loops: small cases only."""
import numpy as np

STRIDE = 8
LOCREF = np.sqrt(53.0)


def nms_candidates(prob, loc, scale=1.0, threshold=0.1, radius=1, max_det=32):
    """prob [J,h,w], loc [2J,h,w] -> (counts [J], dets [J, max_det, 5] = x, y, score, row, col; unused rows: 0,0,0,-1,-1)."""
    J, h, w = prob.shape
    counts = np.zeros(J, np.int32)
    dets = np.zeros((J, max_det, 5), np.float64)
    dets[:, :, 3:] = -1
    for j in range(J):
        cand = []
        for r in range(h):
            for c in range(w):
                v = prob[j, r, c]
                if not v >= threshold:
                    continue
                ok = True
                for y in range(max(0, r - radius), min(h, r + radius + 1)):
                    for x in range(max(0, c - radius), min(w, c + radius + 1)):
                        if (y, x) == (r, c):
                            continue
                        u = prob[j, y, x]
                        if u > v or (u == v and y * w + x < r * w + c):
                            ok = False
                if ok:
                    cand.append((-float(v), r * w + c))
        cand.sort()
        counts[j] = min(len(cand), max_det)
        for k, (neg, cell) in enumerate(cand[:max_det]):
            r, c = divmod(cell, w)
            dets[j, k] = ((c * STRIDE + 0.5 * STRIDE + float(loc[2 * j, r, c]) * LOCREF) / scale,
                          (r * STRIDE + 0.5 * STRIDE + float(loc[2 * j + 1, r, c]) * LOCREF) / scale, -neg, r, c)
    return counts, dets


def pairwise_positions(next_pred, cells, scale=1.0, mean=None, std=None):
    """next_pred [2E,h,w], cells [(row, col)] -> [D, E, 2] predicted (x, y) of the next joint of every edge."""
    E = next_pred.shape[0] // 2
    mean = np.zeros((E, 2)) if mean is None else np.asarray(mean, np.float64).reshape(E, 2)
    std = np.ones((E, 2)) if std is None else np.asarray(std, np.float64).reshape(E, 2)
    out = np.zeros((len(cells), E, 2), np.float64)
    for d, (r, c) in enumerate(cells):
        pt = np.array([c * STRIDE + 0.5 * STRIDE, r * STRIDE + 0.5 * STRIDE], np.float64)
        for l in range(E):
            v = np.array([float(next_pred[2 * l, r, c]), float(next_pred[2 * l + 1, r, c])])
            out[d, l] = (pt + v * std[l] + mean[l]) / scale
    return out


def encode_targets(joint_xy, next_xy, cell, scale, mean, std):
    """The reference's ENCODING for one cell (pose_data_layer.cpp:686,752-802), used by the tests to check that the
    decoders above really invert it: -> (loc target [2], next target [2])."""
    r, c = cell
    pt = np.array([c * STRIDE + 0.5 * STRIDE, r * STRIDE + 0.5 * STRIDE], np.float64) * (1.0 / scale)
    loc = (np.asarray(joint_xy, np.float64) - pt) * scale / LOCREF
    nxt = ((np.asarray(next_xy, np.float64) - pt) * scale - np.asarray(mean, np.float64)) / np.asarray(std, np.float64)
    return loc, nxt
