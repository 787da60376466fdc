"""Label rasterisation on the MI355X: ``contours2labels`` of the reference's ``cd.data`` (celldetection/data/cpn.py:292-358,
called from celldetection_scripts/cpn_inference.py:811), backed by ``csrc/labels.hip``.

Same result as the reference's sequential loop (contour i -> value i + 1 in the first channel whose gap-expanded
bounding-box region is still empty), computed in parallel rounds over independent contours.  The polygon fill restates
OpenCV's ``drawContours(thickness=-1)`` rule for integer vertices (cv2 is absent from the build image: see
``oracle/labels_oracle.py`` -- parity with cv2 itself is unpinned).  Everything around the fill -- ``sort_by``, rounding,
clipping, ``ioa_thresh`` / ``return_indices``, the gap rule, channels, numbering -- is checked against outputs of the imported
reference loop (``tests/golden/labels.npz``).
"""
from ctypes import c_int32

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

__all__ = ['contours2labels']


def contours2labels(contours, size, rounded=True, clip=True, initial_depth=1, gap=3, dtype='int32', ioa_thresh=None,
                    sort_by=None, sort_descending=True, return_indices=False, return_stats=False):
    """Contours [K, S, 2] (xy, one image; Tensor on the GPU, or a list of arrays) -> label image Tensor[H, W, channels]
    int32 on the GPU.  Arguments as in the reference (data/cpn.py:292-358).  ``ioa_thresh``: contours whose own area is
    covered by earlier labels to more than that fraction are skipped (:341-350) and the painted ones numbered consecutively;
    ``return_indices``: additionally the list of kept contour positions -- like the reference, filled only when
    ``ioa_thresh`` is given (:356-357)."""
    if np.dtype(dtype) != np.int32:
        raise NotImplementedError("contours2labels on the HIP path produces dtype 'int32' (the reference's default)")
    if not isinstance(contours, torch.Tensor):
        # List[Array[num_points, 2]] with different lengths: pad by repeating the last point (a zero-length edge draws the
        # same pixel again and takes no part in the scanline fill -> identical raster)
        arrs = [np.asarray(c.detach().cpu() if isinstance(c, torch.Tensor) else c, np.float32).reshape(-1, 2) for c in contours]
        smax = max([len(a) for a in arrs] + [1])
        arrs = [np.concatenate((a, np.repeat(a[-1:], smax - len(a), 0))) if 0 < len(a) < smax else a for a in arrs]
        if any(len(a) == 0 for a in arrs):
            # the reference fails on an empty contour as well (np.min of an empty array in render_contour, data/cpn.py:248);
            # dropping it would silently shift the label values and the returned indices against the caller's list
            raise ValueError('contours2labels: zero-length contour at position '
                             f'{[i for i, a in enumerate(arrs) if len(a) == 0][0]}')
        contours = torch.as_tensor(np.stack(arrs) if arrs else np.zeros((0, 1, 2), np.float32))
        contours = contours.cuda() if torch.cuda.is_available() else contours
    if not contours.is_cuda:
        raise RuntimeError('celldetection_amd.contours2labels runs on the MI355X only (got a CPU tensor).')
    lib = _lib.load()
    dev = contours.device
    H, W = int(size[0]), int(size[1])
    con = contours.contiguous().float()
    if sort_by is not None:  # np.argsort, reversed for descending (data/cpn.py:330-334)
        order = torch.as_tensor(np.argsort(torch.as_tensor(sort_by).cpu().numpy()))
        if sort_descending:
            order = order.flip(0)
        con = con[order.to(dev)].contiguous()
    K, S = int(con.shape[0]), int(con.shape[1])
    depth = max(int(initial_depth), 1)
    use_ioa = ioa_thresh is not None

    def result(out, keep, stats):
        res = (out,) + ((keep,) if return_indices else ()) + ((stats,) if return_stats else ())
        return res[0] if len(res) == 1 else res

    if K == 0:
        return result(torch.zeros((H, W, depth), dtype=torch.int32, device=dev), [], dict(rounds=0, channels=depth))
    i32 = dict(dtype=torch.int32, device=dev)
    pts = torch.empty((K, S, 2), **i32)
    boxes = torch.empty((K, 4), **i32)
    check(lib.cpn_labels_prepare(ptr(con), K, S, H, W, int(bool(rounded)), int(bool(clip)), ptr(pts), ptr(boxes),
                                 stream_ptr()), 'labels_prepare')
    extent = int((boxes[:, 2:] - boxes[:, :2]).max().item()) + 1
    cell = max(extent + int(gap) + 1, 8)  # predecessors (boxes within `gap` of each other) sit in adjacent cells
    bmin = boxes[:, :2].min(0).values.cpu().tolist()
    bmax = boxes[:, 2:].max(0).values.cpu().tolist()
    span_w, span_h = max(bmax[0], W - 1) + 1, max(bmax[1], H - 1) + 1
    if min(bmin) < 0:
        raise ValueError('contours2labels: negative coordinates need clip=True')
    gw, gh = -(-span_w // cell), -(-span_h // cell)
    cid, idx = torch.empty(K, **i32), torch.empty(K, **i32)
    check(lib.cpn_labels_bin(ptr(boxes), K, gw, gh, cell, ptr(cid), ptr(idx), stream_ptr()), 'labels_bin')
    scid, perm = torch.sort(cid, stable=True)  # indices ascend within a cell
    sidx = idx[perm].contiguous()
    cbegin, cend = torch.zeros(gw * gh, **i32), torch.zeros(gw * gh, **i32)
    check(lib.cpn_labels_cell_bounds(ptr(scid.contiguous()), K, ptr(cbegin), ptr(cend), stream_ptr()), 'labels_cell_bounds')
    if not clip and (bmax[0] >= W or bmax[1] >= H):
        raise ValueError('contours2labels: contours outside the image need clip=True')
    channels = max(depth, 2)
    canvas = torch.zeros((channels, H, W), **i32)
    state = torch.zeros(K, dtype=torch.uint8, device=dev)
    ready = torch.empty(K, dtype=torch.uint8, device=dev)
    ready_list = torch.empty(K, **i32)
    channel = torch.full((K,), -1, **i32)
    counters = torch.zeros(4, **i32)
    host = (c_int32 * 3)()
    painted, rounds = 0, 0
    while painted < K:
        check(lib.cpn_labels_round(ptr(pts), ptr(boxes), K, S, H, W, int(gap), gw, gh, cell, ptr(sidx), ptr(cbegin),
                                   ptr(cend), ptr(canvas), channels, ptr(state), ptr(ready), ptr(ready_list),
                                   ptr(channel), ptr(counters), host, int(use_ioa), float(ioa_thresh if use_ioa else 0.),
                                   stream_ptr()), 'labels_round')
        rounds += 1
        painted += int(host[0])
        if int(host[1]):  # a contour found every allocated channel occupied: grow the canvas, it retries next round
            canvas = torch.cat((canvas, torch.zeros((channels, H, W), **i32)))
            channels *= 2
        elif int(host[2]) == 0:
            raise RuntimeError('contours2labels: no progress (internal error)')
    used = max(depth, int(channel.max().item()) + 1)
    keep = []
    if use_ioa:
        # painted contours carry the provisional value k + 1: renumber to the reference's running label (lbl only advances
        # for painted contours) with one table look-up over the canvas
        skipped = state == 2
        table = torch.zeros(K + 1, **i32)
        table[1:] = torch.arange(1, K + 1, **i32) - torch.cumsum(skipped.to(torch.int32), 0).to(torch.int32)
        canvas = table[canvas[:used].long()]
        keep = torch.nonzero(~skipped).squeeze(1).cpu().tolist()
    out = canvas[:used].permute(1, 2, 0).contiguous()
    return result(out, keep, dict(rounds=rounds, channels=used))
