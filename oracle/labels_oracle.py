"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY: contours -> label image.

Restates ``celldetection.data.contours2labels`` (celldetection/data/cpn.py:292-358) and ``render_contour``
(:245-255) for the default arguments used by celldetection_scripts/cpn_inference.py:811 (rounded, clip, gap=3,
initial_depth=1, int32) and the ``ioa_thresh`` / ``return_indices`` options.

**Pinning (round 5).**  The LOOP -- ordering (``sort_by`` / ``sort_descending``), rounding, clipping, the bounding box,
``ioa_thresh`` / ``return_indices``, the gap rule, the channel search, label numbering -- is pinned to the reference
itself: ``tests/golden/labels.npz`` holds outputs of the IMPORTED ``celldetection.data.cpn.contours2labels`` (with
``cv2.drawContours`` replaced by ``ref_shim.cv2_drawContours``, which calls ``fill_polygon`` below), and
``tests/test_labels.py`` checks this restatement against them.  **The fill rule itself stays unpinned (third party):**
``render_contour`` calls ``cv2.drawContours(thickness=-1)``; OpenCV is not installed in the build
image and is not part of /root/reference, so the polygon fill rule below is a restatement of OpenCV's published
algorithm (modules/imgproc/src/drawing.cpp, 4.x: ``CollectPolyEdges`` draws every edge with the 8-connected
``LineIterator`` (left-to-right), ``FillEdgeCollection`` fills the scanlines between paired, rounded 16.16 fixed-point
edge crossings, top-inclusive / bottom-exclusive) that could not be checked against cv2 here.  The channel/gap logic is
the reference's own Python and is restated line by line.
"""
import numpy as np


def _line_pixels(ax, ay, bx, by):
    """8-connected LineIterator with left_to_right=True: err0 = dmaj - 2*dmin, diagonal step iff err < 0."""
    if bx < ax:
        ax, ay, bx, by = bx, by, ax, ay
    dx, dy = bx - ax, abs(by - ay)
    sy = -1 if by < ay else 1
    out = []
    if dy <= dx:
        err, x, y = dx - 2 * dy, ax, ay
        for _ in range(dx + 1):
            out.append((x, y))
            if err < 0:
                err += 2 * dx - 2 * dy
                y += sy
            else:
                err -= 2 * dy
            x += 1
    else:
        err, x, y = dy - 2 * dx, ax, ay
        for _ in range(dy + 1):
            out.append((x, y))
            if err < 0:
                err += 2 * dy - 2 * dx
                x += 1
            else:
                err -= 2 * dx
            y += sy
    return out


def _cdiv(a, b):
    """C integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def fill_polygon(points, x0, y0, w, h):
    """Boolean mask [h, w] (origin x0, y0) of the filled polygon with integer vertices ``points`` [S, 2] (xy)."""
    m = np.zeros((h, w), bool)
    pts = [(int(p[0]), int(p[1])) for p in points]
    n = len(pts)
    for i in range(n):
        (ax, ay), (bx, by) = pts[i], pts[(i + 1) % n]
        for x, y in _line_pixels(ax, ay, bx, by):
            if 0 <= x - x0 < w and 0 <= y - y0 < h:
                m[y - y0, x - x0] = True
    for y in range(y0, y0 + h):
        xs = []
        for i in range(n):
            (ax, ay), (bx, by) = pts[i], pts[(i + 1) % n]
            if ay == by:
                continue
            (tx, ty), yb = ((ax, ay), by) if ay < by else ((bx, by), ay)
            if not (ty <= y < yb):
                continue
            ddx = _cdiv((bx - ax) * 65536, by - ay)
            xs.append((tx * 65536 + (y - ty) * ddx + 32768) >> 16)
        xs.sort()
        for a, b in zip(xs[0::2], xs[1::2]):
            lo, hi = max(a - x0, 0), min(b - x0, w - 1)
            if hi >= lo:
                m[y - y0, lo:hi + 1] = True
    return m


def contours2labels(contours, size, rounded=True, clip=True, initial_depth=1, gap=3, dtype='int32', ioa_thresh=None,
                    sort_by=None, sort_descending=True, return_indices=False):
    """data/cpn.py:329-358 (sort_by: :330-334 -- ``np.argsort``, reversed for descending; the returned indices are then
    positions in the SORTED sequence, as there; ioa_thresh: :341-350; return_indices: :356-357 -- the index list is only filled
    when ``ioa_thresh`` is given, like in the reference)."""
    H, W = size
    if sort_by is not None:
        indices = np.argsort(sort_by)
        if sort_descending:
            indices = indices[::-1]
        contours = [contours[i] for i in indices]
    labels = np.zeros((H, W, initial_depth), dtype=dtype)
    lbl = 1
    keep = []
    for idx, contour in enumerate(contours):
        contour = np.array(contour, np.float32)
        if rounded:
            contour = np.round(contour)
        if clip:
            contour[..., 0] = np.clip(contour[..., 0], 0, W - 1)
            contour[..., 1] = np.clip(contour[..., 1], 0, H - 1)
        xmin, ymin = np.floor(contour.min(0)).astype(int)
        xmax, ymax = np.ceil(contour.max(0)).astype(int)
        a = fill_polygon(contour.astype(np.int32), xmin, ymin, xmax - xmin + 1, ymax - ymin + 1).astype(dtype) * lbl
        if ioa_thresh is not None:
            m = a > 0
            crp = (labels[ymin:ymin + a.shape[0], xmin:xmin + a.shape[1]] > 0).any(-1)
            ioa = crp[m].sum() / m.sum()
            if ioa > ioa_thresh:
                continue
            keep.append(idx)
        lbl += 1
        s = (labels[max(0, ymin - gap): gap + ymin + a.shape[0], max(0, xmin - gap): gap + xmin + a.shape[1]] > 0).sum((0, 1))
        i = next(i for i in range(labels.shape[2] + 1) if not (i < labels.shape[2] and np.any(s[i])))
        if i >= labels.shape[2]:
            labels = np.concatenate((labels, np.zeros((H, W, 1), dtype=dtype)), axis=-1)
        labels[ymin:ymin + a.shape[0], xmin:xmin + a.shape[1], i] += a
    if return_indices:
        return labels, keep
    return labels
