"""contours -> label image.  The LOOP of ``contours2labels`` is pinned to the reference itself:
``tests/golden/labels.npz`` = outputs of the imported ``celldetection.data.cpn.contours2labels`` (make_golden.py ``labels``);
the numpy oracle (CPU) and ``labels.hip`` (GPU) are checked against them.  cv2 is absent from the image, so the polygon
FILL rule inside that loop is a restatement of OpenCV's (third party, unpinned -- see oracle/labels_oracle.py)."""
import json
import os

import numpy as np
import pytest
import torch

import labels_oracle as lo


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'labels.npz')


def golden_label_cases():
    g = np.load(GOLDEN)
    for case in json.loads(str(g['cases'])):
        name, kw = case['name'], dict(case['kwargs'])
        if kw.get('sort_by') == 'sort_by':
            kw['sort_by'] = g[f'{name}.sort_by']
        con = g[f'{name}.contours']
        if f'{name}.lengths' in g:
            con = [c[:n].copy() for c, n in zip(con, g[f'{name}.lengths'])]
        yield name, con, tuple(case['size']), kw, g[f'{name}.labels'], (g[f'{name}.keep'] if f'{name}.keep' in g else None)


def test_oracle_matches_reference_loop_goldens():
    """labels_oracle.contours2labels == the imported reference's loop on all 16 fixtures (sort_by / sort_descending,
    ioa_thresh + return_indices, gap, initial_depth, unrounded / unclipped inputs, ragged lists, no contours)."""
    n = 0
    for name, con, size, kw, labels, keep in golden_label_cases():
        arg = [c.copy() for c in con] if isinstance(con, list) else con.copy()
        res = lo.contours2labels(arg, size, **kw)
        got, got_keep = res if kw.get('return_indices') else (res, None)
        assert got.dtype == labels.dtype and got.shape == labels.shape, name
        np.testing.assert_array_equal(got, labels, err_msg=name)
        if keep is not None:
            assert list(got_keep) == list(keep), name
        n += 1
    assert n == 16


@pytest.mark.gpu
def test_contours2labels_matches_reference_loop_goldens():
    """``labels.hip`` through ``cda.contours2labels`` == the imported reference's loop (same 16 fixtures)."""
    import celldetection_amd as cda
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    for name, con, size, kw, labels, keep in golden_label_cases():
        arg = [c.copy() for c in con] if isinstance(con, list) else torch.as_tensor(con).cuda()
        res = cda.contours2labels(arg, size, **kw)
        got, got_keep = res if kw.get('return_indices') else (res, None)
        assert got.dtype == torch.int32 and tuple(got.shape) == labels.shape, (name, tuple(got.shape), labels.shape)
        np.testing.assert_array_equal(got.cpu().numpy(), labels, err_msg=name)
        if keep is not None:
            assert list(got_keep) == list(keep), name


def test_empty_contour_in_a_list_raises_like_the_reference():
    """a zero-length contour must not be dropped silently (labels and returned indices would refer to positions of
    the filtered list); the reference fails on it as well (np.min of an empty array in render_contour, data/cpn.py:248)."""
    import celldetection_amd as cda
    sq = np.array([[1, 1], [5, 1], [5, 5], [1, 5]], np.float32)
    with pytest.raises(ValueError, match='zero-length contour at position 1'):
        cda.contours2labels([sq, np.zeros((0, 2), np.float32), sq + 8], (20, 20))


def random_contours(rng, k, size, s=16, rmin=2., rmax=9., spread=1.):
    H, W = size
    t = np.linspace(0, 2 * np.pi, s, endpoint=False)
    ctr = rng.uniform([-3, -3], [W * spread + 3, H * spread + 3], (k, 1, 2))
    rad = rng.uniform(rmin, rmax, (k, 1, 1)) * rng.uniform(.7, 1.3, (k, s, 1))
    con = ctr + rad * np.stack((np.cos(t), np.sin(t)), -1)[None]
    con[::7] = np.round(con[::7]) + .5  # exact halves: round-half-even
    return con.astype(np.float32)


def test_oracle_fill_rule_and_channels():
    rng = np.random.default_rng(0)
    # convex polygons with integer vertices: the fill contains every lattice point strictly inside and nothing farther
    # than one pixel outside
    for _ in range(20):
        c = np.round(random_contours(rng, 1, (40, 40), s=12, rmin=4, rmax=12)[0] + 20).astype(np.int64)
        m = lo.fill_polygon(c, 0, 0, 64, 64)
        ys, xs = np.mgrid[0:64, 0:64]
        inside = np.ones((64, 64), bool)
        area2 = 0
        for i in range(len(c)):
            a, b = c[i], c[(i + 1) % len(c)]
            cross = (b[0] - a[0]) * (ys - a[1]) - (b[1] - a[1]) * (xs - a[0])
            inside &= cross > 0
            area2 += a[0] * b[1] - b[0] * a[1]
        if area2 < 0:
            continue  # orientation-dependent helper: only check counter-clockwise samples
        assert m[inside].all()
    # channels: overlapping / near contours go to different channels, far ones share channel 0
    t = np.linspace(0, 2 * np.pi, 16, endpoint=False)
    c = np.stack((10 + 6 * np.cos(t), 10 + 6 * np.sin(t)), -1)
    L = lo.contours2labels([c, c + [8, 0], c + [30, 0], c + [15, 0]], (24, 60))
    assert L.shape == (24, 60, 3) and set(np.unique(L)) == {0, 1, 2, 3, 4}
    assert (L[..., 0] == 1).any() and (L[..., 1] == 2).any() and (L[..., 0] == 3).any() and (L[..., 2] == 4).any()
    # gap rule: two boxes 3 px apart still conflict (gap=3), 4 px apart do not
    sq = np.array([[0, 0], [5, 0], [5, 5], [0, 5]], np.float32)
    assert lo.contours2labels([sq, sq + [9, 0]], (10, 30)).shape[2] == 1
    assert lo.contours2labels([sq, sq + [8, 0]], (10, 30)).shape[2] == 2


@pytest.mark.gpu
@pytest.mark.parametrize('k,size,spread,s', [(1, (20, 30), 1., 8), (60, (120, 160), 1., 16), (300, (200, 260), 1., 32),
                                             (150, (64, 64), 1., 12), (40, (50, 70), .3, 16)])
def test_contours2labels_matches_oracle(k, size, spread, s):
    import celldetection_amd as cda
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    rng = np.random.default_rng(k)
    con = random_contours(rng, k, size, s=s, spread=spread)
    if k > 30:
        con[3] = con[3][:1]          # degenerate: a single point
        con[4, :, 1] = con[4, 0, 1]  # degenerate: a horizontal line
    exp = lo.contours2labels(con, size)
    got, st = cda.contours2labels(torch.as_tensor(con).cuda(), size, return_stats=True)
    print(f'contours2labels k={k} size={size}: channels {st["channels"]}, rounds {st["rounds"]}')
    assert tuple(got.shape) == exp.shape, (got.shape, exp.shape)
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
    assert got.dtype == torch.int32
    if k == 60:  # List[Array[num_points, 2]] of different lengths (the reference's second input form)
        ragged = [c[:len(c) - (i % 5)] for i, c in enumerate(con)]
        np.testing.assert_array_equal(cda.contours2labels(ragged, size).cpu().numpy(), lo.contours2labels(ragged, size))
    exp2 = lo.contours2labels(con, size, gap=0, initial_depth=2)
    got2 = cda.contours2labels(torch.as_tensor(con).cuda(), size, gap=0, initial_depth=2)
    np.testing.assert_array_equal(got2.cpu().numpy(), exp2)


def test_oracle_ioa_thresh_and_indices():
    """data/cpn.py:341-357 restated: a contour that lies inside an earlier one is skipped, labels stay consecutive, the index
    list is only filled when ioa_thresh is given."""
    t = np.linspace(0, 2 * np.pi, 16, endpoint=False)
    big = np.stack((20 + 10 * np.cos(t), 20 + 10 * np.sin(t)), -1)
    small = np.stack((20 + 3 * np.cos(t), 20 + 3 * np.sin(t)), -1)
    far = big + [40, 0]
    L, keep = lo.contours2labels([big, small, far], (40, 80), ioa_thresh=.5, return_indices=True)
    assert keep == [0, 2] and set(np.unique(L)) == {0, 1, 2} and (L[20, 60] == 2).any()
    L, keep = lo.contours2labels([small, big, far], (40, 80), ioa_thresh=.5, return_indices=True)
    assert keep == [0, 1, 2] and set(np.unique(L)) == {0, 1, 2, 3}  # the big one is covered to ~10 % only
    L, keep = lo.contours2labels([big, small, far], (40, 80), return_indices=True)
    assert keep == [] and set(np.unique(L)) == {0, 1, 2, 3}
    assert lo.contours2labels([big, big], (40, 80), ioa_thresh=1.).max() == 2  # ioa == 1 is not > 1


@pytest.mark.gpu
@pytest.mark.parametrize('k,size,thr', [(80, (64, 64), .3), (300, (120, 160), 0.), (300, (120, 160), .6), (40, (40, 40), .95)])
def test_contours2labels_ioa_thresh_matches_oracle(k, size, thr):
    import celldetection_amd as cda
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    rng = np.random.default_rng(1000 + k)
    con = random_contours(rng, k, size, s=16)
    exp, keep_exp = lo.contours2labels(con, size, ioa_thresh=thr, return_indices=True)
    got, keep, st = cda.contours2labels(torch.as_tensor(con).cuda(), size, ioa_thresh=thr, return_indices=True, return_stats=True)
    print(f'ioa_thresh={thr} k={k}: kept {len(keep_exp)} of {k}, channels {st["channels"]}, rounds {st["rounds"]}')
    assert 0 < len(keep_exp) < k, 'degenerate case: the threshold skips nothing / everything'
    assert keep == keep_exp
    assert tuple(got.shape) == exp.shape, (got.shape, exp.shape)
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
    # return_indices without a threshold: the reference's list stays empty
    lab, idx = cda.contours2labels(torch.as_tensor(con).cuda(), size, return_indices=True)
    assert idx == [] and torch.equal(lab, cda.contours2labels(torch.as_tensor(con).cuda(), size))
    # sort_by + ioa_thresh (cpn_inference.py passes scores): indices refer to the sorted sequence
    sb = rng.uniform(size=k)
    order = np.argsort(sb)[::-1]
    exp2, keep2 = lo.contours2labels(con[order], size, ioa_thresh=thr, return_indices=True)
    got2, k2 = cda.contours2labels(torch.as_tensor(con).cuda(), size, ioa_thresh=thr, sort_by=sb, return_indices=True)
    assert k2 == keep2
    np.testing.assert_array_equal(got2.cpu().numpy(), exp2)


@pytest.mark.gpu
def test_contours2labels_slide_scale():
    """1e5 contours on an 8192^2 canvas (the post-processing step that follows the slide loop in cpn_inference.py:811):
    every label is present exactly in one channel, channel 0 holds most of them, timing printed."""
    import time
    import celldetection_amd as cda
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    rng = np.random.default_rng(1)
    K, size = 100_000, (8192, 8192)
    con = torch.as_tensor(random_contours(rng, K, size, s=32, rmin=4, rmax=12)).cuda()
    cda.contours2labels(con[:1000], size)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lab, st = cda.contours2labels(con, size, return_stats=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'contours2labels: {K} contours -> {tuple(lab.shape)} in {dt * 1e3:.1f} ms, rounds {st["rounds"]}')
    present = torch.zeros(K + 1, dtype=torch.bool, device='cuda')
    present[lab.reshape(-1).long().unique()] = True
    assert int(present[1:].sum()) >= 0.999 * K  # (contours clipped onto the same border pixels may coincide)
    assert lab.shape[2] >= 2 and int((lab[..., 0] > 0).sum()) > int((lab[..., 1] > 0).sum())


def test_fill_rule_agrees_with_independent_rasterisers_away_from_the_boundary():
    """NOT a pin of cv2's fill rule (cv2 is absent) -- a sanity bound on the restated rule: for random
    simple (star-shaped) and self-intersecting polygons with integer vertices the oracle's filled mask must agree with two
    independent polygon rasterisers -- ``matplotlib.path.Path.contains_points`` at the pixel centres (non-zero winding /
    even-odd differ only for self-intersecting polygons, which are compared with even-odd = cv2's scan-line pairing) and
    ``PIL.ImageDraw.polygon`` -- at every pixel farther than one pixel from the polygon's boundary; and it must contain every
    boundary pixel of the 8-connected outline."""
    mpath = pytest.importorskip('matplotlib.path')
    pil_draw = pytest.importorskip('PIL.ImageDraw')
    from PIL import Image
    import labels_oracle as lo
    rng = np.random.default_rng(0)

    def dist_to_boundary(pts, h, w):
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        d = np.full((h, w), np.inf)
        n = len(pts)
        for i in range(n):
            a, b = pts[i].astype(np.float64), pts[(i + 1) % n].astype(np.float64)
            ab = b - a
            t = np.clip(((xx - a[0]) * ab[0] + (yy - a[1]) * ab[1]) / max(ab @ ab, 1e-12), 0., 1.)
            d = np.minimum(d, np.hypot(xx - (a[0] + t * ab[0]), yy - (a[1] + t * ab[1])))
        return d

    checked = 0
    for case in range(40):
        h, w = int(rng.integers(24, 90)), int(rng.integers(24, 90))
        s = int(rng.integers(3, 24))
        if case % 2 == 0:  # star-shaped around the centre: simple polygon
            ang = np.sort(rng.uniform(0, 2 * np.pi, s))
            rad = rng.uniform(.25, .48, s) * min(h, w)
            pts = np.stack((w / 2 + rad * np.cos(ang), h / 2 + rad * np.sin(ang)), 1)
        else:              # arbitrary vertex order: self-intersections
            pts = np.stack((rng.uniform(1, w - 2, s), rng.uniform(1, h - 2, s)), 1)
        pts = np.rint(pts).astype(np.int32)
        mask = lo.fill_polygon(pts, 0, 0, w, h)
        far = dist_to_boundary(pts, h, w) > 1.0
        yy, xx = np.mgrid[0:h, 0:w]
        if case % 2 == 0:
            inside = mpath.Path(pts.astype(np.float64)).contains_points(np.stack((xx.ravel(), yy.ravel()), 1)).reshape(h, w)
            assert np.array_equal(mask[far], inside[far]), f'case {case}: differs from matplotlib away from the boundary'
        # even-odd reference by ray casting (what scan-line pairing computes), also for self-intersecting polygons
        eo = np.zeros((h, w), bool)
        n = len(pts)
        for i in range(n):
            (ax, ay), (bx, by) = pts[i].astype(np.float64), pts[(i + 1) % n].astype(np.float64)
            if ay == by:
                continue
            cond = ((ay <= yy) & (yy < by)) | ((by <= yy) & (yy < ay))
            xcross = ax + (yy - ay) * (bx - ax) / (by - ay)
            eo ^= cond & (xx < xcross)
        assert np.array_equal(mask[far], eo[far]), f'case {case}: differs from the even-odd rule away from the boundary'
        if case % 2 == 0:
            img = Image.new('1', (w, h), 0)
            pil_draw.Draw(img).polygon([tuple(int(v) for v in p) for p in pts], fill=1, outline=1)
            pil = np.array(img, bool)
            assert np.array_equal(mask[far], pil[far]), f'case {case}: differs from PIL away from the boundary'
        for i in range(n):  # the outline itself is part of the fill (drawContours draws the edges, too)
            for x, y in lo._line_pixels(*pts[i], *pts[(i + 1) % n]):
                assert mask[y, x]
        checked += int(far.sum())
    assert checked > 50000
