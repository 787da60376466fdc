"""contours -> label image: the numpy oracle's invariants on the CPU, the HIP path against the
oracle on the GPU.  cv2 is absent from the image, so the polygon fill rule is a restatement of OpenCV's (parity with
cv2 itself is unpinned -- see oracle/labels_oracle.py); the channel / gap logic follows the reference's Python."""
import numpy as np
import pytest
import torch

import labels_oracle as lo


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
