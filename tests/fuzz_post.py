"""Seeded random inputs through the integer / index / fp32-exact kernels behind the conv graph against the oracle -- bit for bit:
Fourier -> contour decode (any order / sample count), local refinement (out-of-range and x.5 coordinates, buckets), box NMS (ties,
duplicates, degenerate and touching boxes, thresholds incl. 0; the segmented bit-mask kernel and the spatially binned slide-scale
kernel), box voting, border rules, label rasterisation (contours2labels), percentile normalisation.

    python tests/fuzz_post.py [cases] [seed]
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import cpn_oracle as orc  # noqa: E402
from celldetection_amd import ops  # noqa: E402


def boxes_case(rng, g, P):
    """Boxes on a coarse grid (many exact duplicates / touching edges / equal IoUs), a few degenerate ones, scores with ties."""
    q = rng.choice([1., .5, 4.])
    size = rng.choice([32, 64, 200, 1000])
    xy = torch.rand(P, 2, generator=g) * size
    wh = torch.rand(P, 2, generator=g) * rng.choice([8., 30., 120.]) + (0. if rng.random() < .2 else 1.)
    b = torch.cat((xy, xy + wh), 1)
    b = torch.round(b / q) * q
    if P > 4 and rng.random() < .5:  # exact duplicates
        idx = torch.randint(0, P, (P // 4,), generator=g)
        b[torch.randint(0, P, (P // 4,), generator=g)] = b[idx]
    levels = rng.choice([4, 16, 10 ** 6])
    s = torch.round(torch.rand(P, generator=g) * levels) / levels
    return b.float(), s.float()


def run(cases=60, seed=0):
    rng = random.Random(seed)
    dev = torch.device('cuda:0')
    failed = 0

    def report(i, what, msg):
        nonlocal failed
        failed += 1
        print(f'[{i}] {what} FAILED {msg}', flush=True)

    for i in range(cases):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        # ---- decode
        P, order, S = rng.choice([0, 1, 7, 64, 1000]), rng.choice([1, 3, 5, 8, 25]), rng.choice([8, 32, 64, 128])
        four = torch.randn(P, order, 4, generator=g) * rng.choice([.5, 4., 30.])
        loc = torch.rand(P, 2, generator=g) * 300
        got = ops.fouriers2contours(four.to(dev), loc.to(dev), samples=S)[0].cpu().numpy()
        exp = orc.fouriers2contours(four.numpy(), loc.numpy(), S)
        if got.shape != exp.shape or not np.array_equal(got, exp):
            report(i, f'fouriers2contours P={P} order={order} S={S}', f'max abs diff {np.abs(got - exp).max() if got.shape == exp.shape else "shape"}')
        # ---- local refinement
        N, H, W = rng.choice([1, 2, 5]), rng.randrange(8, 90), rng.randrange(8, 120)
        nb = rng.choice([1, 1, 2, 4])
        P = rng.choice([0, 1, 33, 500])
        S = rng.choice([8, 32, 64])
        con = (torch.rand(P, S, 2, generator=g) * torch.tensor([W + 20., H + 20.]) - 10.)
        con = torch.where(torch.rand(P, S, 2, generator=g) < .3, torch.round(con) + .5, con)  # exact x.5: round half to even
        ref = torch.randn(N, 2 * nb, H, W, generator=g) * rng.choice([.3, 3.])
        b = torch.randint(0, N, (P,), generator=g)
        it = rng.choice([0, 1, 4])
        got = ops.local_refinement(con.to(dev), ref.to(dev), it, b.to(dev), num_buckets=nb).cpu().numpy()
        exp = orc.local_refinement(con.numpy(), ref.numpy(), b.numpy(), it, (H, W), num_buckets=nb)[0] if it else con.numpy()
        if not np.array_equal(got, exp):
            report(i, f'local_refinement P={P} S={S} N={N} {H}x{W} buckets={nb} it={it}', f'max abs diff {np.abs(got - exp).max()}')
        # ---- NMS: segmented bit-mask kernel, binned kernel, per-image lists
        P = rng.choice([0, 1, 2, 65, 300, 2500])
        bx, sc = boxes_case(rng, g, P)
        thr = rng.choice([0., .2, .5, .9])
        exp = orc.nms(bx.numpy(), sc.numpy(), thr)
        got = ops.nms(bx.to(dev), sc.to(dev), thr).cpu().numpy()
        if not np.array_equal(got, exp):
            report(i, f'nms P={P} thr={thr}', f'{len(got)} vs {len(exp)} kept')
        if P:
            got = ops.nms_binned(bx.to(dev), sc.to(dev), thr).cpu().numpy()
            if not np.array_equal(got, exp):
                report(i, f'nms_binned P={P} thr={thr}', f'{len(got)} vs {len(exp)} kept')
        parts = sorted(rng.sample(range(P + 1), min(P, rng.choice([0, 1, 3])))) if P else []
        cuts = [0] + parts + [P]
        bl = [bx[a:c] for a, c in zip(cuts, cuts[1:])]
        sl = [sc[a:c] for a, c in zip(cuts, cuts[1:])]
        got = ops.batched_box_nmsi([t.to(dev) for t in bl], [t.to(dev) for t in sl], thr)
        exp = orc.batched_box_nmsi([t.numpy() for t in bl], [t.numpy() for t in sl], thr)
        for j, (a, e) in enumerate(zip(got, exp)):
            if not np.array_equal(a.cpu().numpy(), np.asarray(e)):
                report(i, f'batched_box_nmsi P={P} segment {j} of {len(bl)} thr={thr}', f'{len(a)} vs {len(e)} kept')
        # ---- box voting
        P = rng.choice([1, 40, 600])
        bx, _ = boxes_case(rng, g, P)
        vt, mv = rng.choice([.3, .5]), rng.choice([1., 1.5, 2.2])
        got = np.isin(np.arange(P), ops.filter_by_box_voting(bx.to(dev), vt, mv).cpu().numpy())
        b64 = bx.numpy().astype(np.float64)  # (votes are fp32 sums: boxes whose vote sits on min_vote are left out of the comparison)
        area = (b64[:, 2] - b64[:, 0]) * (b64[:, 3] - b64[:, 1])
        wh = np.clip(np.minimum(b64[:, None, 2:], b64[None, :, 2:]) - np.maximum(b64[:, None, :2], b64[None, :, :2]), 0, None)
        with np.errstate(invalid='ignore', divide='ignore'):
            iou = wh[..., 0] * wh[..., 1] / (area[:, None] + area[None] - wh[..., 0] * wh[..., 1])
        iou32 = iou.astype(np.float32)
        votes = np.where(iou32 > np.float32(vt), iou, 0.).sum(-1)
        clear = (np.abs(votes - mv) > 1e-3) & ~((np.abs(iou - vt) < 1e-6) & (iou > 0)).any(-1)
        exp = np.isin(np.arange(P), orc.filter_by_box_voting(bx.numpy(), vt, mv)[0])
        if not np.array_equal(got[clear], exp[clear]) or not np.array_equal(exp[clear], (votes >= mv)[clear]):
            report(i, f'box voting P={P} thresh={vt} min_vote={mv}', f'{got.sum()} vs {exp.sum()} kept')
        # ---- border rule
        P, S = rng.choice([0, 5, 400]), rng.choice([8, 32])
        size = (rng.randrange(32, 200), rng.randrange(32, 200))
        con = torch.round((torch.rand(P, S, 2, generator=g) * torch.tensor([size[1] + 8., size[0] + 8.]) - 4.) * 2) / 2
        sides = [rng.random() < .5 for _ in range(4)]
        pad = rng.choice([0, 1, 4, 6])
        off = np.array([rng.randrange(-50, 50), rng.randrange(-50, 50)], np.float32) if rng.random() < .5 else None
        got = ops.remove_border_contours(con.to(dev), size, pad, top=sides[0], right=sides[1], bottom=sides[2], left=sides[3],
                                         offsets=None if off is None else torch.as_tensor(off).to(dev))
        exp = orc.remove_border_contours(con.numpy(), size, pad, top=sides[0], right=sides[1], bottom=sides[2], left=sides[3], offsets=off)
        got = got.cpu().numpy()
        got = got if got.dtype == bool else np.isin(np.arange(P), got)
        exp = np.asarray(exp)
        exp = exp if exp.dtype == bool else np.isin(np.arange(P), exp)
        if not np.array_equal(got, exp):
            report(i, f'border rule P={P} size={size} pad={pad} sides={sides} offsets={off}', f'{got.sum()} vs {exp.sum()} kept')
        # ---- label rasterisation (contours2labels, data/cpn.py:292-358) and percentile normalisation (data/misc.py:156-161)
        if i % 3 == 0:
            import labels_oracle as lo
            import preprocess_oracle as po
            import celldetection_amd as cda
            from celldetection_amd.preprocess import normalize_percentile
            nrng = np.random.default_rng(seed * 7919 + i)
            k, S = rng.choice([1, 12, 90, 250]), rng.choice([6, 16, 33])
            size = (rng.randrange(16, 150), rng.randrange(16, 190))
            t = np.linspace(0, 2 * np.pi, S, endpoint=False)
            ctr = nrng.uniform([-3, -3], [size[1] + 3, size[0] + 3], (k, 1, 2))
            rad = nrng.uniform(1.5, rng.choice([5., 14.]), (k, 1, 1)) * nrng.uniform(.6, 1.4, (k, S, 1))
            con = (ctr + rad * np.stack((np.cos(t), np.sin(t)), -1)[None]).astype(np.float32)
            con[::5] = np.round(con[::5]) + .5
            kw = rng.choice([{}, dict(gap=0), dict(gap=2, initial_depth=2)])
            exp = lo.contours2labels(con, size, **kw)
            got = cda.contours2labels(torch.as_tensor(con).to(dev), size, **kw).cpu().numpy()
            if got.shape != exp.shape or not np.array_equal(got, exp):
                report(i, f'contours2labels k={k} S={S} size={size} {kw}', f'shape {got.shape} vs {exp.shape}' if got.shape != exp.shape
                       else f'{(got != exp).sum()} pixels differ')
            dt = rng.choice(['uint8', 'uint16', 'float32'])
            shp = (rng.choice([1, 3]), rng.randrange(20, 200), rng.randrange(20, 230))
            if dt == 'uint8':
                x = nrng.integers(0, rng.choice([40, 256]), shp).astype(np.uint8)
            elif dt == 'uint16':
                x = nrng.gamma(2., rng.choice([60., 900.]), shp).clip(0, 65535).astype(np.uint16)
            else:
                x = (nrng.standard_normal(shp) * rng.choice([.1, 50.])).astype(np.float32)
            pct = rng.choice([99.9, 99., (1., 97.5), (0.5, 99.5)])
            tx = torch.as_tensor(x.astype(np.int32)).to(torch.uint16) if dt == 'uint16' else torch.as_tensor(x)
            got = normalize_percentile(tx.to(dev), pct).cpu().numpy()
            exp = po.normalize_percentile(x, pct)
            if got.shape != exp.shape or not np.array_equal(got, exp):
                report(i, f'normalize_percentile {dt} {shp} pct={pct}', f'{(got != exp).sum()} values differ' if got.shape == exp.shape else 'shape')
    print('fuzz_post:', cases, 'cases,', failed, 'failed')
    return failed


if __name__ == '__main__':
    sys.exit(1 if run(*(int(a) for a in sys.argv[1:3])) else 0)
