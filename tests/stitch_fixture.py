"""Helpers around tests/golden/stitch_dups.npz (per-tile detection lists with cross-tile duplicates + what the
reference's own border rule / torchvision nms make of them)."""
import os
from collections import OrderedDict

import numpy as np
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
KEYS = ('contours', 'boxes', 'scores', 'classes', 'locations', 'fourier', 'contour_proposals')


def load():
    return np.load(os.path.join(G, 'stitch_dups.npz'))


def tile_table(g):
    """(w0, h0) -> tile index, from the reference tiling table of the fixture's geometry."""
    import cpn_oracle as orc
    slices, overlaps, shape = orc.get_tiling_slices(tuple(int(i) for i in g['size']), tuple(int(i) for i in g['crop']),
                                                    tuple(int(i) for i in g['stride']))
    return {(w0, h0): i for i, ((h0, h1), (w0, w1)) in enumerate(slices)}, slices, overlaps, shape


def forward_fn(g, device):
    """``forward_fn`` for ``inference.tiled_inference``: returns the fixture's detections of the requested tiles."""
    table = tile_table(g)[0]

    def fn(tiles, offsets, **kw):
        out = OrderedDict((k, []) for k in KEYS)
        for n in range(tiles.shape[0]):
            i = table[(int(offsets[n, 0]), int(offsets[n, 1]))]
            for k in KEYS:
                out[k].append(torch.as_tensor(g[f'tile{i}.{k}']).to(device))
        return out

    return fn


class StubModel:
    nms_thresh, samples, order = .2, 16, 3

    class core:
        order = 3
