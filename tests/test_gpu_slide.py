"""BASELINE.json configs[3] at its own geometry on one GPU: CpnResNeXt101UNet (full width, the bench's
synthetic ginoro-shaped weights), a synthetic uint8 slide, tiles 512 / stride 384, batch 16 -- the slide loop of
celldetection_scripts/cpn_inference.py:311-429 (tiling -> forward(offsets) -> border rule -> gather -> global NMS).

* the product default (``sparse_heads = 'auto'``: score-gated location / Fourier heads) == the dense reference graph, bit for bit;
* the loop == per-tile ``model(tile, offsets=...)`` + the ORACLE's border rule + the ORACLE's NMS applied to the same per-tile GPU
  detections, exactly (ragged last batch: 25 tiles = 16 + 9);
* one tile of the slide through the fp32 verification path against the fp32 CPU oracle: the north-star statement (identical
  index sets, contours within 1e-4)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def flagship(dev):
    sys.path.insert(0, ROOT)
    from bench import build_model
    model, sd = build_model('CpnResNeXt101UNet', dev)
    return model, sd


def test_slide_loop_at_configs3_geometry(dev, flagship):
    import cpn_oracle as orc
    from celldetection_amd import inference
    from test_gpu_model import north_star_check
    model, sd = flagship
    S, crop, stride, border = 2048, (512, 512), (384, 384), 4
    slide = torch.randint(0, 256, (3, S, S), dtype=torch.uint8, device=dev, generator=torch.Generator(dev).manual_seed(3))
    kw = dict(crop_size=crop, strides=stride, batch_size=16, border_removal=border)
    model.sparse_heads = False
    t = {}
    dense = inference.tiled_inference(model, slide, timings=t, **kw)
    assert t['tiles_local'] == 25 and t['detections_final'] > 2000, t
    assert t['detections_gathered'] > t['detections_final']  # overlapping tiles: the global NMS has work
    model.sparse_heads = 'auto'
    gated = inference.tiled_inference(model, slide, **kw)
    for k in inference.KEYS:
        assert torch.equal(gated[k], dense[k]), f'score-gated heads changed {k}'
    # per-tile forward + oracle stitching of the same per-tile detections
    model.sparse_heads = False
    slices, overlaps, shape = orc.get_tiling_slices((S, S), crop, stride)
    assert len(slices) == 25
    coll = {}
    for idx, ((h0, h1), (w0, w1)) in enumerate(slices):
        offs = torch.tensor([[w0, h0]])
        y = model(slide[None, :, h0:h1, w0:w1], offsets=offs)
        h_i, w_i = np.unravel_index(idx, shape)
        con = y['contours'][0].cpu().numpy()
        keep = orc.remove_border_contours(con, crop, border, top=h_i > 0, right=w_i < shape[1] - 1, bottom=h_i < shape[0] - 1,
                                          left=w_i > 0, offsets=-offs[0].numpy().astype(np.float32))
        for k in inference.KEYS:
            v = y[k][0].cpu().numpy()[keep]
            coll[k] = np.concatenate((coll[k], v)) if k in coll else v
    assert len(coll['scores']) == t['detections_gathered']
    keep = orc.nms(coll['boxes'], coll['scores'], model.nms_thresh)
    for k in inference.KEYS:
        np.testing.assert_array_equal(dense[k].cpu().numpy(), coll[k][keep], err_msg=k)
    # one tile (second row, second column: neighbours on all sides) on the fp32 path vs the fp32 CPU oracle
    (h0, h1), (w0, w1) = slices[6]
    tile = slide[None, :, h0:h1, w0:w1]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    offs = torch.tensor([[w0, h0]])
    ref = orc.cpn_forward({k: v.cpu() for k, v in sd.items()}, tile.cpu().float() / 255, nms=True, offsets=offs)
    assert len(ref['scores'][0]) > 100
    model.precision = 'fp32'
    try:
        north_star_check('configs[3] tile 6 (fp32 path, uint8 crop, offsets)', model(tile, offsets=offs), ref)
    finally:
        model.precision = 'bf16'
