"""GPU parity tests of the whole path (through libcpn_hip.so) against golden vectors from the reference.

Stage-wise parity (SURVEY section 7 "hard parts"):
  (a) conv stack (bf16 MFMA) vs the reference's fp32 head maps: tolerance (bf16 activations/weights);
  (b) post-processing on the REFERENCE's head maps: index sets bit-exact, coordinates within 1e-4;
  (c) end-to-end: detections matched by IoU against the reference's (match rate).
"""
import os

import numpy as np
import pytest
import torch

from model_specs import ALL_SPECS, HEAD_SPECS, MODEL_SPECS, SIZE_SPECS, VARIANT_SPECS, bf16_bounds

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
KEYS = ('contours', 'boxes', 'scores', 'classes', 'locations', 'fourier', 'contour_proposals')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def build(name, dev, fixture=None):
    import celldetection_amd as cda
    from celldetection_amd.synth import synth_state_dict
    spec = ALL_SPECS[name]
    g = np.load(os.path.join(G, fixture or f'model_{name}.npz'))
    model = getattr(cda.models, spec['cls'])(**spec['kwargs'])
    assert list(model.state_dict().keys()) == [str(k) for k in g['sd_keys']]
    overrides = {k[len('override.'):]: torch.as_tensor(g[k]) for k in g.files if k.startswith('override.')}
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=int(g['seed']) if 'seed' in g.files else 0,
                                           overrides=overrides))
    return model.to(dev), g


def check_exact(prefix, y, g, n, raw_atol=1e-4, flip_frac=0.):
    """Index sets equal (shapes/classes), final contours / boxes / scores within 1e-4 (north-star tolerance);
    ``raw_atol`` applies to the un-snapped regression outputs (locations, fourier, contour_proposals).
    ``flip_frac``: fraction of contour / box coordinates that may differ by more (end-to-end runs only: a proposal
    coordinate within 1e-6 of x.5 rounds to the other pixel in ``local_refinement`` -- a discontinuity of the
    algorithm, not of the arithmetic)."""
    keys = KEYS + (('box_uncertainties',) if f'{prefix}.box_uncertainties.0' in g.files else ())
    if prefix == 'nms':  # a fixture that keeps a handful of detections pins next to nothing of the NMS keep set
        assert all(len(g[f'nms.scores.{i}']) >= 20 for i in range(n)), 'thin golden fixture: regenerate with >= 30 kept detections'
    if len(keys) == len(KEYS):
        assert y['box_uncertainties'] is None
    for k in keys:
        for i in range(n):
            exp, got = g[f'{prefix}.{k}.{i}'], y[k][i].cpu().numpy()
            assert got.shape == exp.shape, (prefix, k, i, got.shape, exp.shape)
            if k == 'classes':
                np.testing.assert_array_equal(got, exp)
            else:
                atol = 1e-4 if k in ('contours', 'boxes', 'scores', 'box_uncertainties') else raw_atol
                if flip_frac and k in ('contours', 'boxes') and exp.size:
                    bad = float((np.abs(got - exp) > atol).mean())
                    assert bad <= flip_frac, f'{prefix}.{k}.{i}: {bad:.2e} of the coordinates off by > {atol}'
                    continue
                np.testing.assert_allclose(got, exp, rtol=0, atol=atol, err_msg=f'{prefix}.{k}.{i}')


def north_star_check(label, got, ref, flip_frac=1e-3, raw_atol=1e-3):
    """BASELINE.json north_star, literally: ``outputs match the reference CPU PyTorch forward on identical inputs/weights
    (contour coords within 1e-4 fp32, score-threshold/NMS index sets bit-exact)``.  ``got`` = the HIP path with
    ``precision='fp32'``, ``ref`` = ``cpn_oracle.cpn_forward`` (pinned to the imported reference by the goldens).
    Index sets: same number of detections per image in the same order -- proposals come in ``torch.where`` order and NMS
    survivors in descending-score order, so equal counts + equal classes + positions that agree to a fraction of a pixel
    (``locations``: the proposal's own pixel + a sub-pixel offset, scaled) ARE equal index sets.  Coordinates: contours /
    boxes / scores within 1e-4, except the documented pixel-snap flips (``local_refinement`` rounds half-to-even; a
    coordinate within summation-order noise of x.5 lands on the other pixel and reads another refinement vector): their
    fraction is printed and bounded by ``flip_frac``.  The raw, un-snapped regression outputs (locations, fourier,
    contour_proposals: O(100) px x 1e-6 relative) get ``raw_atol``.  -> dict of the measured figures."""
    rep = {}
    n = len(ref['scores'])
    assert len(got['scores']) == n
    for i in range(n):
        # NMS survivors come in descending-score order: two scores within the conv stack's fp32 summation-order noise (~1e-6) may
        # swap places without any index SET differing.  Such swaps are undone here (rows matched by their proposal pixel through
        # `locations`), counted and reported; a displaced row whose score is NOT within 5e-6 of the row it swapped with fails.
        perm = None
        gl, el = got['locations'][i].cpu().numpy(), np.asarray(ref['locations'][i])
        if gl.shape == el.shape and len(el) and np.abs(gl - el).max() >= .25:
            d2 = ((el[:, None, :].astype(np.float64) - gl[None, :, :]) ** 2).sum(-1)
            perm = d2.argmin(1)
            es = np.asarray(ref['scores'][i], np.float64)
            moved = np.nonzero(perm != np.arange(len(perm)))[0]
            assert len(set(perm.tolist())) == len(perm) and d2[np.arange(len(perm)), perm].max() < .25 ** 2, \
                f'{label}: the detections of image {i} are not a re-ordering of the reference\'s'
            assert np.abs(es[moved] - es[perm[moved]]).max() < 5e-6, f'{label}: image {i}: order differs beyond score ties'
            rep['score_tie_swaps'] = rep.get('score_tie_swaps', 0) + len(moved)
        for k in ('scores', 'classes', 'locations', 'contours', 'boxes', 'fourier', 'contour_proposals'):
            g, e = got[k][i].cpu().numpy(), np.asarray(ref[k][i])
            if perm is not None and g.shape == e.shape:
                g = g[perm]
            assert g.shape == e.shape, f'{label}: {k}[{i}] index sets differ: {g.shape} vs {e.shape}'
            if k == 'classes':
                np.testing.assert_array_equal(g, e, err_msg=f'{label}.{k}.{i}')
                continue
            if e.size == 0:
                continue
            d = np.abs(g.astype(np.float64) - e.astype(np.float64))
            rep[f'{k}.max'] = max(rep.get(f'{k}.max', 0.), float(d.max()))
            if k == 'locations':
                assert d.max() < .25, f'{label}: proposal {int(d.max(1).argmax())} of image {i} sits on another pixel'
            if k in ('contours', 'boxes'):
                bad = float((d > 1e-4).mean())
                rep[f'{k}.frac_off'] = max(rep.get(f'{k}.frac_off', 0.), bad)
                rep[f'{k}.max_noflip'] = max(rep.get(f'{k}.max_noflip', 0.), float(d[d < .1].max()) if (d < .1).any() else 0.)
                assert bad <= flip_frac, f'{label}.{k}.{i}: {bad:.2e} of the coordinates off by > 1e-4 (allowed {flip_frac})'
            elif k == 'scores':
                np.testing.assert_allclose(g, e, rtol=0, atol=1e-4, err_msg=f'{label}.{k}.{i}')
            else:
                np.testing.assert_allclose(g, e, rtol=0, atol=raw_atol, err_msg=f'{label}.{k}.{i}')
    print(f'north-star check {label}: detections {[len(t) for t in ref["scores"]]}; ' +
          ', '.join(f'{k} {v:.2e}' for k, v in sorted(rep.items())))
    return rep


@pytest.mark.parametrize('name', list(MODEL_SPECS))
def test_postprocess_on_reference_head_maps(dev, name):
    """(b): decode -> refinement -> boxes -> NMS on the reference's fp32 head maps is exact."""
    model, g = build(name, dev)
    x = g['x']
    n, size = x.shape[0], tuple(x.shape[-2:])
    maps = [torch.sigmoid(torch.as_tensor(g['core.scores'])).to(dev), torch.as_tensor(g['core.locations']).to(dev),
            torch.as_tensor(g['core.refinement']).to(dev), torch.as_tensor(g['core.fourier']).to(dev)]
    check_exact('nms', model.postprocess(*maps, size), g, n)
    check_exact('nonms', model.postprocess(*maps, size, nms=False), g, n)
    check_exact('offs', model.postprocess(*maps, size, offsets=torch.as_tensor(g['offsets'])), g, n)
    # score bounds: the masks are resized by the library's own fp32 bilinear kernel (torch CPU's arithmetic), so the
    # bounded scores / index sets are compared like every other output
    check_exact('bounds', model.postprocess(*maps, size, scores_upper_bound=torch.as_tensor(g['scores_upper_bound']).to(dev),
                                            scores_lower_bound=torch.as_tensor(g['scores_lower_bound']).to(dev)), g, n)
    if name == 'CpnU22':
        model.refinement_iterations = 0  # contours IS contour_proposals in the reference: offsets land on it twice
        check_exact('noref_offs', model.postprocess(*maps, size, offsets=torch.as_tensor(g['offsets'])), g, n)
        model.refinement_iterations = 4
    if name == 'CpnU22':
        model.samples, model.refinement_iterations, model.score_thresh, model.nms_thresh = 17, 2, .7, .5
        check_exact('attr', model.postprocess(*maps, size), g, n)
        model.order = 3
        check_exact('attr_order3', model.postprocess(*maps, size), g, n)


@pytest.mark.parametrize('name', list(MODEL_SPECS))
def test_conv_stack_vs_reference_maps(dev, name):
    """(a): bf16 MFMA conv stack vs the reference's fp32 maps (and implicitly vs the oracle, pinned to them)."""
    model, g = build(name, dev)
    x = torch.as_tensor(g['x']).to(dev)
    s, l, r, f = [t.cpu() for t in model.core_forward(x)]
    torch.cuda.synchronize()
    exp = dict(scores=torch.sigmoid(torch.as_tensor(g['core.scores'])), locations=torch.as_tensor(g['core.locations']),
               refinement=torch.as_tensor(g['core.refinement']), fourier=torch.as_tensor(g['core.fourier']))
    report = {}
    for key, got in (('scores', s), ('locations', l), ('refinement', r), ('fourier', f)):
        e = exp[key]
        assert got.shape == e.shape
        assert torch.isfinite(got).all(), key
        err = (got - e).abs()
        # relative L2 error of the map; bf16 (8 mantissa bits) through O(10..100) layers
        rel = (err.norm() / (e.norm() + 1e-12)).item()
        report[key] = (rel, err.max().item(), e.abs().max().item())
    print(name, {k: f'relL2 {v[0]:.3e} max {v[1]:.3e} (ref max {v[2]:.2e})' for k, v in report.items()})
    bounds, _ = bf16_bounds(name)  # 2 x the error measured on the MI355X (tests/golden/bf16_measured.json)
    for key, (rel, mx, ref_mx) in report.items():
        assert rel < bounds[key], f'{name} {key}: relative L2 error {rel:.3e} (bound {bounds[key]:.3e})'


@pytest.mark.parametrize('name', ['CpnU22', 'CpnResNeXt101UNet', 'CpnResNet18FPN', 'CpnResNet50FPN', 'CpnResNet18UNet'])
def test_bf16_gates_catch_a_one_ulp_weight_perturbation(dev, name):
    """the bf16 gates (2 x the measured error, tests/golden/bf16_measured.json) are the only model-level guard on
    the product kernels, so they must notice a kernel-sized regression.  Here every packed bf16 weight of the engine is moved by
    ONE ulp away from zero (bit pattern + 1: twice the rounding error the weights already carry, in one direction) -- at least one
    head map must leave its bound."""
    model, g = build(name, dev)
    x = torch.as_tensor(g['x']).to(dev)
    exp = dict(scores=torch.sigmoid(torch.as_tensor(g['core.scores'])), locations=torch.as_tensor(g['core.locations']),
               refinement=torch.as_tensor(g['core.refinement']), fourier=torch.as_tensor(g['core.fourier']))
    bounds, _ = bf16_bounds(name)
    rel = lambda maps: {k: ((m.cpu() - exp[k]).norm() / (exp[k].norm() + 1e-12)).item()
                        for k, m in zip(('scores', 'locations', 'refinement', 'fourier'), maps)}
    clean = rel(model.core_forward(x))
    assert all(clean[k] < bounds[k] for k in clean), (clean, bounds)
    eng = model.engine(dev)
    assert eng.wblob.dtype == torch.bfloat16
    eng.wblob.view(torch.int16).add_(1)  # in place: the native plan reads this very buffer
    torch.cuda.synchronize()
    bad = rel(model.core_forward(x))
    print(name, 'clean', {k: f'{v:.3e}' for k, v in clean.items()}, 'perturbed', {k: f'{v:.3e}' for k, v in bad.items()},
          'bounds', {k: f'{v:.3e}' for k, v in bounds.items()})
    assert any(bad[k] >= bounds[k] for k in bad), 'a 1-ulp perturbation of every weight passes the bf16 gates'


def _iou_match_rate(boxes_a, boxes_b, thr=.5):
    if len(boxes_a) == 0 or len(boxes_b) == 0:
        return 1. if len(boxes_a) == len(boxes_b) else 0.
    a, b = torch.as_tensor(boxes_a), torch.as_tensor(boxes_b)
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    iou = inter / (area(a)[:, None] + area(b)[None] - inter + 1e-9)
    return float(((iou.max(1).values > thr).float().mean() + (iou.max(0).values > thr).float().mean()) / 2)


@pytest.mark.parametrize('name', list(MODEL_SPECS))
def test_end_to_end_match_rate(dev, name):
    """(c): full HIP forward vs the reference's detections (bf16 => IoU-matched, not bit-exact)."""
    model, g = build(name, dev)
    x = torch.as_tensor(g['x']).to(dev)
    y = model(x, nms=False)
    rates = []
    for i in range(x.shape[0]):
        rates.append(_iou_match_rate(y['boxes'][i].cpu().numpy(), g[f'nonms.boxes.{i}']))
        n_ref, n_got = len(g[f'nonms.scores.{i}']), len(y['scores'][i])
        # measured (round 2, six model families): match rates 0.956 .. 1.0, counts within a few per cent
        assert abs(n_ref - n_got) <= max(3, 0.1 * n_ref), f'{name}[{i}]: proposals {n_got} vs reference {n_ref}'
    print(name, 'proposal IoU>0.5 match rates', rates)
    assert min(rates) > bf16_bounds(name)[1], (rates, bf16_bounds(name)[1])  # measured - 0.03
    y = model(x)  # with NMS: output contract
    assert list(y.keys()) == ['contours', 'boxes', 'scores', 'classes', 'locations', 'fourier', 'contour_proposals',
                              'box_uncertainties']
    assert y['box_uncertainties'] is None and len(y['contours']) == x.shape[0]
    s = MODEL_SPECS[name]['cpn_kwargs']['samples']
    assert y['contours'][0].shape[1:] == (s, 2) and y['classes'][0].dtype == torch.int64


def test_input_range_assert_and_uint8(dev):
    model, g = build('CpnU22', dev)
    x = torch.as_tensor(g['x']).to(dev)
    bad = x.clone()
    bad[0, 0, 0, 0] = 1.5
    with pytest.raises(AssertionError, match='Inputs should be in interval'):
        model(bad)
    with pytest.raises(RuntimeError):
        model(x.cpu())
    u8 = (x * 255).to(torch.uint8)
    a = model.core_forward(u8)
    b = model.core_forward(u8.float() / 255)
    for p, q in zip(a, b):
        assert torch.equal(p, q)


def test_fetchable_model_roundtrip(dev, tmp_path):
    import celldetection_amd as cda
    model, g = build('CpnU22', dev)
    f = cda.save_fetchable_model(model, str(tmp_path / 'tiny_CpnU22'))
    m2 = cda.load_model(f, map_location=dev)
    assert type(m2).__name__ == 'CpnU22' and m2.hparams['backbone_kwargs'] == model.hparams['backbone_kwargs']
    x = torch.as_tensor(g['x']).to(dev)
    for p, q in zip(model.core_forward(x), m2.core_forward(x)):
        assert torch.equal(p, q)


def test_checkpoint_written_by_the_reference_loads_and_reproduces_its_outputs(dev):
    """tests/golden/ref_checkpoint_CpnU22.pt was written by the REFERENCE's own ``save_fetchable_model``
    (util/util.py:545-560; generator: make_golden.py gen_checkpoint) with two attributes changed after construction
    (``updated_kwargs``): ``cda.load_model`` must build the same model, and the fp32 path must reproduce the reference's
    detections for the recorded input (index sets exact, coordinates within 1e-4)."""
    import celldetection_amd as cda
    from model_specs import G
    model = cda.load_model(os.path.join(G, 'ref_checkpoint_CpnU22.pt'), map_location=dev)
    assert type(model).__name__ == 'CpnU22' and model.score_thresh == .85 and model.samples == 24
    g = np.load(os.path.join(G, 'ref_checkpoint_CpnU22_outputs.npz'))
    x = torch.as_tensor(g['x']).to(dev)
    model.precision = 'fp32'
    assert len(g['nms.scores.0']) >= 20
    check_exact('nms', model(x), g, 1, raw_atol=5e-4)
    model.precision = 'bf16'
    y = model(x)
    rate = _iou_match_rate(y['boxes'][0].cpu().numpy(), g['nms.boxes.0'])
    assert rate > .75, rate  # bf16: matched, not identical (23 small detections at score_thresh .85: measured 0.84)


def test_tiled_inference_stitching(dev):
    """Slide-level loop (tiling -> forward(offsets) -> border removal -> global NMS) on the GPU:
    (1) exact agreement with the oracle's stitching applied to the SAME per-tile GPU detections,
    (2) IoU match against the reference's final detections of the golden stitch fixture."""
    import cpn_oracle as orc
    from celldetection_amd import inference
    model, g = build('CpnU22', dev, fixture='stitch.npz')
    img = torch.as_tensor(g['img']).to(dev)
    crop, stride, border = tuple(int(i) for i in g['crop']), tuple(int(i) for i in g['stride']), int(g['border'])
    res = inference.tiled_inference(model, img, crop, stride, batch_size=4, border_removal=border)
    # (1) oracle stitching of the per-tile GPU outputs
    slices, overlaps, shape = orc.get_tiling_slices(tuple(img.shape[-2:]), crop, stride)
    coll = {}
    for idx, ((h0, h1), (w0, w1)) in enumerate(slices):
        offs = torch.tensor([[w0, h0]])
        y = model(img[..., h0:h1, w0:w1], offsets=offs)
        h_i, w_i = np.unravel_index(idx, shape)
        con = y['contours'][0].cpu().numpy()
        keep = orc.remove_border_contours(con, crop, border, top=h_i > 0, right=w_i < shape[1] - 1,
                                          bottom=h_i < shape[0] - 1, left=w_i > 0, offsets=-offs[0].numpy().astype(np.float32))
        for k in inference.KEYS:
            v = y[k][0].cpu().numpy()[keep]
            coll[k] = np.concatenate((coll[k], v)) if k in coll else v
    keep = orc.nms(coll['boxes'], coll['scores'], model.nms_thresh)
    for k in inference.KEYS:
        np.testing.assert_array_equal(res[k].cpu().numpy(), coll[k][keep], err_msg=k)
    # (2) vs the reference (bf16 conv stack => matched, not identical)
    rate = _iou_match_rate(res['boxes'].cpu().numpy(), g['final.boxes'])
    print('stitch: detections', len(res['scores']), 'reference', len(g['final.scores']), 'match rate', rate)
    assert rate > .85  # measured 0.93


@pytest.mark.parametrize('name', list(MODEL_SPECS))
def test_fp32_path_end_to_end_parity(dev, name):
    """North-star parity statement, checked on the fp32 verification path (precision='fp32'): the WHOLE HIP path
    (conv graph + heads + decode + refinement + NMS) against the reference's fp32 CPU forward on identical inputs and
    weights: head maps to ~1e-5 relative, identical threshold / NMS index sets, contour coordinates within 1e-4."""
    model, g = build(name, dev)
    model.precision = 'fp32'
    x = torch.as_tensor(g['x']).to(dev)
    s, l, r, f = [t.cpu().numpy() for t in model.core_forward(x)]
    exp = dict(scores=torch.sigmoid(torch.as_tensor(g['core.scores'])).numpy(), locations=g['core.locations'],
               refinement=g['core.refinement'], fourier=g['core.fourier'])
    for key, got in (('scores', s), ('locations', l), ('refinement', r), ('fourier', f)):
        e = exp[key]
        err = np.abs(got - e).max()
        print(name, key, 'max abs err', err, 'ref max', np.abs(e).max())
        np.testing.assert_allclose(got, e, rtol=2e-4, atol=2e-4 * max(1., float(np.abs(e).max())), err_msg=key)
    n = x.shape[0]
    # index sets equal; refined contours / boxes / scores within 1e-4; the raw (un-rounded, x2..x4 up-scaled)
    # regression outputs carry the conv stack's fp32 summation-order noise: 5e-4 px on coordinates of O(100) px
    # (flip_frac: the thick fixtures hold 100-400 proposals x 32 samples x 4 refinement iterations -- a handful of coordinates
    # sit within fp32 summation-order noise of x.5, where local_refinement's pixel snap is discontinuous: measured 2 of 12480
    # on CpnResNet50FPN; same allowance as the other whole-forward fp32 tests below)
    check_exact('nms', model(x), g, n, raw_atol=5e-4, flip_frac=1e-3)
    check_exact('nonms', model(x, nms=False), g, n, raw_atol=5e-4, flip_frac=1e-3)
    check_exact('offs', model(x, offsets=torch.as_tensor(g['offsets'])), g, n, raw_atol=5e-4, flip_frac=1e-3)


def test_tile_loops_without_refinement_take_one_offset_back(dev):
    """with ``refinement_iterations == 0`` CPN.forward adds the tile offset to the contours twice (the reference's
    aliasing, models/cpn.py:655-699, reproduced on purpose); the tile loops correct it, so that contours, boxes and the
    border rule agree again: every contour's bounding box must be its box (boxes receive the offset once)."""
    import warnings
    from celldetection_amd import inference
    model, g = build('CpnU22', dev, fixture='stitch.npz')
    model.refinement_iterations = 0
    img = torch.as_tensor(g['img']).to(dev)
    crop, stride = tuple(int(i) for i in g['crop']), tuple(int(i) for i in g['stride'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        res = inference.tiled_inference(model, img, crop, stride, batch_size=4, border_removal=int(g['border']))
        til = model.forward_tiled(img if img.ndim == 4 else img[None], crop_size=crop, stride=stride)
    assert res['scores'].shape[0] > 5
    for con, box in ((res['contours'], res['boxes']), (til['contours'][0], til['boxes'][0])):
        H, W = img.shape[-2:]
        assert con.shape[0] > 0 and con[..., 0].min() >= -1 and con[..., 0].max() <= W and con[..., 1].max() <= H
        assert torch.allclose(con.min(1).values, box[:, :2], atol=1e-3) and torch.allclose(con.max(1).values, box[:, 2:], atol=1e-3)
    # the switch for bit-parity with the reference script: contours keep the doubled offset, i.e. every tile but
    # the one at the origin returns contours that no longer fit their boxes
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)
        bug = inference.tiled_inference(model, img, crop, stride, batch_size=4, border_removal=int(g['border']),
                                        bug_compatible_offsets=True)
    d = (bug['contours'].min(1).values - bug['boxes'][:, :2]).abs().max(1).values
    assert bug['scores'].shape[0] > 0 and (d > 1.).any()
    tile = (img if img.ndim == 4 else img[None])[..., :crop[0], :crop[1]]
    y = model(tile, offsets=torch.tensor([[100, 50]]))  # CPN.forward itself stays bug-compatible
    assert not torch.allclose(y['contours'][0].min(1).values, y['boxes'][0][:, :2], atol=1.)


def test_forward_tiled_and_mask(dev):
    """In-model tiling (LitCpn.forward_tiled semantics) and mask handling of the slide loop."""
    import cpn_oracle as orc
    from celldetection_amd import inference
    model, g = build('CpnU22', dev, fixture='stitch.npz')
    img = torch.as_tensor(g['img']).to(dev)  # [1, 3, 160, 224]
    res = model.forward_tiled(img, crop_size=96, stride=64, border_removal=6)
    assert list(res.keys()) == ['contours', 'scores', 'boxes'] and len(res['contours']) == 1
    # oracle-side stitching of the same per-tile GPU outputs
    slices, _, shape = orc.get_tiling_slices((160, 224), (96, 96), (64, 64))
    cons, scos, boxs = [], [], []
    for idx, ((h0, h1), (w0, w1)) in enumerate(slices):
        offs = torch.tensor([[w0, h0]])
        y = model(img[..., h0:h1, w0:w1], offsets=offs)
        h_i, w_i = np.unravel_index(idx, shape)
        con, box, sco = (y[k][0].cpu().numpy() for k in ('contours', 'boxes', 'scores'))
        keep = ((box[:, 2] - box[:, 0]) >= 1.) & ((box[:, 3] - box[:, 1]) >= 1.)
        keep &= orc.remove_border_contours(con, (96, 96), 6, top=h_i > 0, right=w_i < shape[1] - 1,
                                           bottom=h_i < shape[0] - 1, left=w_i > 0,
                                           offsets=-offs[0].numpy().astype(np.float32))
        cons.append(con[keep]), scos.append(sco[keep]), boxs.append(box[keep])
    con, sco, box = np.concatenate(cons), np.concatenate(scos), np.concatenate(boxs)
    keep = orc.nms(box, sco, model.nms_thresh)
    np.testing.assert_array_equal(res['contours'][0].cpu().numpy(), con[keep])
    np.testing.assert_array_equal(res['scores'][0].cpu().numpy(), sco[keep])
    # mask: an empty mask skips every tile; a half mask only yields detections on that half (+ tile context)
    mask = torch.zeros(160, 224, device=dev)
    out = inference.tiled_inference(model, img, (96, 96), (64, 64), mask=mask)
    assert out['scores'].numel() == 0
    mask[:, :112] = 1
    out = inference.tiled_inference(model, img, (96, 96), (64, 64), mask=mask)
    full = inference.tiled_inference(model, img, (96, 96), (64, 64))
    assert 0 < out['scores'].numel() <= full['scores'].numel()
    assert float(out['locations'][:, 0].max()) < 112 + 8
    # point mask: seeds force detections (scores_lower_bound = 1 at the seed pixels); exclusive -> only the seeds
    pm = torch.zeros(160, 224, device=dev)
    pm[40:44, 60:64] = 1  # 4x4 blobs: the bilinear resize to the stride-2 score grid keeps 1.0 inside
    pm[120:124, 180:184] = 1
    seeded = inference.tiled_inference(model, img, (96, 96), (64, 64), point_mask=pm, stitching_rule='')
    only = inference.tiled_inference(model, img, (96, 96), (64, 64), point_mask=pm, point_mask_exclusive=True,
                                     stitching_rule='')
    assert only['scores'].numel() >= 2 and bool((only['scores'] == 1).all())
    assert seeded['scores'].numel() > only['scores'].numel()


# ---- CPN.forward variants: bucketed refinement, uncertainty head, multi-class scores, head options --------------------
def _variant_maps(g, dev, multi):
    sc = torch.as_tensor(g['core.scores'])
    maps = [(sc if multi else torch.sigmoid(sc)).to(dev), torch.as_tensor(g['core.locations']).to(dev),
            torch.as_tensor(g['core.refinement']).to(dev), torch.as_tensor(g['core.fourier']).to(dev)]
    unc = torch.as_tensor(g['core.uncertainty']).to(dev) if 'core.uncertainty' in g.files else None
    return maps, unc


@pytest.mark.parametrize('name', list(VARIANT_SPECS))
def test_variant_postprocess_on_reference_head_maps(dev, name):
    """Variant post-processing (class softmax/argmax, certainty filter, uncertainty-weighted NMS, bucketed
    refinement) on the reference's head maps: index sets equal, values within 1e-4."""
    model, g = build(name, dev)
    x = g['x']
    n, size = x.shape[0], tuple(x.shape[-2:])
    maps, unc = _variant_maps(g, dev, multi=model.score_channels > 1)
    check_exact('nms', model.postprocess(*maps, size, uncertainty=unc), g, n)
    check_exact('nonms', model.postprocess(*maps, size, nms=False, uncertainty=unc), g, n)
    check_exact('offs', model.postprocess(*maps, size, offsets=torch.as_tensor(g['offsets']), uncertainty=unc), g, n)
    y = model.postprocess(*maps, size, uncertainty=unc,
                          scores_upper_bound=torch.as_tensor(g['scores_upper_bound']).to(dev),
                          scores_lower_bound=torch.as_tensor(g['scores_lower_bound']).to(dev))
    check_exact('bounds', y, g, n)


@pytest.mark.parametrize('name', list(VARIANT_SPECS))
def test_variant_fp32_end_to_end_and_bf16_stack(dev, name):
    model, g = build(name, dev)
    x = torch.as_tensor(g['x']).to(dev)
    multi = model.score_channels > 1
    exp = dict(scores=torch.as_tensor(g['core.scores']) if multi else torch.sigmoid(torch.as_tensor(g['core.scores'])),
               locations=torch.as_tensor(g['core.locations']), refinement=torch.as_tensor(g['core.refinement']),
               fourier=torch.as_tensor(g['core.fourier']))
    if 'core.uncertainty' in g.files:
        exp['uncertainty'] = torch.as_tensor(g['core.uncertainty'])
    for precision, tol in (('bf16', None), ('fp32', 2e-4)):
        model.precision = precision
        s, l, r, f = [t.cpu() for t in model.core_forward(x)]
        got = dict(scores=s, locations=l, refinement=r, fourier=f)
        if 'uncertainty' in exp:
            got['uncertainty'] = model._last_uncertainty.cpu()
        for key, e in exp.items():
            assert got[key].shape == e.shape, (key, got[key].shape, e.shape)
            rel = ((got[key] - e).norm() / (e.norm() + 1e-12)).item()
            print(name, precision, key, f'relL2 {rel:.3e}')
            bound = bf16_bounds(name)[0][key] if tol is None else tol  # bf16: 2 x the error measured on the MI355X
            assert rel < bound, (name, precision, key, rel, bound)
    n = x.shape[0]  # fp32 path end to end
    check_exact('nms', model(x), g, n, raw_atol=5e-4, flip_frac=1e-3)
    check_exact('nonms', model(x, nms=False), g, n, raw_atol=5e-4, flip_frac=1e-3)
    check_exact('offs', model(x, offsets=torch.as_tensor(g['offsets'])), g, n, raw_atol=5e-4, flip_frac=1e-3)


def test_ensemble_inference_and_inference_wrapper(dev):
    """Two models on one slide: per-model tiled results -> concat -> box voting -> final NMS
    (cpn_inference.py:419-427), checked against the oracle's voting + NMS applied to the same per-model results."""
    import cpn_oracle as orc
    import celldetection_amd as cda
    from celldetection_amd import inference
    from celldetection_amd.synth import synth_state_dict
    m0, g = build('CpnU22', dev, fixture='stitch.npz')
    m1, _ = build('CpnU22', dev, fixture='stitch.npz')
    sd = m1.state_dict()  # second ensemble member: perturbed heads -> overlapping but different detections
    for k in ('core.fourier_head.block.4.weight', 'core.location_head.block.4.weight'):
        sd[k] = sd[k] * 1.05
    m1.load_state_dict(sd)
    m1 = m1.to(dev)
    img = torch.as_tensor(g['img']).to(dev)
    kw = dict(crop_size=(96, 96), strides=(64, 64))
    parts = [inference.tiled_inference(m, img, **kw) for m in (m0, m1)]
    boxes = torch.cat([p['boxes'] for p in parts]).cpu().numpy()
    scores = torch.cat([p['scores'] for p in parts]).cpu().numpy()
    for min_vote in (1, 1.5):
        res = inference.ensemble_inference([m0, m1], img, min_vote=min_vote, **kw)
        b, s = boxes, scores
        if min_vote > 1:
            keep, votes = orc.filter_by_box_voting(b, m1.nms_thresh, min_vote)
            b, s = b[keep], s[keep]
            assert 'votes' in res
        keep = orc.nms(b, s, m1.nms_thresh)
        np.testing.assert_array_equal(res['boxes'].cpu().numpy(), b[keep])
        np.testing.assert_array_equal(res['scores'].cpu().numpy(), s[keep])
    assert 0 < res['scores'].numel() < len(scores)
    # cd.models.Inference mirror: HWC-free array in, numpy dict out
    out = cda.models.Inference(m0)(g['img'][0])
    assert isinstance(out['contours'][0], np.ndarray) and out['box_uncertainties'] is None
    y = m0(img)
    np.testing.assert_array_equal(out['boxes'][0], y['boxes'][0].cpu().numpy())


def test_hip_graph_replay_equals_eager_launches(dev, monkeypatch):
    """The conv graph of a shape that is seen repeatedly is replayed as ONE hipGraph from a ring of GRAPH_SLOTS captured
    instances (cpn._Engine.run): every output must be bit-identical to the eager launches, results handed out by the public
    ``core_forward`` must survive later runs (they are cloned out of the slot), and the pipelined tile loop -- which reads
    the slots from a second stream -- must equal per-batch ``forward``."""
    model, g = build('CpnU22_wide', dev)
    model.sparse_heads = False  # ONE engine behind forward() and the public core_forward() (the default 'auto' keeps a gated
    # engine for the forward paths next to the dense one: its replays are covered by tests/test_gpu_sparse_heads.py)
    xs = [torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(s)).to(dev) for s in range(7)]
    monkeypatch.setenv('CPN_HIP_GRAPH', '0')
    eager = [model(x) for x in xs]
    eager_maps = [tuple(t.clone() for t in model.core_forward(x)) for x in xs[:2]]
    assert not model.engine(dev)._graphs
    monkeypatch.setenv('CPN_HIP_GRAPH', '1')
    held = model.core_forward(xs[0])          # first sighting of the shape: eager
    held2 = model.core_forward(xs[1])         # second: captured into slot 0 and cloned out
    replay = [model(x) for x in xs]           # slots 1, 2 captured, then pure replays
    eng = model.engine(dev)
    assert len(eng._graphs) == 1 and len(next(iter(eng._graphs.values()))['slots']) == eng.GRAPH_SLOTS and not eng._graph_broken
    for a, b in zip(eager, replay):
        for k in a:
            if a[k] is not None:
                for p, q in zip(a[k], b[k]):
                    assert torch.equal(p, q), k
    for got, want in ((held, eager_maps[0]), (held2, eager_maps[1])):
        for p, q in zip(got, want):
            assert torch.equal(p, q)
    pipe = list(model.forward_pipelined(iter(xs)))
    for a, b in zip(eager, pipe):
        for k in a:
            if a[k] is not None:
                for p, q in zip(a[k], b[k]):
                    assert torch.equal(p, q), k
    bad = xs[0].clone()
    bad[0, 0, 0, 0] = 2.                      # the range flag lives in the slot and is re-armed by every replay
    with pytest.raises(AssertionError):
        model(bad)
    model(xs[0])


def test_forward_pipelined_equals_forward(dev):
    """Two-stream throughput mode returns exactly what forward() returns, batch by batch (incl. per-batch kwargs)."""
    model, g = build('CpnU22', dev)
    xs = [torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(s)).to(dev) for s in range(5)]
    offs = [torch.tensor([[10 * i, 3 * i], [0, i]]) for i in range(5)]
    ref = [model(x, offsets=o) for x, o in zip(xs, offs)]
    got = list(model.forward_pipelined(((x, dict(offsets=o)) for x, o in zip(xs, offs))))
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        for k in KEYS:
            for ta, tb in zip(a[k], b[k]):
                assert torch.equal(ta, tb), k
    assert list(model.forward_pipelined([])) == []


def test_full_size_properties(dev):
    """BASELINE.json configs[2] at full size (CpnResNeXt101UNet, 16 x 3x512x512, synthetic ginoro-shaped weights):
    size-independent properties of the whole HIP path -- the oracle needs ~1 s per tile here, so only one tile is
    compared with it (IoU-matched, bf16) and the rest is checked through invariants."""
    import sys
    import cpn_oracle as orc
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_model
    model, sd = build_model('CpnResNeXt101UNet', dev)
    x = torch.rand(16, 3, 512, 512, generator=torch.Generator().manual_seed(100)).to(dev)
    y = model(x)
    n_det = [len(s) for s in y['scores']]
    assert min(n_det) > 20, n_det
    # determinism
    y2 = model(x)
    for k in KEYS:
        for a, b in zip(y[k], y2[k]):
            assert torch.equal(a, b), k
    # batch independence: a tile alone == the same tile inside the batch of 16 (other tile shapes / grid sizes are
    # chosen for N=1, the per-pixel accumulation order must not change)
    for i in (0, 7):
        yi = model(x[i:i + 1])
        for k in KEYS:
            assert torch.equal(yi[k][0], y[k][i]), (k, i)
    # uint8 input == float input / 255 (LitBase.prepare_inputs)
    xu = (x[:2] * 255).round().to(torch.uint8)
    ya, yb = model(xu), model(xu.float() / 255)
    for k in KEYS:
        for a, b in zip(ya[k], yb[k]):
            assert torch.equal(a, b), k
    # per image: scores sorted descending (NMS keep order) and above the threshold, boxes = min/max of the contours,
    # contours inside the image, NMS idempotent (no kept pair overlaps above the threshold)
    for i in range(16):
        s, c, b = y['scores'][i], y['contours'][i], y['boxes'][i]
        assert bool((s[:-1] >= s[1:]).all()) and float(s.min()) > model.score_thresh
        assert torch.equal(b, torch.cat((c.min(1).values, c.max(1).values), 1))
        assert float(c.min()) >= 0 and float(c[..., 0].max()) <= 511 and float(c[..., 1].max()) <= 511
    for i in (0, 15):
        b, s = y['boxes'][i].cpu().numpy(), y['scores'][i].cpu().numpy()
        np.testing.assert_array_equal(orc.nms(b, s, model.nms_thresh), np.arange(len(s)))
    # offsets translate exactly (integer offsets, fp32 adds)
    offs = torch.tensor([[384 * i, 768] for i in range(16)])
    yo = model(x, offsets=offs)
    for i in (3, 12):
        assert torch.equal(yo['boxes'][i], y['boxes'][i] + offs[i].repeat(2).to(dev))
    # one tile against the fp32 CPU oracle (bf16 conv stack => IoU-matched proposals)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = orc.cpn_forward({k: v.cpu() for k, v in sd.items()}, x[:1].cpu(), nms=False)
    got = model(x[:1], nms=False)
    rate = _iou_match_rate(got['boxes'][0].cpu().numpy(), ref['boxes'][0])
    n_ref, n_got = len(ref['scores'][0]), len(got['scores'][0])
    print('full size tile 0: proposals', n_got, 'oracle', n_ref, 'IoU>0.5 match rate', rate)
    assert abs(n_ref - n_got) <= 0.1 * n_ref and rate > .9
    # THE NORTH-STAR STATEMENT AT THE HEADLINE CONFIG: the fp32 verification path on a full-width 3x512x512
    # tile against the fp32 CPU oracle -- identical proposal and NMS index sets, contours within 1e-4 (snap flips reported)
    model.precision = 'fp32'
    north_star_check('configs[2] tile 0, proposals', model(x[:1], nms=False), ref)
    ref_nms = orc.cpn_forward({k: v.cpu() for k, v in sd.items()}, x[:1].cpu(), nms=True)
    assert len(ref_nms['scores'][0]) > 20
    north_star_check('configs[2] tile 0, after NMS', model(x[:1], nms=True), ref_nms)
    # fp8 graph at full size: proposals IoU-matched against the fp32 oracle's (e4m3 activations: 3-bit mantissa)
    model.precision = 'fp8'
    model.calibrate_fp8(x[:2])
    got8 = model(x[:1], nms=False)
    rate8 = _iou_match_rate(got8['boxes'][0].cpu().numpy(), ref['boxes'][0])
    print('full size tile 0 fp8: proposals', len(got8['scores'][0]), 'IoU>0.5 match rate', rate8)
    assert abs(n_ref - len(got8['scores'][0])) <= 0.2 * n_ref and rate8 > .8  # measured 0.897


@pytest.mark.parametrize('name', ['CpnU22', 'CpnU22_wide', 'CpnResNeXt101UNet', 'CpnResNet18FPN', 'CpnResNet50FPN', 'CpnU22_headact',
                                  'CpnResNet18FPN_fuse5'])
def test_fp8_precision_vs_reference_maps(dev, name):
    """fp8 (e4m3 activations + weights, K=64 scaled MFMA) conv stack against the reference's fp32 head maps: e4m3 has a
    3-bit mantissa (2^-4 relative rounding per value), so the check is a relative L2 bound per head map plus an
    IoU-matched proposal comparison -- there is no reference fp8 path to be exact against."""
    model, g = build(name, dev)
    x = torch.as_tensor(g['x']).to(dev)
    model.precision = 'fp8'
    scales = model.calibrate_fp8(x)
    assert len(scales) == len(model.plan_for('fp8').tensors) and min(scales) > 0
    s, l, r, f = [t.cpu() for t in model.core_forward(x)]
    exp = dict(scores=torch.sigmoid(torch.as_tensor(g['core.scores'])), locations=torch.as_tensor(g['core.locations']),
               refinement=torch.as_tensor(g['core.refinement']), fourier=torch.as_tensor(g['core.fourier']))
    rep = {}
    for key, got in (('scores', s), ('locations', l), ('refinement', r), ('fourier', f)):
        e = exp[key]
        assert got.shape == e.shape and torch.isfinite(got).all(), key
        rep[key] = ((got - e).norm() / (e.norm() + 1e-12)).item()
    print(name, 'fp8 relL2', {k: f'{v:.3f}' for k, v in rep.items()})
    # (no absolute bound here: measured 0.08 .. 0.53 on the synthetic-weight tiny models, a number that would also pass a
    # broken head -- VERDICT r4.  The gate is the comparison with the CPU simulation of the SAME algorithm at the end of this
    # test, map by map, plus the IoU-matched proposals.)
    y = model(x, nms=False)
    rates = [_iou_match_rate(y['boxes'][i].cpu().numpy(), g[f'nonms.boxes.{i}']) for i in range(x.shape[0])]
    print(name, 'fp8 proposal IoU>0.5 match rates', rates)
    # measured 0.89 .. 0.99; the fused-feature toy model: 0.74 (its two-feature sibling 0.65, Fuse2d over three 0.87 -- the CPU
    # simulation below shows the same error level map by map, so it is e4m3 on this model, not the split of the fusion conv)
    assert min(rates) > (.65 if name.endswith('_fuse5') else .8), rates
    # the same fp8 algorithm restated on the CPU (oracle/fp8_sim.py: identical codes, scales and weights).  A deep
    # quantised graph is chaotic -- one e4m3 rounding that differs because of the fp32 summation order shifts ~1
    # rounding decision in the next layer, so after 30..120 layers the two noise realisations are decorrelated
    # (HIP vs simulation differ by about as much as either differs from fp32; the kernels themselves are pinned per
    # layer by tests/test_gpu_kernels.py::test_conv_fp8_vs_dequantised_reference).  What must hold is that the HIP path is
    # not noisier than the restated algorithm: same error level against the reference's fp32 maps, map by map.
    import fp8_sim
    from celldetection_amd import _lib, graph
    eff = []
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    graph.pack(model.plan_for('fp8'), sd, 'cpu', precision='fp8', act_scales=scales, effective_weights=eff)
    sim = fp8_sim.simulate(model.plan_for('fp8'), sd, eff, scales, torch.as_tensor(g['x']))
    for key, got, idx in (('scores', s, _lib.OUT_SCORES), ('locations', l, _lib.OUT_LOCATIONS),
                          ('fourier', f, _lib.OUT_FOURIER), ('refinement', r, _lib.OUT_REFINEMENT)):
        e_sim = ((sim[idx] - exp[key]).norm() / (exp[key].norm() + 1e-12)).item()
        d = ((got - sim[idx]).norm() / (sim[idx].norm() + 1e-12)).item()
        print(f'{name} {key}: fp8 error vs fp32 reference: HIP {rep[key]:.3f}, CPU simulation {e_sim:.3f}; HIP vs sim {d:.3f}')
        assert rep[key] < 1.5 * e_sim + 0.03, (key, rep[key], e_sim)


@pytest.mark.parametrize('name', list(SIZE_SPECS))
def test_arbitrary_input_sizes(dev, name):
    """Inputs whose sizes are not multiples of 32 (75x101, 100x140, 300x300): tensor sizes follow the reference's
    modules (floor-mode conv/pool, nearest resize to the lateral's size, bilinear resize of the refinement features to
    the input size).  (b) post-processing on the reference's maps exact; (f) fp32 path end to end at the north-star
    tolerance; (a) bf16 MFMA stack within the bf16 tolerance."""
    model, g = build(name, dev)
    x = torch.as_tensor(g['x']).to(dev)
    n, size = x.shape[0], tuple(x.shape[-2:])
    maps = [torch.sigmoid(torch.as_tensor(g['core.scores'])).to(dev), torch.as_tensor(g['core.locations']).to(dev),
            torch.as_tensor(g['core.refinement']).to(dev), torch.as_tensor(g['core.fourier']).to(dev)]
    check_exact('nms', model.postprocess(*maps, size), g, n)
    check_exact('offs', model.postprocess(*maps, size, offsets=torch.as_tensor(g['offsets'])), g, n)
    check_exact('bounds', model.postprocess(*maps, size, scores_upper_bound=torch.as_tensor(g['scores_upper_bound']).to(dev),
                                            scores_lower_bound=torch.as_tensor(g['scores_lower_bound']).to(dev)), g, n)
    exp = dict(scores=torch.sigmoid(torch.as_tensor(g['core.scores'])), locations=torch.as_tensor(g['core.locations']),
               refinement=torch.as_tensor(g['core.refinement']), fourier=torch.as_tensor(g['core.fourier']))
    for precision, tol in (('bf16', None), ('fp32', 2e-4)):
        model.precision = precision
        got = dict(zip(('scores', 'locations', 'refinement', 'fourier'), [t.cpu() for t in model.core_forward(x)]))
        for key, e in exp.items():
            assert got[key].shape == e.shape, (key, got[key].shape, e.shape)
            rel = ((got[key] - e).norm() / (e.norm() + 1e-12)).item()
            print(name, precision, key, f'relL2 {rel:.3e}')
            bound = bf16_bounds(name)[0][key] if tol is None else tol  # bf16: 2 x the error measured on the MI355X
            assert rel < bound, (name, precision, key, rel, bound)
    check_exact('nms', model(x), g, n, raw_atol=5e-4, flip_frac=1e-3)  # fp32 path, whole forward
    check_exact('nonms', model(x, nms=False), g, n, raw_atol=5e-4, flip_frac=1e-3)


def test_slide_smaller_than_crop_and_ragged_tiles(dev):
    """tiled_inference on a slide smaller than the crop (one tile of the slide's own size, no padding needed) and on a
    slide whose size is no multiple of anything: same detections as the direct forward / as the oracle's stitching."""
    import cpn_oracle as orc
    from celldetection_amd import inference
    model, g = build('CpnU22_300', dev)
    x = torch.as_tensor(g['x']).to(dev)  # [1, 3, 300, 300]
    res = inference.tiled_inference(model, x, crop_size=(512, 512), strides=(384, 384))
    y = model(x)
    for k in inference.KEYS:  # single tile, no neighbours: nothing is border-filtered, global NMS is idempotent
        assert torch.equal(res[k], y[k][0]), k
    res = inference.tiled_inference(model, x[..., :277, :300], crop_size=(128, 160), strides=(96, 101), batch_size=3)
    slices, overlaps, shape = orc.get_tiling_slices((277, 300), (128, 160), (96, 101))
    coll = {}
    for idx, ((h0, h1), (w0, w1)) in enumerate(slices):
        offs = torch.tensor([[w0, h0]])
        yt = model(x[..., h0:h1, w0:w1], offsets=offs)
        h_i, w_i = np.unravel_index(idx, shape)
        con = yt['contours'][0].cpu().numpy()
        keep = orc.remove_border_contours(con, (h1 - h0, w1 - w0), 4, top=h_i > 0, right=w_i < shape[1] - 1,
                                          bottom=h_i < shape[0] - 1, left=w_i > 0, offsets=-offs[0].numpy().astype(np.float32))
        for k in inference.KEYS:
            v = yt[k][0].cpu().numpy()[keep]
            coll[k] = np.concatenate((coll[k], v)) if k in coll else v
    keep = orc.nms(coll['boxes'], coll['scores'], model.nms_thresh)
    for k in inference.KEYS:
        np.testing.assert_array_equal(res[k].cpu().numpy(), coll[k][keep], err_msg=k)


def test_stitching_with_cross_tile_duplicates(dev):
    """Slide loop on the GPU (batched border kernel, row selection, packing, global NMS) fed with the fixture's per-tile
    detections that contain cross-tile duplicates: bit-identical to what the reference's functions produced, the
    global NMS removes > 10 % (incl. the ex_br rule's keep masks and the binned NMS on the same set)."""
    import stitch_fixture as sf
    from celldetection_amd import inference, ops
    g = sf.load()
    H, W = (int(i) for i in g['size'])
    img = torch.zeros(1, 3, H, W, device=dev)
    kw = dict(crop_size=tuple(int(i) for i in g['crop']), strides=tuple(int(i) for i in g['stride']), batch_size=5,
              border_removal=int(g['border']), forward_fn=sf.forward_fn(g, dev))
    res = inference.tiled_inference(sf.StubModel(), img, rank=0, world_size=1, **kw)
    assert res['scores'].shape[0] <= 0.9 * int(g['pre_nms_count'])
    for k in inference.KEYS:
        np.testing.assert_array_equal(res[k].cpu().numpy(), g[f'final.{k}'], err_msg=k)
    pre = inference.tiled_inference(sf.StubModel(), img, rank=0, world_size=1, stitching_rule='', **kw)
    assert pre['scores'].shape[0] == int(g['pre_nms_count'])
    keep = ops.nms_binned(pre['boxes'], pre['scores'], float(g['nms_thresh']))
    np.testing.assert_array_equal(pre['boxes'][keep].cpu().numpy(), g['final.boxes'])
    exbr = inference.tiled_inference(sf.StubModel(), img, rank=0, world_size=1, stitching_rule='ex_br', **kw)
    n_exp = sum(int(g[f'tile{i}.keep_border_exbr'].sum()) for i in range(int(g['n_tiles'])))
    assert exbr['scores'].shape[0] == n_exp


def _invariants(model, y, x, size):
    """Size-independent properties of a forward result (per image): NMS order / threshold, boxes = contour extrema,
    contours inside the image."""
    H, W = size
    for i in range(x.shape[0]):
        s, c, b = y['scores'][i], y['contours'][i], y['boxes'][i]
        if s.numel() == 0:
            continue
        assert bool((s[:-1] >= s[1:]).all()) and float(s.min()) > model.score_thresh
        assert torch.equal(b, torch.cat((c.min(1).values, c.max(1).values), 1))
        assert float(c.min()) >= 0 and float(c[..., 0].max()) <= W - 1 and float(c[..., 1].max()) <= H - 1


def test_full_size_properties_config1_resnet18fpn(dev):
    """BASELINE.json configs[1] at full size: CpnResNet18FPN, bf16, 8 x 3x512x512 (refinement head reads its 256-channel
    features through the fused bilinear loader).  Invariants on the batch + one tile against the fp32 CPU oracle."""
    import sys
    import cpn_oracle as orc
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_model
    model, sd = build_model('CpnResNet18FPN', dev)
    x = torch.rand(8, 3, 512, 512, generator=torch.Generator().manual_seed(101)).to(dev)
    y = model(x)
    assert min(len(s) for s in y['scores']) > 5, [len(s) for s in y['scores']]
    y2 = model(x)
    for k in KEYS:  # determinism
        for a, b in zip(y[k], y2[k]):
            assert torch.equal(a, b), k
    yi = model(x[3:4])  # batch independence
    for k in KEYS:
        assert torch.equal(yi[k][0], y[k][3]), k
    _invariants(model, y, x, (512, 512))
    b, s = y['boxes'][0].cpu().numpy(), y['scores'][0].cpu().numpy()
    np.testing.assert_array_equal(orc.nms(b, s, model.nms_thresh), np.arange(len(s)))  # NMS idempotent
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = orc.cpn_forward({k: v.cpu() for k, v in sd.items()}, x[:1].cpu(), nms=False)
    got = model(x[:1], nms=False)
    rate = _iou_match_rate(got['boxes'][0].cpu().numpy(), ref['boxes'][0])
    n_ref, n_got = len(ref['scores'][0]), len(got['scores'][0])
    print('configs[1] tile 0: proposals', n_got, 'oracle', n_ref, 'IoU>0.5 match rate', rate)
    assert abs(n_ref - n_got) <= max(3, 0.1 * n_ref) and rate > .9
    # the fp32 verification path on the same tile: the north-star statement in the same form as at configs[2]
    model.precision = 'fp32'
    north_star_check('configs[1] tile 0, proposals', model(x[:1], nms=False), ref)
    ref_nms = orc.cpn_forward({k: v.cpu() for k, v in sd.items()}, x[:1].cpu(), nms=True)
    north_star_check('configs[1] tile 0, after NMS', model(x[:1], nms=True), ref_nms)


def test_full_width_cpnu22_config0(dev):
    """BASELINE.json configs[0] on the HIP path: the FULL-WIDTH ``CpnU22(3)`` (31.6 M parameters, base_channels 64) on one
    3x256x256 tile -- bf16 (the product precision) IoU-matched against the fp32 CPU oracle, then the north-star statement on
    the fp32 verification path, and the default-init model (0 detections) through the whole path."""
    import sys
    import cpn_oracle as orc
    import celldetection_amd as cda
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_model
    model, sd = build_model('CpnU22', dev, tile=256)
    assert sum(v.numel() for k, v in model.state_dict().items() if 'num_batches' not in k and 'running' not in k
               and k != 'order_weights') > 31e6
    torch.manual_seed(0)
    x = torch.rand(1, 3, 256, 256).to(dev)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    ref = orc.cpn_forward(sd_cpu, x.cpu(), nms=False)
    n_ref = len(ref['scores'][0])
    got = model(x, nms=False)
    rate = _iou_match_rate(got['boxes'][0].cpu().numpy(), ref['boxes'][0])
    print('configs[0] full-width CpnU22 bf16: proposals', len(got['scores'][0]), 'oracle', n_ref, 'IoU>0.5 match rate', rate)
    # (contours of ~3 px radius at stride 1: an IoU of 0.5 is a displacement of one pixel -- measured 0.85; the tiny-model
    # fixtures with larger objects ask for 0.9)
    assert n_ref > 50 and abs(n_ref - len(got['scores'][0])) <= max(3, 0.1 * n_ref) and rate > .8
    _invariants(model, model(x), x, (256, 256))
    model.precision = 'fp32'
    north_star_check('configs[0] CpnU22, proposals', model(x, nms=False), ref)
    north_star_check('configs[0] CpnU22, after NMS', model(x, nms=True), orc.cpn_forward(sd_cpu, x.cpu(), nms=True))
    # PyTorch-default initialisation, as configs[0] states it: no pixel passes the 0.9 threshold -> empty lists of the
    # contract's shapes and dtypes (a20)
    torch.manual_seed(0)
    plain = cda.models.CpnU22(3).to(dev)
    y = plain(x)
    assert list(y) == list(model(x)) and len(y['contours']) == 1
    k = len(y['scores'][0])
    print('configs[0] default-init CpnU22: detections', k)
    assert k == 0, 'SURVEY 8d config 1: the default initialisation yields no detections'
    assert tuple(y['contours'][0].shape) == (0, 32, 2) and y['classes'][0].dtype == torch.int64 and y['boxes'][0].shape == (0, 4)


def test_full_size_properties_config4_resnet50fpn_fp8(dev):
    """BASELINE.json configs[4], the per-GPU share: CpnResNet50FPN, fp8 (e4m3), 8 x 3x1024x1024.  Invariants on the
    batch, bf16 and fp8 proposals of one tile IoU-matched against the fp32 CPU oracle."""
    import sys
    import cpn_oracle as orc
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_model
    model, sd = build_model('CpnResNet50FPN', dev, tile=1024, calib_tiles=1)
    x = torch.rand(8, 3, 1024, 1024, generator=torch.Generator().manual_seed(102)).to(dev)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = orc.cpn_forward({k: v.cpu() for k, v in sd.items()}, x[:1].cpu(), nms=False)
    n_ref = len(ref['scores'][0])
    got = model(x[:1], nms=False)
    rate = _iou_match_rate(got['boxes'][0].cpu().numpy(), ref['boxes'][0])
    print('configs[4] tile 0 bf16: proposals', len(got['scores'][0]), 'oracle', n_ref, 'IoU>0.5 match rate', rate)
    assert abs(n_ref - len(got['scores'][0])) <= max(3, 0.1 * n_ref) and rate > .9
    model.precision = 'fp8'
    model.calibrate_fp8(x[:1])
    y = model(x)
    assert min(len(s) for s in y['scores']) > 5
    _invariants(model, y, x, (1024, 1024))
    y2 = model(x)
    for k in KEYS:
        for a, b in zip(y[k], y2[k]):
            assert torch.equal(a, b), k
    got8 = model(x[:1], nms=False)
    rate8 = _iou_match_rate(got8['boxes'][0].cpu().numpy(), ref['boxes'][0])
    print('configs[4] tile 0 fp8: proposals', len(got8['scores'][0]), 'oracle', n_ref, 'IoU>0.5 match rate', rate8)
    assert abs(n_ref - len(got8['scores'][0])) <= 0.25 * n_ref and rate8 > .75


@pytest.mark.parametrize('name', list(HEAD_SPECS))
def test_head_options(dev, name):
    """Strided ReadOut heads, heads on other decoder / encoder features, Fuse2d over two features: bf16 stack within the
    bf16 tolerance, fp32 path end to end at the north-star tolerance, post-processing on the reference's maps exact."""
    model, g = build(name, dev)
    x = torch.as_tensor(g['x']).to(dev)
    n, size = x.shape[0], tuple(x.shape[-2:])
    maps = [torch.sigmoid(torch.as_tensor(g['core.scores'])).to(dev), torch.as_tensor(g['core.locations']).to(dev),
            torch.as_tensor(g['core.refinement']).to(dev), torch.as_tensor(g['core.fourier']).to(dev)]
    check_exact('nms', model.postprocess(*maps, size), g, n)
    check_exact('offs', model.postprocess(*maps, size, offsets=torch.as_tensor(g['offsets'])), g, n)
    exp = dict(scores=torch.sigmoid(torch.as_tensor(g['core.scores'])), locations=torch.as_tensor(g['core.locations']),
               refinement=torch.as_tensor(g['core.refinement']), fourier=torch.as_tensor(g['core.fourier']))
    for precision, tol in (('bf16', None), ('fp32', 2e-4)):
        model.precision = precision
        got = dict(zip(('scores', 'locations', 'refinement', 'fourier'), [t.cpu() for t in model.core_forward(x)]))
        for key, e in exp.items():
            assert got[key].shape == e.shape, (key, got[key].shape, e.shape)
            rel = ((got[key] - e).norm() / (e.norm() + 1e-12)).item()
            print(name, precision, key, f'relL2 {rel:.3e}')
            bound = bf16_bounds(name)[0][key] if tol is None else tol  # bf16: 2 x the error measured on the MI355X
            assert rel < bound, (name, precision, key, rel, bound)
    check_exact('nms', model(x), g, n, raw_atol=5e-4, flip_frac=1e-3)  # fp32 path, whole forward


@pytest.mark.parametrize('name', ['CpnResNet18FPN', 'CpnResNet50FPN', 'CpnResNet18FPN_heads', 'CpnResNet18FPN_fuse'])
def test_bilinear_phase_refinement_head_on_golden_models(dev, name, monkeypatch):
    """The refinement head over the x2 bilinear-resized level-0 map (cpn.py:277-278 + ReadOut, commons.py:461-511) as four
    5 x 5 (k = 5: 4 x 4 ... k = 7: 5 x 5) phase convs on the low-resolution map + the frame of the conv over the resized map
    (CPN_SUBPIXEL_BL_*; forced here with CPN_BLPHASE=2 -- by default the executor takes it only where it saves MACs, which
    these 64 x 96 fixtures are too small for): the refinement map stays within the bf16 tolerance of the reference's fp32 map,
    is no farther from it than the conv over the resized map, and the two statements agree closely with each other."""
    spec = ALL_SPECS[name]
    if spec['kwargs'].get('kernel_size_refinement', 7) % 4 != 3:
        pytest.skip('k = 5: the two output phases have different supports -> the plan keeps the conv over the resized map')
    monkeypatch.setenv('CPN_BLPHASE', '0')
    m0, g = build(name, dev)
    x = torch.as_tensor(g['x']).to(dev)
    ref = torch.as_tensor(g['core.refinement'])
    old = m0.core_forward(x)[2].cpu()
    assert not any(p['gflop'] > 0 for p in m0.engine(dev).profile(x, m0.core.order, True) if isinstance(p.get('name'), str)
                   and 'refinement' in p['name'] and p['k'] == 5)
    monkeypatch.setenv('CPN_BLPHASE', '2')
    m1, _ = build(name, dev)
    new = m1.core_forward(x)[2].cpu()
    prof = m1.engine(dev).profile(x, m1.core.order, True)
    ran = [(p['k'], p['gflop'] > 0) for p in prof if 'refinement_head.block.0' in (p['name'] or '')]
    assert ran == [(7, False), (5, True), (7, True)], ran  # HEAD skipped, PHASE + FRAME executed
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    e_old, e_new, e_pair = rel(old, ref), rel(new, ref), rel(new, old)
    print(f'{name}: refinement map relL2 vs fp32 reference: resized-map conv {e_old:.3e}, phases + frame {e_new:.3e}; '
          f'between the two {e_pair:.3e}')
    assert e_old < bf16_bounds(name)[0]['refinement'] and e_new < 1.25 * e_old + 1e-3 and e_pair < 2e-2
    for a, b in zip(m1.core_forward(x)[:2], m0.core_forward(x)[:2]):  # the other head maps are untouched
        assert torch.equal(a, b)
    # the frame really comes from the frame op and the interior from the phase convs: both regions carry finite, distinct values
    assert torch.isfinite(new).all() and new[..., 8:-8, 8:-8].abs().sum() > 0 and new[..., :4, :].abs().sum() > 0


def test_bilinear_phase_refinement_head_at_configs1_size(dev, monkeypatch):
    """BASELINE configs[1] shape (CpnResNet18FPN, 512^2 tiles, 256-channel FPN): the executor picks the decomposition by itself
    (15 % frame tiles) and the refinement map agrees with the conv over the resized map to bf16 accuracy."""
    import celldetection_amd as cda
    from celldetection_amd.synth import synth_state_dict
    x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(0)).to(dev)
    outs = {}
    for mode in ('0', None):
        if mode is None:
            monkeypatch.delenv('CPN_BLPHASE', raising=False)
        else:
            monkeypatch.setenv('CPN_BLPHASE', mode)
        m = cda.models.CpnResNet18FPN(3)
        m.load_state_dict(synth_state_dict(m.state_dict(), seed=3))
        m = m.to(dev)
        outs[mode] = m.core_forward(x)[2].float().cpu()
        prof = m.engine(dev).profile(x, m.core.order, True)
        ran = [(p['k'], p['gflop'] > 0) for p in prof if 'refinement_head.block.0' in (p['name'] or '')]
        assert ran == ([(7, True), (5, False), (7, False)] if mode == '0' else [(7, False), (5, True), (7, True)]), (mode, ran)
    a, b = outs[None], outs['0']
    rel = lambda p, q: ((p - q).norm() / (q.norm() + 1e-12)).item()
    # un-calibrated synthetic weights drive the 256-channel net hard: the yardstick is the oracle's fp32 map of the first tile
    import cpn_oracle as orc
    ref = orc.core_forward({k: v.detach().cpu() for k, v in m.state_dict().items()}, x[:1].cpu())[2]
    e_new, e_old = rel(a[:1], ref), rel(b[:1], ref)
    print(f'configs[1] shape: refinement map relL2 vs the fp32 oracle: resized-map conv {e_old:.3e}, phases + frame {e_new:.3e}; '
          f'between the two {rel(a, b):.3e}')
    assert torch.isfinite(a).all() and e_new < 1.25 * e_old + 1e-3 and rel(a, b) < 2.5 * e_old


@pytest.mark.parametrize('name', ['CpnResNet18FPN', 'CpnResNet50FPN'])
def test_bilinear_phase_refinement_head_fp8(dev, name, monkeypatch):
    """fp8 plans: the resize in front of the refinement head is an op of its own (e4m3 codes); with the phase decomposition the
    four 5 x 5 phase convs read the map in FRONT of that resize, the resize writes only the border ring the frame conv reads.
    Frame pixels: same kernel, same inputs as the conv over the whole resized map -> same values; interior: no farther from
    the reference's fp32 map than the old path (one e4m3 rounding fewer: the resized map is never quantised there)."""
    maps = {}
    for mode in ('0', '2'):
        monkeypatch.setenv('CPN_BLPHASE', mode)
        m, g = build(name, dev)
        x = torch.as_tensor(g['x']).to(dev)
        m.precision = 'fp8'
        m.calibrate_fp8(x)
        maps[mode] = [t.cpu() for t in m.core_forward(x)]
        prof = m.engine(dev).profile(x, m.core.order, True)
        ran = [(p['k'], p['gflop'] > 0) for p in prof if 'refinement_head.block.0' in (p['name'] or '')]
        assert ran == ([(7, True), (5, False), (7, False)] if mode == '0' else [(7, False), (5, True), (7, True)]), (mode, ran)
    old, new = maps['0'][2], maps['2'][2]
    ref = torch.as_tensor(g['core.refinement'])
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    e_old, e_new = rel(old, ref), rel(new, ref)
    print(f'{name} fp8: refinement map relL2 vs fp32 reference: resized-map conv {e_old:.3e}, phases + frame {e_new:.3e}; '
          f'between the two {rel(new, old):.3e}')
    assert torch.isfinite(new).all() and e_new < 1.1 * e_old + 1e-2
    f = 4  # bilinear_frame(7)
    for sl in ((..., slice(0, f), slice(None)), (..., slice(-f, None), slice(None)), (..., slice(None), slice(0, f)),
               (..., slice(None), slice(-f, None))):
        assert (new[sl] - old[sl]).abs().max().item() < 1e-5, sl  # the ring-only resize feeds the frame conv the same codes
    assert (new[..., f:-f, f:-f] - old[..., f:-f, f:-f]).abs().max() > 0
    for a, b in zip(maps['0'][:2] + maps['0'][3:], maps['2'][:2] + maps['2'][3:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('name', ['CpnU22', 'CpnU22_wide', 'CpnResNeXt101UNet', 'CpnResNet50UNet', 'CpnResNeXt101UNet_odd'])
def test_fp8_subpixel_triples_of_the_unet_decoders(dev, name):
    """fp8 plans carry the sub-pixel triples of the UNet decoder convs (models/unet.py:213-224) since round 5: four 2 x 2 phase
    convs on the e4m3 top-down map write their partial sums as bf16, the 3 x 3 lateral conv adds them as a pixel-shuffled
    residual.  Against the reference's fp32 head maps the triples must not be noisier than the conv over the virtual concat
    (``model.subpixel = False``: same e4m3 operands, one accumulation), map by map; at sizes where the top-down map is not an
    exact half (75 x 101) the HEAD conv runs either way -> identical maps."""
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    maps = {}
    for sub in (False, True):
        m, g = build(name, dev)
        x = torch.as_tensor(g['x']).to(dev)
        m.subpixel = sub
        m.precision = 'fp8'
        m.calibrate_fp8(x)
        maps[sub] = [t.cpu() for t in m.core_forward(x)]
        prof = m.engine(dev).profile(x, m.core.order, True)
        ran_k2 = sum(1 for p in prof if p['op'] == 'conv' and p['k'] == 2 and p['gflop'] > 0)
        exact = all(v % 32 == 0 for v in x.shape[-2:])
        assert ran_k2 == (4 if (sub and exact) else 0), (name, sub, ran_k2)  # four decoder levels with a lateral
    ref = (torch.sigmoid(torch.as_tensor(g['core.scores'])), torch.as_tensor(g['core.locations']),
           torch.as_tensor(g['core.refinement']), torch.as_tensor(g['core.fourier']))
    for key, a, b, e in zip(('scores', 'locations', 'refinement', 'fourier'), maps[False], maps[True], ref):
        ea, eb = rel(a, e), rel(b, e)
        print(f'{name} fp8 {key}: relL2 vs fp32 reference: stated conv {ea:.3e}, sub-pixel triples {eb:.3e}; between the two {rel(b, a):.3e}')
        assert torch.isfinite(b).all()
        if all(v % 32 == 0 for v in x.shape[-2:]):
            assert eb < 1.15 * ea + 1e-2, (key, ea, eb)
        else:
            assert torch.equal(a, b), key


def test_fp8_scales_of_another_plan_are_never_applied(dev):
    """the fp8 activation scales are indexed by the tensor ids of the plan they were calibrated with, and toggling
    ``model.subpixel`` changes that plan (one more tensor per decoder level).  A toggle after calibrate_fp8() must drop the scales
    (warning + recalibration on the forwarded batch, or an error without one) -- never shift every scale by a tensor."""
    m, g = build('CpnResNeXt101UNet', dev)
    x = torch.as_tensor(g['x']).to(dev)
    m.precision = 'fp8'
    m.calibrate_fp8(x)
    m.subpixel = False
    with pytest.raises(RuntimeError, match='calibrate_fp8'), pytest.warns(RuntimeWarning, match='subpixel'):
        m.engine(dev)
    m.calibrate_fp8(x)
    m.subpixel = True
    with pytest.warns(RuntimeWarning):
        got = [t.cpu() for t in m.core_forward(x)]  # recalibrates on x itself
    m2, _ = build('CpnResNeXt101UNet', dev)
    m2.precision = 'fp8'
    m2.calibrate_fp8(x)
    for a, b in zip(got, m2.core_forward(x)):
        assert torch.equal(a, b.cpu())


@pytest.mark.parametrize('size', [(64, 96), (100, 140), (48, 68), (130, 260), (24, 520)])
def test_bilinear_phase_frame_pixels_identical_any_size(dev, size, monkeypatch):
    """Frame launches (CPN_SUBPIXEL_BL_FRAME) cut the two sides of a row into ONE wrap tile (output columns W - 16 .. W - 1 and
    0 .. 15: two halo segments, csrc/conv_igemm.hip frame_tiles): on the frame the op is the head conv itself -- same kernel,
    same operands, same K order -- so its pixels are bit-identical to the plain head's at any width (multiples of 32 or not, one
    or several inner tile columns), and the phase convs fill the interior to bf16 accuracy."""
    import celldetection_amd as cda
    from celldetection_amd.synth import synth_state_dict
    x = torch.rand(2, 3, *size, generator=torch.Generator().manual_seed(size[1])).to(dev)
    maps = {}
    for mode in ('0', '2'):
        monkeypatch.setenv('CPN_BLPHASE', mode)
        m = cda.models.CpnResNet18FPN(3, backbone_kwargs={'fpn_channels': 32, 'backbone_kwargs': {'base_channel': 8}})
        m.load_state_dict(synth_state_dict(m.state_dict(), seed=5))
        m = m.to(dev)
        maps[mode] = m.core_forward(x)[2].float().cpu()
        prof = m.engine(dev).profile(x, m.core.order, True)
        ran = [(p['k'], p['gflop'] > 0) for p in prof if 'refinement_head.block.0' in (p['name'] or '')]
        assert ran == ([(7, True), (5, False), (7, False)] if mode == '0' else [(7, False), (5, True), (7, True)]), (mode, ran)
    old, new = maps['0'], maps['2']
    f = 4
    assert torch.equal(new[..., :f, :], old[..., :f, :]) and torch.equal(new[..., -f:, :], old[..., -f:, :])
    assert torch.equal(new[..., :, :f], old[..., :, :f]) and torch.equal(new[..., :, -f:], old[..., :, -f:])
    inner = (new[..., f:-f, f:-f] - old[..., f:-f, f:-f])
    rel = (inner.norm() / (old[..., f:-f, f:-f].norm() + 1e-12)).item()
    print(size, 'interior relL2 between phases and the conv over the resized map', f'{rel:.3e}')
    assert 0 < rel < 5e-2


def test_error_behaviour_of_options_recorded_from_the_reference(dev):
    """options of built rows whose behaviour in the reference IS an error (tests/golden/reference_behaviours.json,
    recorded by make_golden.py gen_behaviours from the imported reference) behave the same here -- no NotImplementedError in
    their place: ``functional=True`` (IndexError as soon as there is a proposal, empty outputs with a [0, 2 * order, 2] Fourier
    tensor otherwise) and a non-interpolating ``refinement_interpolation`` (torch's ValueError where a resize is needed, nothing
    where none is)."""
    import json
    import celldetection_amd as cda
    with open(os.path.join(G, 'reference_behaviours.json')) as f:
        rec = json.load(f)
    model, g = build('CpnU22', dev)
    x = torch.as_tensor(g['x']).to(dev)
    model.functional = True
    e = rec['functional_true_with_proposals']
    assert e['type'] == 'IndexError'
    with pytest.raises(IndexError, match=e['message']):
        model(x)
    model.score_thresh = 1.1  # no proposals
    y = model(x)
    assert list(y['fourier'][0].shape) == rec['functional_true_no_proposals']['fourier_shape']
    assert list(y['contours'][0].shape) == rec['functional_true_no_proposals']['contours_shape']
    fpn = dict(backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}})
    for mode in ('nearest', 'area', 'nearest-exact'):
        e = rec[f'refinement_interpolation_{mode}_fpn_forward']
        m = cda.models.CpnResNet18FPN(3, refinement_interpolation=mode, **fpn).to(dev)
        with pytest.raises(ValueError, match='align_corners option can only be set'):
            m(x)
        assert e['type'] == 'ValueError' and e['message'].startswith('align_corners option can only be set')
    assert rec['refinement_interpolation_nearest_u22_forward'] is None
    mu = cda.models.CpnU22(3, refinement_interpolation='nearest', backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}).to(dev)
    mu(x)  # level 0 has the input size: no resize, the mode is never used


def test_split_batches_replay_their_graphs_and_equal_whole_batches(dev, monkeypatch):
    """A batch the engine must split (2^31-byte tensors; forced here through ``max_batch``) runs as balanced sub-batches that are
    graph runs of their own (round 6: they used to be eager launches): same head maps as the unsplit batch, bit for bit, across
    repeated calls (the parts live in hipGraph slots and are copied out before a slot returns) -- two parts (<= GRAPH_SLOTS) and
    five parts (> GRAPH_SLOTS), ragged last part included."""
    from celldetection_amd import cpn
    model, g = build('CpnResNet18FPN', dev)
    model.sparse_heads = False
    x = torch.as_tensor(g['x']).to(dev)
    x = torch.cat([x, x.flip(-1), x.flip(-2)])[:5].contiguous()   # 5 images
    whole = [t.clone() for t in model.core_forward(x)]
    for cap in (3, 1):
        monkeypatch.setattr(cpn._Engine, 'max_batch', lambda self, n, h, w, cap=cap: -(-n // -(-n // min(n, cap))))
        for rep in range(4):  # (a shape is captured the second time in a row it is seen; later calls replay)
            parts = model.core_forward(x)
            for a, b in zip(parts, whole):
                assert torch.equal(a, b), (cap, rep)
        y = model(x)
        assert len(y['scores']) == 5
    monkeypatch.undo()
    y0 = model(x)
    for k in KEYS:
        for a, b in zip(y[k], y0[k]):
            assert torch.equal(a, b), k
