"""Pins the CPU oracle (oracle/cpn_oracle.py) against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import cpn_oracle as orc
from celldetection_amd.synth import synth_state_dict
from model_specs import HEAD_SPECS, MODEL_SPECS, SIZE_SPECS, VARIANT_SPECS, ref_template_state_dict

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def ops():
    return np.load(os.path.join(G, 'ops.npz'))


@pytest.mark.parametrize('tag,samples', [('a', 32), ('b', 128), ('c', 7), ('d', 64)])
def test_fouriers2contours_bit_exact(ops, tag, samples):
    out = orc.fouriers2contours(ops[f'f2c_{tag}_fourier'], ops[f'f2c_{tag}_loc'], samples)
    np.testing.assert_array_equal(out, ops[f'f2c_{tag}_out'])


def test_fouriers2contours_custom_sampling_bit_exact(ops):
    out = orc.fouriers2contours(ops['f2c_s_fourier'], ops['f2c_s_loc'], 5, sampling=ops['f2c_s_sampling'])
    np.testing.assert_array_equal(out, ops['f2c_s_out'])


def test_local_refinement_bit_exact(ops):
    for iters in (1, 4):
        res, all_res = orc.local_refinement(ops['refine_in'], ops['refine_map'], ops['refine_b'], iters, (24, 40))
        np.testing.assert_array_equal(res, ops[f'refine_out_{iters}'])
    np.testing.assert_array_equal(np.stack(all_res), ops['refine_all'])


@pytest.mark.parametrize('thr', [.2, .5, 0.])
def test_nms_index_sets(ops, thr):
    for fn in (orc.nms, orc.nms_numpy):
        keep = fn(ops['nms_boxes'], ops['nms_scores'], thr)
        np.testing.assert_array_equal(keep, ops[f'nms_keep_{thr}'])


def test_nmsi_chunked(ops):
    keep = orc.batched_box_nmsi([ops['nms_boxes']], [ops['nms_scores']], .2, batch_size=128)[0]
    np.testing.assert_array_equal(keep, ops['nmsi_chunked_keep'])
    keep = orc.batched_box_nmsi([ops['nms_boxes'][:50]], [ops['nms_scores'][:50]], .2)[0]
    np.testing.assert_array_equal(keep, ops['nmsi_plain_keep'])


def test_border_and_stitch_rule(ops):
    flags = [(True, True, True, True), (False, True, False, True), (True, False, True, False)]
    for i, (top, right, bottom, left) in enumerate(flags):
        keep = orc.remove_border_contours(ops['border_in'], (48, 64), 4, top=top, right=right, bottom=bottom,
                                          left=left, offsets=ops['border_offsets'])
        np.testing.assert_array_equal(keep, ops[f'border_keep_{i}'])
    keep = orc.filter_contours_by_stitching_rule(ops['border_in'], (48, 64), ops['stitch_overlaps'],
                                                 offsets=ops['border_offsets'])
    np.testing.assert_array_equal(keep, ops['stitch_keep'])


def test_tiling_tables():
    t = np.load(os.path.join(G, 'tiling.npz'))
    for tag in 'abcdef':
        slices, overlaps, shape = orc.get_tiling_slices(tuple(t[f'{tag}_size']), tuple(t[f'{tag}_crop']),
                                                        tuple(t[f'{tag}_stride']))
        np.testing.assert_array_equal(np.array(slices), t[f'{tag}_slices'])
        np.testing.assert_array_equal(np.array(overlaps), t[f'{tag}_overlaps'])
        np.testing.assert_array_equal(np.array(shape), t[f'{tag}_shape'])


def _load_model_fixture(name):
    g = np.load(os.path.join(G, f'model_{name}.npz'))
    overrides = {k[len('override.'):]: torch.as_tensor(g[k]) for k in g.files if k.startswith('override.')}
    sd = synth_state_dict(ref_template_state_dict(name), seed=int(g['seed']), overrides=overrides)
    return g, sd


def _check_outputs(prefix, y, g, n):
    keys = ('contours', 'boxes', 'scores', 'classes', 'locations', 'fourier', 'contour_proposals')
    if prefix == 'nms':  # a fixture that keeps a handful of detections pins next to nothing of the NMS keep set
        assert all(len(g[f'nms.scores.{i}']) >= 20 for i in range(n)), 'thin golden fixture: regenerate with >= 30 kept detections'
    if f'{prefix}.box_uncertainties.0' in g.files:
        keys += ('box_uncertainties',)
    else:
        assert y['box_uncertainties'] is None
    for k in keys:
        for i in range(n):
            exp = g[f'{prefix}.{k}.{i}']
            got = y[k][i]
            assert got.shape == exp.shape, (prefix, k, i, got.shape, exp.shape)
            if k == 'classes':
                np.testing.assert_array_equal(got, exp)
            else:
                np.testing.assert_allclose(got, exp, rtol=0, atol=1e-4, err_msg=f'{prefix}.{k}.{i}')


@pytest.mark.parametrize('name', list(MODEL_SPECS))
def test_model_core_and_forward(name):
    g, sd = _load_model_fixture(name)
    kw = MODEL_SPECS[name]['cpn_kwargs']
    x = torch.as_tensor(g['x'])
    # the fixtures were generated with 4 CPU threads; with the same thread count the oracle's conv graph is
    # bit-identical to the reference's here, other counts change oneDNN's summation order (<= 1e-3 abs on the
    # tanh*3 refinement map of the deep synthetic ResNets) -- hence the loose absolute tolerance.
    torch.set_num_threads(4)
    s, l, r, f = orc.core_forward(sd, x)
    for got, key in ((s, 'scores'), (l, 'locations'), (r, 'refinement'), (f, 'fourier')):
        exp = g[f'core.{key}']
        np.testing.assert_allclose(got.numpy(), exp, rtol=1e-4, atol=1e-3)
    n = x.shape[0]
    size = tuple(x.shape[-2:])
    # post-processing pinned on the REFERENCE's head maps => index sets must be bit-exact
    maps = (g['core.scores'], g['core.locations'], g['core.refinement'], g['core.fourier'])
    _check_outputs('nms', orc.cpn_postprocess(*maps, input_size=size, **kw), g, n)
    _check_outputs('nonms', orc.cpn_postprocess(*maps, input_size=size, nms=False, **kw), g, n)
    _check_outputs('offs', orc.cpn_postprocess(*maps, input_size=size, offsets=g['offsets'], **kw), g, n)
    _check_outputs('bounds', orc.cpn_postprocess(*maps, input_size=size, scores_upper_bound=g['scores_upper_bound'],
                                                 scores_lower_bound=g['scores_lower_bound'], **kw), g, n)
    if name == 'CpnU22':
        # no refinement: contours IS contour_proposals in the reference -> the offsets are added to it twice
        _check_outputs('noref_offs', orc.cpn_postprocess(*maps, input_size=size, offsets=g['offsets'],
                                                         **dict(kw, refinement_iterations=0)), g, n)
        assert np.abs(g['noref_offs.contours.0'] - g['noref_offs.contour_proposals.0']).max() == 0
        kw2 = dict(kw, samples=17, refinement_iterations=2, score_thresh=.7, nms_thresh=.5)
        _check_outputs('attr', orc.cpn_postprocess(*maps, input_size=size, **kw2), g, n)
        _check_outputs('attr_order3', orc.cpn_postprocess(*maps, input_size=size, order=3, **kw2), g, n)


@pytest.mark.parametrize('name', list(VARIANT_SPECS))
def test_variant_core_and_forward(name):
    """CPN.forward variants: bucketed refinement, uncertainty head (+ certainty filter, uncertainty-weighted NMS),
    multi-class softmax scores, narrower head channels / other head kernel sizes."""
    g, sd = _load_model_fixture(name)
    kw = VARIANT_SPECS[name]['cpn_kwargs']
    x = torch.as_tensor(g['x'])
    torch.set_num_threads(4)
    s, l, r, f, u = orc.core_forward(sd, x, with_uncertainty=True)
    for got, key in ((s, 'scores'), (l, 'locations'), (r, 'refinement'), (f, 'fourier'), (u, 'uncertainty')):
        if got is None:
            assert f'core.{key}' not in g.files
            continue
        np.testing.assert_allclose(got.numpy(), g[f'core.{key}'], rtol=1e-4, atol=1e-3)
    n, size = x.shape[0], tuple(x.shape[-2:])
    maps = (g['core.scores'], g['core.locations'], g['core.refinement'], g['core.fourier'])
    unc = g['core.uncertainty'] if 'core.uncertainty' in g.files else None
    _check_outputs('nms', orc.cpn_postprocess(*maps, input_size=size, uncertainty=unc, **kw), g, n)
    _check_outputs('nonms', orc.cpn_postprocess(*maps, input_size=size, nms=False, uncertainty=unc, **kw), g, n)
    _check_outputs('offs', orc.cpn_postprocess(*maps, input_size=size, offsets=g['offsets'], uncertainty=unc, **kw),
                   g, n)
    _check_outputs('bounds', orc.cpn_postprocess(*maps, input_size=size, uncertainty=unc,
                                                 scores_upper_bound=g['scores_upper_bound'],
                                                 scores_lower_bound=g['scores_lower_bound'], **kw), g, n)


def test_stitch():
    g = np.load(os.path.join(G, 'stitch.npz'))
    overrides = {k[len('override.'):]: torch.as_tensor(g[k]) for k in g.files if k.startswith('override.')}
    sd = synth_state_dict(ref_template_state_dict('CpnU22', 'stitch.npz'), seed=0, overrides=overrides)
    res, pre = orc.tiled_inference(sd, torch.as_tensor(g['img']), tuple(g['crop']), tuple(g['stride']),
                                   border_removal=int(g['border']))
    assert pre == int(g['pre_nms_count'])
    for k, v in res.items():
        exp = g[f'final.{k}']
        assert v.shape == exp.shape, k
        np.testing.assert_allclose(v, exp, rtol=0, atol=2e-4, err_msg=k)


@pytest.mark.parametrize('name', ['CpnU22', 'CpnResNet18FPN', 'CpnResNeXt101UNet'])
def test_fp8_simulator_against_reference_maps(name):
    """The CPU restatement of OUR fp8 algorithm (oracle/fp8_sim.py + the packer's dequantised weights) stays within the
    e4m3 error level of the reference's fp32 head maps, and its un-quantised walk reproduces them (i.e. the plan, the
    BN folding and the effective-weight bookkeeping are right)."""
    import celldetection_amd as cda
    import fp8_sim
    from celldetection_amd import _lib, graph
    g, sd = _load_model_fixture(name)
    spec = MODEL_SPECS[name]
    plan = getattr(cda.models, spec['cls'])(**spec['kwargs']).plan_for('fp8')
    x = torch.as_tensor(g['x'])
    torch.set_num_threads(4)
    exp = {_lib.OUT_SCORES: torch.sigmoid(torch.as_tensor(g['core.scores'])),
           _lib.OUT_LOCATIONS: torch.as_tensor(g['core.locations']),
           _lib.OUT_FOURIER: torch.as_tensor(g['core.fourier']), _lib.OUT_REFINEMENT: torch.as_tensor(g['core.refinement'])}
    folded = [dict(zip(('w', 'b'), graph._fold(sd, op))) for op in plan.ops if op['op'] == 'conv']
    flt = fp8_sim.simulate(plan, sd, folded, None, x, _absmax={})
    for k, e in exp.items():  # un-quantised walk (bf16 only in the ReadOut tails): ~1e-2 of the map
        assert ((flt[k] - e).norm() / (e.norm() + 1e-12)).item() < 3e-2, k
    scales = fp8_sim.calibrate(plan, sd, x)
    eff = []
    graph.pack(plan, sd, 'cpu', precision='fp8', act_scales=scales, effective_weights=eff)
    sim = fp8_sim.simulate(plan, sd, eff, scales, x)
    for k, e in exp.items():
        rel = ((sim[k] - e).norm() / (e.norm() + 1e-12)).item()
        assert torch.isfinite(sim[k]).all() and 1e-3 < rel < 0.6, (k, rel)


@pytest.mark.parametrize('nb', [6, 3])
def test_bucketed_refinement_bit_exact(ops, nb):
    """resolve_refinement_buckets tables and bucketed local_refinement (cpn.py:72-82) vs the reference's outputs."""
    tab = orc.refinement_buckets_table(16, nb)
    np.testing.assert_array_equal(np.stack([i for i, _ in tab]), ops[f'bucket_idx_{nb}'])
    np.testing.assert_array_equal(np.stack([w for _, w in tab]), ops[f'bucket_w_{nb}'])
    got, _ = orc.local_refinement(ops['refine_in'], ops[f'refine_bucket_map_{nb}'], ops['refine_b'], 3, (24, 40),
                                  num_buckets=nb)
    np.testing.assert_array_equal(got, ops[f'refine_bucket_out_{nb}'])
    # the product's host-side table builder (pure torch CPU arithmetic, no kernel involved)
    from celldetection_amd.ops import bucket_tables
    idx, wgt = bucket_tables(16, nb, 'cpu')
    np.testing.assert_array_equal(idx.numpy(), ops[f'bucket_idx_{nb}'])
    np.testing.assert_array_equal(wgt.numpy(), ops[f'bucket_w_{nb}'])


def test_stitch_with_cross_tile_duplicates():
    """The stitching rule (border removal per tile -> concat -> ONE global NMS) on per-tile detections that contain
    cross-tile duplicates: the oracle's functions against what the reference's own functions produced (the global NMS
    removes > 10 % here, unlike the model-level stitch fixture)."""
    import stitch_fixture as sf
    g = sf.load()
    _, slices, overlaps, shape = sf.tile_table(g)
    crop, border = tuple(int(i) for i in g['crop']), int(g['border'])
    coll = {}
    for idx, ((h0, h1), (w0, w1)) in enumerate(slices):
        h_i, w_i = np.unravel_index(idx, shape)
        con = g[f'tile{idx}.contours']
        neg = -np.array([w0, h0], np.float32)
        keep = orc.remove_border_contours(con, crop, border, top=h_i > 0, right=w_i < shape[1] - 1,
                                          bottom=h_i < shape[0] - 1, left=w_i > 0, offsets=neg)
        np.testing.assert_array_equal(keep, g[f'tile{idx}.keep_border'])
        keep2 = keep & orc.filter_contours_by_stitching_rule(con, crop, np.array(overlaps[idx]), offsets=neg)
        np.testing.assert_array_equal(keep2, g[f'tile{idx}.keep_border_exbr'])
        for k in sf.KEYS:
            v = g[f'tile{idx}.{k}'][keep]
            coll[k] = np.concatenate((coll[k], v)) if k in coll else v
    assert len(coll['scores']) == int(g['pre_nms_count'])
    keep = orc.nms(coll['boxes'], coll['scores'], float(g['nms_thresh']))
    assert len(keep) <= 0.9 * len(coll['scores'])
    for k in sf.KEYS:
        np.testing.assert_array_equal(coll[k][keep], g[f'final.{k}'], err_msg=k)


@pytest.mark.parametrize('name', list(SIZE_SPECS))
def test_arbitrary_input_sizes(name):
    """Inputs that are not multiples of 32 (odd sizes included): the oracle's conv graph (nearest resize to the
    lateral's size, bilinear resize of the features to the input size, floor-mode pooling) and post-processing (float
    scale factors W/w, H/h) against the reference's outputs at 75x101 / 100x140 / 300x300."""
    g, sd = _load_model_fixture(name)
    kw = SIZE_SPECS[name]['cpn_kwargs']
    x = torch.as_tensor(g['x'])
    torch.set_num_threads(4)
    s, l, r, f = orc.core_forward(sd, x)
    for got, key in ((s, 'scores'), (l, 'locations'), (r, 'refinement'), (f, 'fourier')):
        assert got.shape == g[f'core.{key}'].shape
        np.testing.assert_allclose(got.numpy(), g[f'core.{key}'], rtol=1e-4, atol=1e-3)
    n, size = x.shape[0], tuple(x.shape[-2:])
    maps = (g['core.scores'], g['core.locations'], g['core.refinement'], g['core.fourier'])
    _check_outputs('nms', orc.cpn_postprocess(*maps, input_size=size, **kw), g, n)
    _check_outputs('nonms', orc.cpn_postprocess(*maps, input_size=size, nms=False, **kw), g, n)
    _check_outputs('offs', orc.cpn_postprocess(*maps, input_size=size, offsets=g['offsets'], **kw), g, n)
    _check_outputs('bounds', orc.cpn_postprocess(*maps, input_size=size, scores_upper_bound=g['scores_upper_bound'],
                                                 scores_lower_bound=g['scores_lower_bound'], **kw), g, n)


@pytest.mark.parametrize('name', list(HEAD_SPECS))
def test_head_options(name):
    """CPNCore head options (cpn.py:125-236): strided ReadOut heads (maps at half the resolution, refinement map
    bilinear-resized back to the input size), heads reading another decoder level / an encoder feature, Fuse2d over two
    features -- the oracle's conv graph and post-processing against the reference's outputs."""
    g, sd = _load_model_fixture(name)
    spec = HEAD_SPECS[name]
    x = torch.as_tensor(g['x'])
    torch.set_num_threads(4)
    s, l, r, f = orc.core_forward(sd, x, **spec['core_kwargs'])
    for got, key in ((s, 'scores'), (l, 'locations'), (r, 'refinement'), (f, 'fourier')):
        assert got.shape == g[f'core.{key}'].shape, (key, got.shape, g[f'core.{key}'].shape)
        np.testing.assert_allclose(got.numpy(), g[f'core.{key}'], rtol=1e-4, atol=1e-3)
    n, size = x.shape[0], tuple(x.shape[-2:])
    maps = (g['core.scores'], g['core.locations'], g['core.refinement'], g['core.fourier'])
    _check_outputs('nms', orc.cpn_postprocess(*maps, input_size=size, **spec['cpn_kwargs']), g, n)
    _check_outputs('offs', orc.cpn_postprocess(*maps, input_size=size, offsets=g['offsets'], **spec['cpn_kwargs']), g, n)
