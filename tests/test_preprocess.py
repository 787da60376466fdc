"""Percentile normalisation of the slide (cpn_inference.py:196-222): HIP path vs numpy."""
import json
import os

import numpy as np
import pytest
import torch

import preprocess_oracle as po

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess.npz')


def _golden():
    g = np.load(G)
    return g, json.loads(str(g['cases']))


def test_oracle_matches_the_imported_reference():
    """SURVEY row f2 pin: ``preprocess.npz`` holds outputs of the reference's OWN ``cd.data.normalize_percentile``
    (data/misc.py:156-161) and ``preprocess`` (celldetection_scripts/cpn_inference.py:196-222), imported by
    tests/golden/make_golden.py gen_preprocess; the oracle must reproduce every case -- bit for bit, float64 included (same
    numpy expression) -- and reject what the script rejects."""
    g, cases = _golden()
    assert len(cases) > 150
    for c in cases:
        img = g[f'img.{c["img"]}']
        if c['fn'] == 'normalize_percentile':
            pct = tuple(c['percentile']) if isinstance(c['percentile'], list) else c['percentile']
            got = po.normalize_percentile(img.copy(), pct, to_uint8=c['to_uint8'])
            exp = g[c['key']]
            assert got.dtype == exp.dtype and got.shape == exp.shape, c
            np.testing.assert_array_equal(got, exp, err_msg=str(c))
        else:
            kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['kwargs'].items()}
            if c['error']:
                with pytest.raises(Exception):
                    po.preprocess(img.copy(), **kw)
                continue
            got, exp = po.preprocess(img.copy(), **kw), g[c['key']]
            assert got.dtype == exp.dtype and got.shape == exp.shape, c
            np.testing.assert_array_equal(got, exp, err_msg=str(c))


@pytest.mark.gpu
def test_hip_preprocess_matches_the_imported_reference():
    """The GPU path against the same fixture: uint8 results identical, ``to_uint8=False`` within float32 rounding of the
    reference's float64 values; [C, H, W] here = channels-last there."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import warnings
    from celldetection_amd.preprocess import normalize_percentile, preprocess
    g, cases = _golden()

    def dev_tensor(img):
        t = torch.as_tensor(img.astype(np.int32)).to(torch.uint16) if img.dtype == np.uint16 else torch.as_tensor(img)
        return t.cuda()

    for c in cases:
        img = g[f'img.{c["img"]}']
        if c['fn'] == 'normalize_percentile':
            pct = tuple(c['percentile']) if isinstance(c['percentile'], list) else c['percentile']
            got = normalize_percentile(dev_tensor(img), pct, to_uint8=c['to_uint8']).cpu().numpy()
            exp = g[c['key']]
            assert got.shape == exp.shape, c
            if c['to_uint8']:
                assert got.dtype == np.uint8
                np.testing.assert_array_equal(got, exp, err_msg=str(c))
            else:
                np.testing.assert_allclose(got, exp, rtol=0, atol=2e-7, err_msg=str(c))
        else:
            kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['kwargs'].items()}
            t = dev_tensor(img)
            t = t.permute(2, 0, 1).contiguous() if t.ndim == 3 else t
            if c['error']:
                with pytest.raises(Exception):
                    preprocess(t, **kw)
                continue
            with warnings.catch_warnings(record=True) as wl:
                warnings.simplefilter('always')
                got = preprocess(t, **kw)
            assert any('implicit percentile' in str(w.message) for w in wl) == c['warned'], c
            exp = g[c['key']].transpose(2, 0, 1)
            assert got.dtype == torch.uint8 and tuple(got.shape) == exp.shape, (c, tuple(got.shape), exp.shape)
            np.testing.assert_array_equal(got.cpu().numpy(), exp, err_msg=str(c))


def test_oracle_basic():
    x = np.arange(1000, dtype=np.uint16).reshape(10, 100)
    y = po.normalize_percentile(x, 99.)
    assert y.dtype == np.uint8 and y.min() == 0 and y.max() == 255
    assert np.all(np.diff(y.reshape(-1).astype(int)) >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,pct', [('uint8', 99.9), ('uint16', 99.9), ('uint16', (1., 97.5)), ('float32', 99.)])
def test_normalize_percentile_matches_numpy(dtype, pct):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from celldetection_amd.preprocess import normalize_percentile, preprocess
    rng = np.random.default_rng(3)
    if dtype == 'uint8':
        x = rng.integers(0, 256, (3, 301, 257)).astype(np.uint8)
    elif dtype == 'uint16':
        x = (rng.gamma(2., 900., (3, 301, 257))).clip(0, 65535).astype(np.uint16)
    else:
        x = rng.standard_normal((3, 301, 257)).astype(np.float32)
    t = torch.as_tensor(x.astype(np.int32) if dtype == 'uint16' else x)
    t = t.to(torch.uint16) if dtype == 'uint16' else t
    got = normalize_percentile(t.cuda(), pct)
    exp = po.normalize_percentile(x, pct)
    assert got.dtype == torch.uint8 and tuple(got.shape) == exp.shape
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
    if dtype != 'uint8':  # the script's implicit normalisation of non-uint8 inputs
        with pytest.warns(UserWarning):
            np.testing.assert_array_equal(preprocess(t.cuda()).cpu().numpy(), po.normalize_percentile(x))


def test_oracle_preprocess_steps():
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (20, 30, 3)).astype(np.uint8)
    g = po.preprocess(x, grayscale=True)
    assert g.shape == (20, 30, 3) and (g[..., 0] == g[..., 1]).all() and (g[..., 0] == g[..., 2]).all()
    assert abs(float(g[..., 0].mean()) - float((x * [.299, .587, .114]).sum(-1).mean())) < .6
    white = np.full((4, 4, 3), 255, np.uint8)
    assert (po.preprocess(white, grayscale=True) == 255).all()        # the fixed-point coefficients sum to 2^14
    assert (po.preprocess(x, gamma=1.) == x).all() and po.preprocess(x, gamma=2.).mean() < x.mean()
    assert (po.preprocess(x, contrast=1., brightness=.5) == x).all()  # the script ignores brightness when contrast == 1
    assert po.preprocess(x, contrast=1.5).max() == 255


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(grayscale=True), dict(gamma=.7), dict(gamma=2.2, grayscale=True), dict(contrast=1.3),
                                dict(contrast=.8, brightness=.2), dict(contrast=1., brightness=.5),
                                dict(percentile=99., gamma=1.5, contrast=1.2, brightness=-.1, grayscale=True)])
@pytest.mark.parametrize('channels', [1, 3, 4, 0])
def test_preprocess_steps_match_restated_script(kw, channels):
    """grayscale / gamma / contrast / brightness of the script's ``preprocess`` (cpn_inference.py:196-222) on the GPU against the
    numpy statement of the same third-party arithmetic (oracle/preprocess_oracle.py); [C, H, W] here = channels-last there."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from celldetection_amd.preprocess import preprocess
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, (61, 83, channels) if channels else (61, 83)).astype(np.uint8)
    t = torch.as_tensor(x).cuda()
    got = preprocess(t.permute(2, 0, 1).contiguous() if channels else t, **kw)
    exp = po.preprocess(x, **kw)
    exp = exp.transpose(2, 0, 1) if exp.ndim == 3 else exp
    assert got.dtype == torch.uint8 and tuple(got.shape) == exp.shape, (tuple(got.shape), exp.shape)
    np.testing.assert_array_equal(got.cpu().numpy(), exp)
