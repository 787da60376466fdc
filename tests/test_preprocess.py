"""Percentile normalisation of the slide (cpn_inference.py:196-222): HIP path vs numpy."""
import numpy as np
import pytest
import torch

import preprocess_oracle as po


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
