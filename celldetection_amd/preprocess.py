"""Slide preprocessing of the inference script on the MI355X: ``preprocess`` / ``normalize_percentile``
(celldetection_scripts/cpn_inference.py:196-222, celldetection/data/misc.py:156-161).

The slide stays on the device: order statistics come from a value histogram (8/16-bit images; exact, no sort of 10^9
values) or from ``torch.kthvalue`` (float images), the clip / rescale / round-to-uint8 pass is one HBM-bound kernel.
``np.percentile``'s linear interpolation and ``skimage.img_as_ubyte``'s rounding (rint of value * 255 in float64) are
restated; skimage is not part of the reference repository (third-party, unpinned).
"""
import warnings

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

__all__ = ['normalize_percentile', 'preprocess']

_DT = {torch.float32: 0, torch.uint8: 1, torch.uint16: 2, torch.int16: 2}


def _lerp(a: float, b: float, t: float) -> float:
    """numpy's ``_lerp`` (lib/function_base.py): a + (b - a) * t, evaluated from b for t >= 0.5."""
    d = b - a
    return b - d * (1 - t) if t >= 0.5 else a + d * t


def _order_statistics(x: torch.Tensor, ranks):
    """Values at the 0-based sorted positions ``ranks`` (ascending ints) of the flattened tensor, as Python floats."""
    n = x.numel()
    flat = x.reshape(-1)
    if x.dtype in (torch.uint8, torch.uint16, torch.int16):
        if x.dtype == torch.int16:
            if int(flat.min().item()) < 0:
                raise NotImplementedError('negative int16 images: convert to float32 first')
        bins = 256 if x.dtype == torch.uint8 else 65536
        hist = torch.zeros(bins, dtype=torch.int32, device=x.device)
        check(_lib.load().cpn_histogram(ptr(flat), _DT[x.dtype], n, ptr(hist), stream_ptr()), 'histogram')
        cum = torch.cumsum(hist.to(torch.int64) & 0xFFFFFFFF, 0).cpu().numpy()  # bins are uint32 counters
        return [float(np.searchsorted(cum, r, side='right')) for r in ranks]
    # float images: the (up to four) ranks sit in the two tails of the distribution.  ONE top-k per tail and ONE read-back
    # instead of a kthvalue (= a full selection pass + a host sync) per rank; a full sort only when a rank is far from both
    # ends (percentiles near 50: not what the slide normalisation asks for)
    flat = flat.float()
    lo_r = [r for r in ranks if r < n - 1 - r]
    hi_r = [r for r in ranks if r >= n - 1 - r]
    k_lo, k_hi = (max(lo_r) + 1 if lo_r else 0), (n - min(hi_r) if hi_r else 0)
    if max(k_lo, k_hi) > max(4096, n // 64):
        srt = torch.sort(flat).values
        return srt[torch.tensor(ranks, device=x.device)].cpu().tolist()
    parts = []
    if lo_r:
        small = torch.topk(flat, k_lo, largest=False, sorted=True).values  # ascending: position r
        parts.append(small[torch.tensor(lo_r, device=x.device)])
    if hi_r:
        large = torch.topk(flat, k_hi, largest=True, sorted=True).values   # descending: position n - 1 - r
        parts.append(large[torch.tensor([n - 1 - r for r in hi_r], device=x.device)])
    vals = torch.cat(parts).cpu().tolist()
    return [vals[(lo_r + hi_r).index(r)] for r in ranks]


def normalize_percentile(image: torch.Tensor, percentile=99.9, to_uint8=True):
    """``cd.data.normalize_percentile`` for a GPU tensor of any layout (the statistics are global): low / high =
    np.percentile(image, (100 - p, p)) (linear interpolation), ``(clip(image, low, high) - low) / (high - low)``, then
    ``img_as_ubyte``.  uint8 / uint16 / float32 inputs; returns uint8 (or float32 in [0, 1] with ``to_uint8=False``)."""
    if not image.is_cuda:
        raise RuntimeError('celldetection_amd.normalize_percentile runs on the MI355X only (got a CPU tensor).')
    if image.dtype not in _DT:
        image = image.float()
    if not isinstance(percentile, (list, tuple)):
        percentile = (100 - percentile, percentile)
    x = image.contiguous()
    n = x.numel()
    pos = [float(q) / 100. * (n - 1) for q in percentile]  # np.percentile, method='linear'
    lo_i = [min(int(np.floor(p)), n - 1) for p in pos]
    ranks = sorted({r for i in lo_i for r in (i, min(i + 1, n - 1))})
    vals = dict(zip(ranks, _order_statistics(x, ranks)))
    low, high = (_lerp(vals[i], vals[min(i + 1, n - 1)], p - i) for i, p in zip(lo_i, pos))
    if not high > low:
        raise ValueError(f'normalize_percentile: degenerate range (low {low}, high {high})')
    if not to_uint8:
        return ((x.double().clamp(low, high) - low) / (high - low)).float()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(_lib.load().cpn_rescale_to_uint8(ptr(x), _DT[x.dtype], n, low, high, ptr(out), stream_ptr()), 'rescale_to_uint8')
    return out


def preprocess(img: torch.Tensor, gamma=1., contrast=1., brightness=0., percentile=None, grayscale=False):
    """``preprocess`` of the inference script (cpn_inference.py:196-222) for a slide on the GPU: optional percentile
    normalisation, implicit percentile normalisation of non-uint8 inputs.  The cv2 / albumentations steps (grayscale
    conversion, gamma, contrast) are not part of the HIP path."""
    if grayscale or gamma != 1. or contrast != 1. or brightness != 0.:
        raise NotImplementedError('preprocess on the HIP path: grayscale / gamma / contrast / brightness are not supported')
    if percentile is not None:
        img = normalize_percentile(img, percentile)
    if img.element_size() > 1:
        warnings.warn('Performing implicit percentile normalization, since input is not uint8.')
        img = normalize_percentile(img)
    return img
