"""Slide preprocessing of the inference script on the MI355X: ``preprocess`` / ``normalize_percentile``
(celldetection_scripts/cpn_inference.py:196-222, celldetection/data/misc.py:156-161).

The slide stays on the device: order statistics come from a value histogram (8/16-bit images; exact, no sort of 10^9
values) or from ``torch.kthvalue`` (float images), the clip / rescale / round-to-uint8 pass is one HBM-bound kernel.
``np.percentile``'s linear interpolation and ``skimage.img_as_ubyte``'s rounding (rint of value * 255 in float64) are
restated; skimage is not part of the reference repository (third-party, unpinned).  The script's optional grayscale / gamma /
contrast steps (cv2, albumentations: third party, unpinned) are 256-entry table look-ups and one fixed-point luma pass.
"""
import warnings

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

__all__ = ['normalize_percentile', 'preprocess', 'to_grayscale']

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


def _lut(img: torch.Tensor, table) -> torch.Tensor:
    """``cv2.LUT`` for a uint8 tensor: one gather through a 256-entry table."""
    t = torch.as_tensor(np.asarray(table), dtype=torch.uint8, device=img.device)
    return t[img.long()]


def to_grayscale(img: torch.Tensor) -> torch.Tensor:
    """``cv2.cvtColor(img, COLOR_RGB2GRAY | COLOR_RGBA2GRAY)`` for a uint8 [3 | 4, H, W] tensor -> [H, W]: OpenCV's 8-bit path
    is fixed point, ``(R * 4899 + G * 9617 + B * 1868 + 2^13) >> 14`` (0.299 / 0.587 / 0.114 in 14 fractional bits; restated --
    cv2 is not in the image: third-party, unpinned)."""
    r, g, b = (img[i].to(torch.int32) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).to(torch.uint8)


def preprocess(img: torch.Tensor, gamma=1., contrast=1., brightness=0., percentile=None, grayscale=False):
    """``preprocess`` of the inference script (cpn_inference.py:196-222) for a slide on the GPU, Tensor[C, H, W] or [H, W] (the
    layout ``tiled_inference`` takes; the script's arrays are channels-last): optional percentile normalisation, implicit
    percentile normalisation of non-uint8 inputs, ``grayscale`` (1 / 3 / 4 channels -> luma, replicated to three channels like the
    script's GRAY2RGB; the script's 2-channel branch hands a float64 array to cv2 and cannot run), ``gamma`` (albumentations
    ``gamma_transform``: 256-entry table ``(i / 255) ** gamma * 255`` truncated to uint8) and ``contrast`` / ``brightness``
    (albumentations ``brightness_contrast_adjust``: table ``clip(i * alpha + alpha * beta * mean(img), 0, 255)`` truncated to uint8 --
    applied only when ``contrast != 1``, like the script).  cv2 / albumentations are third party and absent here: their documented
    arithmetic is restated (unpinned; ``oracle/preprocess_oracle.py`` holds the numpy statement the GPU path is tested against)."""
    if percentile is not None:
        img = normalize_percentile(img, percentile)
    if img.element_size() > 1:
        warnings.warn('Performing implicit percentile normalization, since input is not uint8.')
        img = normalize_percentile(img)
    if not img.is_cuda:
        raise RuntimeError('celldetection_amd.preprocess runs on the MI355X only (got a CPU tensor).')
    if grayscale and img.ndim == 3:
        channels = img.shape[0]
        if channels == 1:
            img = img[0]
        elif channels in (3, 4):
            img = to_grayscale(img)
        else:
            raise ValueError(f'Unsupported number of channels: {channels}')
    if img.ndim == 2:
        img = img[None].expand(3, -1, -1).contiguous()  # cv2.COLOR_GRAY2RGB
    if gamma != 1.:
        img = _lut(img, (np.arange(0, 256. / 255, 1. / 255) ** gamma * 255).astype(np.uint8))
    if contrast != 1.:
        lut = np.arange(0, 256).astype('float32')
        lut *= contrast
        if brightness != 0:
            lut += (contrast * brightness) * (int(img.sum(dtype=torch.int64).item()) / img.numel())  # np.mean: exact integer sum
        img = _lut(img, np.clip(lut, 0, 255).astype(np.uint8))
    return img
