"""Sub-pixel decomposition of a 3x3 convolution over a x2 nearest-upsampled map (host-side math, not wired into the plan yet).

The UNet decoder's first conv of every level reads ``cat(lateral, F.interpolate(top_down, mode='nearest'))``
(celldetection/models/unet.py:207-230).  For the exact x2 case every output pixel ``(2i + py, 2j + px)`` sees the upsampled map
through only 2 x 2 distinct low-resolution pixels: the three taps of a row collapse onto two source rows

    py = 0:  rows 2i-1, 2i, 2i+1  ->  low-res rows i-1, i, i      weights [w0, w1 + w2]      (offsets -1, 0)
    py = 1:  rows 2i, 2i+1, 2i+2  ->  low-res rows i, i, i+1      weights [w0 + w1, w2]      (offsets 0, +1)

(and the same along x), so the upsampled part of the conv is FOUR 2 x 2 convolutions on the low-resolution map, one per output
phase: 4/9 of the multiply-accumulates.  Zero padding is preserved: the rows / columns outside the upsampled map correspond to
rows / columns outside the low-resolution map.  DESIGN.md section 7 (planned for the conv kernel: per-axis padding, 2 x 2 taps,
pixel-shuffled residual read); ``tests/test_host_logic.py::test_subpixel_decomposition_is_exact`` pins the algebra.
"""
import torch

__all__ = ['collapse_upsampled_taps', 'phase_padding', 'upsampled_conv_by_phases']


def collapse_upsampled_taps(weight: torch.Tensor) -> torch.Tensor:
    """weight [cout, cin, 3, 3] of a conv applied to a x2 nearest-upsampled input -> [2, 2, cout, cin, 2, 2]: the 2 x 2 kernel of
    output phase (py, px) on the low-resolution input (tap offsets ``phase_padding``)."""
    if weight.ndim != 4 or tuple(weight.shape[2:]) != (3, 3):
        raise ValueError('expected a [cout, cin, 3, 3] weight')
    w = weight
    rows = (torch.stack((w[:, :, 0], w[:, :, 1] + w[:, :, 2]), 2),   # py = 0: [w0, w1 + w2]
            torch.stack((w[:, :, 0] + w[:, :, 1], w[:, :, 2]), 2))   # py = 1: [w0 + w1, w2]
    out = []
    for r in rows:  # r: [cout, cin, 2, 3] -> collapse the columns the same way
        out.append(torch.stack((torch.stack((r[..., 0], r[..., 1] + r[..., 2]), -1),
                                torch.stack((r[..., 0] + r[..., 1], r[..., 2]), -1))))
    return torch.stack(out)  # [py, px, cout, cin, 2, 2]


def phase_padding(p: int) -> int:
    """Leading zero padding of the 2-tap kernel of phase ``p`` along one axis (taps at offsets -pad, -pad + 1)."""
    return 1 - p


def upsampled_conv_by_phases(x_low: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``F.conv2d(F.interpolate(x_low, scale_factor=2, mode='nearest'), weight, padding=1)`` computed as four 2 x 2
    convolutions on ``x_low`` [N, cin, h, w] and a pixel shuffle -> [N, cout, 2h, 2w] (reference implementation of the
    decomposition in plain torch; the HIP path will do the same inside the conv kernel)."""
    import torch.nn.functional as F
    n, _, h, w = x_low.shape
    wc = collapse_upsampled_taps(weight)
    out = x_low.new_zeros((n, weight.shape[0], 2 * h, 2 * w))
    for py in (0, 1):
        for px in (0, 1):
            pt, pl = phase_padding(py), phase_padding(px)
            xp = F.pad(x_low, (pl, 1 - pl, pt, 1 - pt))  # (left, right, top, bottom): two taps starting at offset -pad
            out[:, :, py::2, px::2] = F.conv2d(xp, wc[py, px])
    return out
