"""Sub-pixel decompositions of convolutions over x2-upsampled maps (host-side math of the packer, `graph.pack`).

The UNet decoder's first conv of every level reads ``cat(lateral, F.interpolate(top_down, mode='nearest'))``
(celldetection/models/unet.py:207-230).  For the exact x2 case every output pixel ``(2i + py, 2j + px)`` sees the upsampled map
through only 2 x 2 distinct low-resolution pixels: the three taps of a row collapse onto two source rows

    py = 0:  rows 2i-1, 2i, 2i+1  ->  low-res rows i-1, i, i      weights [w0, w1 + w2]      (offsets -1, 0)
    py = 1:  rows 2i, 2i+1, 2i+2  ->  low-res rows i, i, i+1      weights [w0 + w1, w2]      (offsets 0, +1)

(and the same along x), so the upsampled part of the conv is FOUR 2 x 2 convolutions on the low-resolution map, one per output
phase: 4/9 of the multiply-accumulates.  Zero padding is preserved: the rows / columns outside the upsampled map correspond to
rows / columns outside the low-resolution map.  ``tests/test_host_logic.py::test_subpixel_decomposition_is_exact`` pins the
algebra.

The same idea for the BILINEAR x2 resize in front of the FPN models' refinement head (celldetection/models/cpn.py:277-278:
``F.interpolate(features, inputs.shape[2:], mode='bilinear', align_corners=False)`` followed by the k x k ReadOut conv,
commons.py:461-511): for an exact x2 the resized map is a fixed linear filter of the low-resolution map,

    up[2i]     = 0.25 x[i-1] + 0.75 x[i]          up[2i+1] = 0.75 x[i] + 0.25 x[i+1]        (per axis; PyTorch clamps at the edges)

so a 7-tap row of the conv over ``up`` is a 5-tap row over ``x`` for either output phase: the 7 x 7 conv becomes FOUR 5 x 5
convs on the low-resolution map (25 instead of 49 taps per output pixel) -- exact wherever the conv window does not reach
beyond the upsampled image (where the conv's zero padding, not the resize's edge clamp, applies): output rows / columns
[4, 2h - 4).  ``collapse_bilinear_taps`` / ``bilinear_conv_by_phases``; ``test_bilinear_subpixel_decomposition_is_exact``.
"""
import torch

__all__ = ['collapse_upsampled_taps', 'phase_padding', 'upsampled_conv_by_phases', 'collapse_bilinear_taps',
           'bilinear_conv_by_phases', 'bilinear_frame', 'BILINEAR_FRAME']


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


def bilinear_frame(k: int) -> int:
    """Full-resolution pixels along every image edge that the bilinear phase convs of a k x k conv do not produce (there the
    conv's zero padding cuts the window, or a low-resolution tap would leave the map): 4 for k = 7, 2 for k = 3."""
    return 2 * ((k // 2 + 1) // 2)


BILINEAR_FRAME = bilinear_frame(7)


def _bilinear_phase_matrix(k: int, p: int, dtype=torch.float64) -> torch.Tensor:
    """[k, k // 2 + 2]: coefficient of low-resolution tap m (offset m - (k // 2 + 1) // 2 ... see below) in full-resolution tap
    t of a k-tap conv row for output phase p: tap t reads up[2j + p + t - k//2]."""
    r = k // 2
    m0 = (p - r) // 2 - 1 if (p - r) % 2 == 0 else (p - r) // 2  # lowest low-resolution offset any tap reaches
    hi = (p + r) // 2 + (1 if (p + r) % 2 == 1 else 0)
    taps = hi - m0 + 1
    c = torch.zeros(k, taps, dtype=dtype)
    for t in range(k):
        q = p + t - r            # up index relative to 2j
        i, odd = q // 2, q % 2   # python floor division: q = 2 i + odd
        if odd == 0:             # up[2i] = .25 x[i-1] + .75 x[i]
            c[t, i - 1 - m0] += .25
            c[t, i - m0] += .75
        else:                    # up[2i+1] = .75 x[i] + .25 x[i+1]
            c[t, i - m0] += .75
            c[t, i + 1 - m0] += .25
    return c, m0


def collapse_bilinear_taps(weight: torch.Tensor) -> torch.Tensor:
    """weight [cout, cin, k, k] (k odd) of a conv applied to a x2 bilinear-upsampled (align_corners=False) input ->
    [2, 2, cout, cin, k2, k2], k2 = (k + 3) // 2: the kernel of output phase (py, px) on the low-resolution input; its taps sit
    at low-resolution offsets -(k2 // 2) .. k2 // 2 for BOTH phases (padding k2 // 2)."""
    if weight.ndim != 4 or weight.shape[2] != weight.shape[3] or weight.shape[2] % 4 != 3:
        raise ValueError('expected a [cout, cin, k, k] weight with k = 3 (mod 4): both output phases then share one symmetric '
                         'low-resolution support')
    k = int(weight.shape[2])
    k2 = (k + 3) // 2
    out = []
    for py in (0, 1):
        cy, my = _bilinear_phase_matrix(k, py)
        row = []
        for px in (0, 1):
            cx, mx = _bilinear_phase_matrix(k, px)
            assert cy.shape[1] == cx.shape[1] == k2 and my == mx == -(k2 // 2)
            row.append(torch.einsum('tm,octs,sn->ocmn', cy, weight.double(), cx))
        out.append(torch.stack(row))
    return torch.stack(out)  # float64


def bilinear_conv_by_phases(x_low: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """``F.conv2d(F.interpolate(x_low, scale_factor=2, mode='bilinear', align_corners=False), weight, padding=k // 2)`` as four
    k2 x k2 convolutions on ``x_low`` [N, cin, h, w] -> [N, cout, 2h, 2w].  Only the interior ``[F, 2h - F) x [F, 2w - F)``, F =
    ``bilinear_frame(k)``, equals the direct computation: the frame is where the conv's zero padding cuts the window."""
    import torch.nn.functional as F
    n, _, h, w = x_low.shape
    wc = collapse_bilinear_taps(weight)
    k2 = wc.shape[-1]
    out = x_low.new_zeros((n, weight.shape[0], 2 * h, 2 * w), dtype=torch.float64)
    for py in (0, 1):
        for px in (0, 1):
            out[:, :, py::2, px::2] = F.conv2d(x_low.double(), wc[py, px], padding=k2 // 2)
    return out.to(x_low.dtype)
