"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY: ``normalize_percentile`` (celldetection/data/misc.py:156-161) and the inference
script's ``preprocess`` (celldetection_scripts/cpn_inference.py:196-222) with numpy.

PINNED (round 6): tests/golden/preprocess.npz holds outputs of the reference's own two functions, imported by
tests/golden/make_golden.py gen_preprocess; tests/test_preprocess.py checks this file against every case, bit for bit.
Third-party arithmetic inside them -- ``skimage.img_as_ubyte`` (clip(rint(x * 255), 0, 255) in float64), cv2's 8-bit luma,
albumentations' uint8 tables -- is absent from the image and from /root/reference: restated here and, identically, in the
stand-ins of oracle/ref_shim.py the import ran through (third party: unpinned).  ``to_uint8=False`` involves numpy only."""
import numpy as np


def normalize_percentile(image, percentile=99.9, to_uint8=True):
    if not isinstance(percentile, (list, tuple)):
        percentile = (100 - percentile, percentile)
    low, high = np.percentile(image, percentile)
    img = (np.clip(image, low, high) - low) / (high - low)
    if not to_uint8:
        return img
    return np.clip(np.rint(img.astype(np.float64) * 255.), 0, 255).astype(np.uint8)


def preprocess(img, gamma=1., contrast=1., brightness=0., percentile=None, grayscale=False):
    """``preprocess`` of celldetection_scripts/cpn_inference.py:196-222 on a channels-LAST numpy image (the script's layout) with
    the third-party calls restated from their published behaviour (cv2 / albumentations are absent: unpinned): cv2's 8-bit
    RGB2GRAY is the fixed-point luma ``(4899 R + 9617 G + 1868 B + 8192) >> 14``, ``gamma_transform`` and
    ``brightness_contrast_adjust`` are uint8 tables applied with cv2.LUT (albumentations 1.x: ``beta`` scales ``alpha * mean(img)``)."""
    if percentile is not None:
        img = normalize_percentile(img, percentile)
    if img.itemsize > 1:
        img = normalize_percentile(img)
    if grayscale and img.ndim == 3:
        c = img.shape[-1]
        if c == 1:
            img = img.squeeze(-1)
        elif c == 2:  # the script averages the two channels -> float64, which cv2.cvtColor(GRAY2RGB) rejects
            raise TypeError('preprocess(grayscale=True): 2-channel images reach cv2.cvtColor as float64 (unsupported depth)')
        elif c in (3, 4):
            x = img.astype(np.int64)
            img = ((x[..., 0] * 4899 + x[..., 1] * 9617 + x[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)
        else:
            raise ValueError(f'Unsupported number of channels: {c}')
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, -1)
    if gamma != 1.:
        table = (np.arange(0, 256.0 / 255, 1.0 / 255) ** gamma) * 255
        img = table.astype(np.uint8)[img]
    if contrast != 1.:
        lut = np.arange(0, 256).astype('float32')
        lut *= contrast
        if brightness != 0:
            lut += (contrast * brightness) * np.mean(img)
        img = np.clip(lut, 0, 255).astype(np.uint8)[img]
    return img
