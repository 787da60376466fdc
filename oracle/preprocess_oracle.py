"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY: ``normalize_percentile`` (celldetection/data/misc.py:156-161) with numpy.

``np.percentile`` is numpy's own (available here); ``skimage.util.img_as_ubyte`` is third-party and absent from the image
and from /root/reference -- its float -> uint8 rule (clip(rint(x * 255), 0, 255), computed in float64) is restated:
unpinned third party."""
import numpy as np


def normalize_percentile(image, percentile=99.9, to_uint8=True):
    if not isinstance(percentile, (list, tuple)):
        percentile = (100 - percentile, percentile)
    low, high = np.percentile(image, percentile)
    img = (np.clip(image, low, high) - low) / (high - low)
    if not to_uint8:
        return img
    return np.clip(np.rint(img.astype(np.float64) * 255.), 0, 255).astype(np.uint8)
