"""celldetection_amd -- MI355X-native (gfx950, HIP) Contour Proposal Network inference path.

Drop-in for the CPN inference path of FZJ-INM1-BDA/celldetection: ``models.Cpn*`` / ``models.CPN``,
``fetch_model`` / ``load_model`` (reference file format), ``ops`` (decode / NMS) and the tiled slide inference
loop.  Everything computes in hand-written HIP kernels behind the C ABI of ``libcpn_hip.so``
(``include/cpn_hip.h``); there is no CPU fallback.
"""
from . import cpn as models  # ``cd.models.CpnResNeXt101UNet`` -> ``celldetection_amd.models.CpnResNeXt101UNet``
from . import h5, inference, labels, ops, preprocess, synth, util
from .h5 import from_h5, to_h5
from .labels import contours2labels
from .util import (dict2model, fetch_model, get_tiling_slices, load_model, model2dict, save_fetchable_model)

__version__ = '0.1.0'
__all__ = ['models', 'ops', 'util', 'synth', 'inference', 'labels', 'contours2labels', 'preprocess', 'h5', 'to_h5', 'from_h5', 'fetch_model', 'load_model', 'save_fetchable_model', 'dict2model',
           'model2dict', 'get_tiling_slices']
