"""Model (de)serialisation and tiling helpers of the reference's ``cd.util`` that sit on the inference path.

File format = the reference's ``save_fetchable_model`` format (celldetection/util/util.py:545-560):
``torch.save({'cd.__version__', 'cd.models': {'model': class_name, 'kwargs': hparams, 'updated_kwargs': {...}},
'state_dict'})`` -- files written by the reference load here and vice versa.
"""
import os
from itertools import product
from os.path import isfile, splitext

import numpy as np
import torch

__all__ = ['dict2model', 'model2dict', 'load_model', 'fetch_model', 'save_fetchable_model', 'get_tiling_slices',
           'HOSTED_MODELS', 'HOST_TEMPLATE']

HOST_TEMPLATE = 'https://celldetection.org/torch/models/{name}.pt'  # celldetection/models/hosted.py:1
HOSTED_MODELS = {'ginoro': 'ginoro_CpnResNeXt101UNet-fbe875f1a3e5ce2c'}  # celldetection/models/hosted.py:2-4


def dict2model(conf, updated_kwargs=True, **kwargs):
    """celldetection/util/util.py:373-461 (class-name form and model-file form)."""
    from . import cpn as src
    if len(conf) == 1:
        key, = conf.keys()
        if key not in ('model', 'lightning_model') and getattr(src, key, None) is not None:
            return getattr(src, key)(**conf[key])

    def field(*names, default=None):  # first spelling of a field that the config holds (long form, short form)
        return next((conf[n] for n in names if n in conf), default)

    # precedence (lowest to highest): stored constructor kwargs < attributes changed after construction < caller's kwargs
    kw = dict(field('kwargs', 'kw', default={}))
    if updated_kwargs:
        kw.update(conf.get('updated_kwargs', {}))
    kw.update(kwargs)
    name = field('lightning_model', 'model')
    if name is None:
        raise AssertionError("the model config needs a 'model' (or 'lightning_model') entry")
    args = field('args', 'a', default=())
    if isfile(name):
        return load_model(name, **kw)
    if hasattr(src, name):
        return getattr(src, name)(*args, **kw)
    return fetch_model(name, **kw)


def _load_cd_format(m, pretrained=True, **kwargs):
    """celldetection/util/util.py:464-472."""
    assert isinstance(m, dict) and 'cd.models' in m.keys()
    strict = kwargs.pop('pretrained_strict', True)
    model = dict2model(m['cd.models'], **kwargs)
    if pretrained:
        model.load_state_dict(m['state_dict'], strict=strict)
    return model


def load_model(filename, map_location=None, **kwargs):
    """celldetection/util/util.py:474-479."""
    assert isfile(filename), f'Could not find file: {filename}'
    load_kwargs = kwargs.pop('load_kwargs', {})
    load_kwargs.setdefault('weights_only', False)
    m = torch.load(filename, map_location=map_location or 'cpu', **load_kwargs)
    if isinstance(m, dict) and 'cd.models' in m.keys():
        model = _load_cd_format(m, **kwargs)
        if map_location is not None and str(map_location) != 'cpu':
            model = model.to(map_location)
        return model
    return m


def fetch_model(name, map_location=None, **kwargs):
    """celldetection/util/util.py:482-509.  There is no network on the target systems: a local file (or a file in
    ``$CELLDETECTION_AMD_MODEL_DIR`` / the torch hub checkpoint cache named like the hosted model) is used."""
    if name.startswith('cd://'):
        name = name[len('cd://'):]
    name = HOSTED_MODELS.get(name, name)
    cands = [name] if isfile(name) else []
    base = name if splitext(name)[1] in ('.pt', '.pth', '.ckpt') else name + '.pt'
    for d in (os.environ.get('CELLDETECTION_AMD_MODEL_DIR'), os.path.join(torch.hub.get_dir(), 'checkpoints'), '.'):
        if d and isfile(os.path.join(d, os.path.basename(base))):
            cands.append(os.path.join(d, os.path.basename(base)))
    if not cands:
        if name.startswith('http'):
            m = torch.hub.load_state_dict_from_url(name, map_location=map_location or 'cpu', weights_only=False)
            return _load_cd_format(m, **kwargs) if isinstance(m, dict) and 'cd.models' in m else m
        raise FileNotFoundError(f'Model {name!r} not found locally (searched $CELLDETECTION_AMD_MODEL_DIR, torch hub '
                                f'cache, cwd) and cannot be downloaded from {HOST_TEMPLATE.format(name=name)} offline.')
    return load_model(cands[0], map_location=map_location, **kwargs)


def model2dict(model):
    """Serialisable description of a model: class name, constructor hyper-parameters and the attributes that were
    changed after construction (the ``cd.models`` entry of the reference file format, util.py:527-542)."""
    hparams = dict(model.hparams)
    changed = {}
    for name, ctor_value in hparams.items():
        if name not in vars(model):
            continue  # not an instance attribute (e.g. consumed by the plan builder)
        now = vars(model)[name]
        differs = now != ctor_value
        differs = bool(differs.any()) if hasattr(differs, 'any') else bool(differs)
        if differs:
            changed[name] = now
    return {'model': type(model).__name__, 'kwargs': hparams, 'updated_kwargs': changed}


def save_fetchable_model(model, filename, **kwargs):
    """celldetection/util/util.py:545-560 (without the hash suffix renaming)."""
    if not len(splitext(filename)[1]):
        filename += '.pt'
    sd = OrderedDictCPU(model.state_dict())
    torch.save({'cd.__version__': '0.4.9', 'cd.models': model2dict(model), 'state_dict': sd, **kwargs}, filename)
    return filename


def OrderedDictCPU(sd):
    from collections import OrderedDict
    return OrderedDict((k, v.detach().cpu()) for k, v in sd.items())


def _axis_tiles(length: int, crop: int, stride: int):
    """Tiles along one axis: (starts, stops, overlap with the previous tile, overlap with the next tile)."""
    if crop >= length:
        stops = np.array([length])
    else:
        n_steps = -(-(length - crop) // stride)  # ceil: tiles after the first one
        stops = np.minimum(crop + stride * np.arange(n_steps + 1), length)
    starts = np.maximum(stops - crop, 0)  # a tile that would stick out is shifted back: every tile is full-size
    prev_stop = np.concatenate((starts[:1], stops[:-1]))
    before = prev_stop - starts
    after = np.concatenate((before[1:], [0]))
    return starts, stops, before, after


def get_tiling_slices(size, crop_size, strides, return_overlaps=False):
    """Tiling table of the slide loop (semantics of celldetection/util/util.py:1305-1354, pinned by the golden tables in
    tests/golden/tiling.npz): per axis the tiles end at crop, crop + stride, ... clipped to the image; a tile that would
    stick out is shifted back so that all tiles are full-size; overlaps = (with the previous tile, with the next tile).
    Returns an iterator over the row-major cartesian product of the per-axis slices (+ overlaps) and the tile grid."""
    assert isinstance(size, (tuple, list))
    nd = len(size)
    crop_size = (crop_size,) * nd if np.isscalar(crop_size) else tuple(crop_size)
    strides = (strides,) * nd if np.isscalar(strides) else tuple(strides)
    per_axis_slices, per_axis_overlaps, grid = [], [], []
    for length, crop, stride in zip(size, crop_size, strides):
        starts, stops, before, after = _axis_tiles(int(length), int(crop), int(stride))
        per_axis_slices.append([slice(int(a), int(b)) for a, b in zip(starts, stops)])
        per_axis_overlaps.append([[int(a), int(b)] for a, b in zip(before, after)])
        grid.append(len(starts))
    if return_overlaps:
        return product(*per_axis_slices), product(*per_axis_overlaps), grid
    return product(*per_axis_slices), grid
