"""Deterministic synthetic weights for CPN models (no network => no pretrained checkpoints).

The generator is keyed by *state-dict key name* (crc32(key) ^ seed), not by key order, so the same
numbers come out for the reference model (``/root/reference`` imported in the build container by
``tests/golden/make_golden.py``) and for ``celldetection_amd`` models whose ``state_dict()`` mirrors the
reference's key names (celldetection/util/util.py:545-560 ``save_fetchable_model`` format).

Random-init CPNs produce ~0.5 scores everywhere => zero detections, so
``calibrate_heads`` rescales the four final 1x1 head convolutions to obtain a controlled detection density
and non-trivial Fourier / location / refinement outputs; the adjusted tensors are returned as
``overrides`` (small) so golden fixtures can store them verbatim.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

__all__ = ['synth_tensor', 'synth_state_dict', 'calibrate_heads', 'HEAD_FINAL_KEYS']

HEAD_FINAL_KEYS = tuple(f'core.{h}_head.block.4.{p}' for h in ('score', 'location', 'fourier', 'refinement')
                        for p in ('weight', 'bias'))


def _rng(key: str, seed: int):
    return np.random.default_rng([zlib.crc32(key.encode()) & 0xffffffff, seed & 0xffffffff])


def synth_tensor(key: str, shape, seed: int = 0, dtype=torch.float32) -> torch.Tensor:
    """One deterministic tensor for a state-dict entry, with statistics typical for a trained net."""
    shape = tuple(int(s) for s in shape)
    rng = _rng(key, seed)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    if key == 'order_weights':
        order = shape[0]
        x = np.arange(order, dtype=np.float32)
        spread = max(order - 1, 1)
        y = 1 + 4 * (1 - np.clip(x / spread, 0., 1.)) ** 2  # celldetection/ops/cpn.py:230-235
        return torch.as_tensor(y.reshape(shape), dtype=dtype)
    if leaf == 'running_var':
        v = rng.uniform(0.5, 1.5, shape)
    elif leaf == 'running_mean':
        v = rng.normal(0., 0.1, shape)
    elif leaf == 'weight' and len(shape) == 1:  # norm scale
        v = rng.uniform(0.6, 1.4, shape)
    elif leaf == 'bias':
        v = rng.normal(0., 0.05, shape)
    elif leaf == 'weight' and len(shape) >= 2:  # conv kernel: He-normal on fan_in
        fan_in = int(np.prod(shape[1:]))
        v = rng.normal(0., np.sqrt(2. / max(fan_in, 1)), shape)
    else:
        v = rng.normal(0., 0.1, shape)
    return torch.as_tensor(np.asarray(v, dtype=np.float32), dtype=dtype)


def synth_state_dict(template: 'OrderedDict[str, torch.Tensor]', seed: int = 0, overrides: dict = None):
    """Synthetic state dict with the keys/shapes of ``template`` (values of ``template`` are ignored)."""
    out = OrderedDict()
    for k, v in template.items():
        out[k] = synth_tensor(k, v.shape, seed, dtype=(v.dtype if v.dtype.is_floating_point else torch.float32)) \
            if v.dtype.is_floating_point else synth_tensor(k, v.shape, seed)
    if overrides:
        for k, v in overrides.items():
            out[k] = torch.as_tensor(v).to(out[k].dtype).reshape(out[k].shape).clone()
    return out


def calibrate_heads(state_dict, core_fn, score_shift: float = -2., score_gain: float = 3.,
                    fourier_std: float = 2., location_std: float = 1., refinement_raw_std: float = 1.):
    """Rescale the final 1x1 head convs so that a synthetic-weight CPN yields detections.

    Args:
        state_dict: full synthetic state dict (modified copy is returned).
        core_fn: callable(state_dict) -> (raw_scores[N,1,h,w], locations[N,2,h,w], refinement[N,2,H,W],
            fourier[N,4*order,h,w]) as float tensors (the ``CPNCore.forward`` outputs,
            celldetection/models/cpn.py:238-283).

    Returns:
        (new_state_dict, overrides) where ``overrides`` holds only the 8 adjusted tensors.
    """
    sd = OrderedDict((k, v.clone()) for k, v in state_dict.items())
    # the refinement map is only observable after tanh*3: shrink the final conv until it is unsaturated, so that
    # atanh() recovers the pre-activation statistics
    for _ in range(12):
        refinement = torch.as_tensor(core_fn(sd)[2]).float().cpu()
        if float((refinement.abs() > 2.4).float().mean()) < 0.01:
            break
        for p in ('weight', 'bias'):
            sd['core.refinement_head.block.4.' + p] = sd['core.refinement_head.block.4.' + p] * 0.1
    scores, locations, refinement, fourier = [torch.as_tensor(t).float().cpu() for t in core_fn(sd)]
    ov = {}

    def _apply(prefix, gain, new_bias_fn):
        w, b = sd[prefix + 'weight'], sd[prefix + 'bias']
        sd[prefix + 'weight'] = (w * gain).contiguous()
        sd[prefix + 'bias'] = new_bias_fn(b).contiguous()
        ov[prefix + 'weight'] = sd[prefix + 'weight']
        ov[prefix + 'bias'] = sd[prefix + 'bias']

    m, s = float(scores.mean()), float(scores.std()) + 1e-12
    g = score_gain / s
    if scores.shape[1] == 1:
        _apply('core.score_head.block.4.', g, lambda b: g * (b - m) + score_shift)
    else:  # multi-class logits: standardise every class plane, then favour class 0 (background) by |score_shift|
        mc = scores.mean((0, 2, 3))
        gc = score_gain / (scores.std((0, 2, 3)) + 1e-12)

        def _bias(b):
            nb = gc * (b - mc)
            nb[0] = nb[0] - score_shift
            return nb
        _apply('core.score_head.block.4.', gc[:, None, None, None], _bias)
    g = fourier_std / (float(fourier.std()) + 1e-12)
    _apply('core.fourier_head.block.4.', g, lambda b: b * g)
    g = location_std / (float(locations.std()) + 1e-12)
    _apply('core.location_head.block.4.', g, lambda b: b * g)
    raw_std = float(torch.atanh((refinement / 3.).clamp(-.999, .999)).std()) + 1e-12
    g = refinement_raw_std / raw_std
    _apply('core.refinement_head.block.4.', g, lambda b: b * g)
    return sd, ov
