"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY: simulator of the fp8 (e4m3) conv graph of celldetection_amd.

The reference has no fp8 path (BASELINE.json configs[4] asks for one), so there is nothing of the reference to restate
here: this module restates OUR fp8 algorithm ("fp8 precision") in plain torch-CPU so that the HIP kernels
can be pinned against something tighter than "close to fp32": every activation tensor is an e4m3 code times a static
per-tensor scale, weights are e4m3 codes times a per-output-channel scale (``graph.pack(..., effective_weights=)``
returns them dequantised), accumulation is fp32/fp64, ReadOut tails are bf16.  Only ``tests/`` import it.
"""
import torch
import torch.nn.functional as F


def _q(t, scale):
    """value -> nearest e4m3 code * scale (saturating), computed like the kernels: fp32 multiply by 1/scale."""
    inv = torch.tensor(1.0 / scale, dtype=torch.float32)
    return (t.float() * inv).clamp(-448, 448).to(torch.float8_e4m3fn).float() * scale


def _act(t, act, act_scale):
    if act == 'relu':
        return F.relu(t)
    if act == 'sigmoid':
        return torch.sigmoid(t)
    if act == 'tanh_scaled':
        return torch.tanh(t) * act_scale
    fn = dict(sigmoid=torch.sigmoid, leaky_relu=F.leaky_relu, silu=F.silu, gelu=F.gelu, elu=F.elu, tanh=torch.tanh, hardswish=F.hardswish, mish=F.mish,
              selu=F.selu, softplus=F.softplus).get(act)
    return fn(t) if fn is not None else t


def calibrate(plan, state_dict, x):
    """Per-tensor activation scales max|x|/448 from an un-quantised fp32 walk of the plan (the HIP path takes them from
    a bf16 run, ``CPN.calibrate_fp8``); max-pool / bilinear outputs inherit the scale of their input."""
    from celldetection_amd import graph
    folded = [dict(zip(('w', 'b'), graph._fold(state_dict, op))) for op in plan.ops if op['op'] == 'conv']
    absmax = {}
    simulate(plan, state_dict, folded, None, x, _absmax=absmax)
    scales = [max(absmax.get(i, 0.), 1e-12) / 448. for i in range(len(plan.tensors))]
    for op in plan.ops:
        if op['op'] in ('maxpool', 'bilinear'):
            scales[op['dst']] = scales[op['src0']]
        elif op['op'] in ('input', 'input_stem'):  # like CPN.calibrate_fp8: inputs lie in [0, 1]
            scales[op['dst']] = 1. / 448.
    return scales


def simulate(plan, state_dict, effective_weights, act_scales, x, _absmax=None):
    """-> dict out_index -> fp32 NCHW head map.  ``plan``: graph.Plan; ``effective_weights``: list filled by
    ``graph.pack(plan, sd, 'cpu', 'fp8', act_scales, effective_weights=...)``; ``x``: fp32 NCHW input in [0, 1].
    (``_absmax``: dict -> no quantisation, records max|x| per tensor: the calibration walk.)"""
    T, outs, ei = {}, {}, 0
    if _absmax is not None:
        global _q
        real_q = _q

        def _q(t, scale):  # noqa: F811 -- identity during the calibration walk
            return t.float()
        act_scales = [1.] * len(plan.tensors)
    try:
        return _simulate(plan, state_dict, effective_weights, act_scales, x, T, outs, ei, _absmax)
    finally:
        if _absmax is not None:
            _q = real_q


def _stem_fast(x):
    """The executor's rule for the stem alternatives (csrc/cpn_abi.hip): the padded 4-channel bf16 layout (8 bytes per pixel,
    [H + 6][W + 8]) must fit the input tensor's storage (64 bytes per pixel in an fp8 plan)."""
    h, w = x.shape[-2:]
    return (h + 6) * (w + 8) * 8 <= h * w * 64


def _simulate(plan, state_dict, effective_weights, act_scales, x, T, outs, ei, _absmax):
    from celldetection_amd import graph
    up = lambda t: F.interpolate(t, scale_factor=2, mode='nearest')
    fast = _stem_fast(x)
    for op in plan.ops:
        kind = op['op']
        if op.get('alt') and (op['alt'] == 2) != fast:  # the stem alternative that does not run at this input size
            if kind == 'conv':
                ei += 1  # (its dequantised weights are still in the list)
            continue
        if kind == 'input_stem':  # bf16 copy of the input in the padded layout: no e4m3 quantisation
            T[op['dst']] = x.float().to(torch.bfloat16).float()
            continue
        if kind == 'stem7':  # conv 7x7 s2 + BN + ReLU in bf16 (csrc/stem.hip); only the OUTPUT is e4m3
            wf, bf = graph._fold(state_dict, op)
            y = F.relu(F.conv2d(T[op['src0']].double(), wf.float().to(torch.bfloat16).double(), bf, 2, 3).float())
            T[op['dst']] = _q(y, act_scales[op['dst']])
            if _absmax is not None:
                _absmax[op['dst']] = float(T[op['dst']].abs().max())
            continue
        if kind == 'input':
            T[op['dst']] = _q(x, act_scales[op['dst']])
        elif kind == 'act':  # elementwise activation op (hidden activation of a head other than ReLU): decode, act, re-quantise
            T[op['dst']] = _q(_act(T[op['src0']], op['act'], 0.), act_scales[op['dst']])
        elif kind == 'maxpool':
            T[op['dst']] = F.max_pool2d(T[op['src0']], op['k'], op['stride'], op['pad'])  # codes unchanged
        elif kind == 'bilinear':
            t = T[op['src0']]
            s = act_scales[op['src0']]
            if t.shape[2:] != x.shape[2:]:  # _equal_size(features, inputs), models/cpn.py:277-278
                t = F.interpolate(t, x.shape[2:], mode='bilinear', align_corners=False)
            T[op['dst']] = _q(t, s)
        else:
            e = effective_weights[ei]
            ei += 1
            if isinstance(op.get('sub'), tuple) and op['sub'][0] in ('blphase', 'blframe', 'phase', 'lateral'):
                continue  # members of a (bilinear) sub-pixel triple: the simulation follows the head op they restate
            if op['up0'] == 'bilinear':  # resize fused into the conv's loader: blended values are re-quantised with
                t0 = T[op['src0']]        # the source tensor's scale (codes), exactly like the separate op did
                xin = t0 if t0.shape[2:] == x.shape[2:] else _q(
                    F.interpolate(t0, x.shape[2:], mode='bilinear', align_corners=False), act_scales[op['src0']])
            else:
                xin = up(T[op['src0']]) if op['up0'] else T[op['src0']]
            if op['src1'] is not None:  # (a resized second source / residual takes the first source's / the output's size)
                xin = torch.cat((xin, F.interpolate(T[op['src1']], xin.shape[2:]) if op['up1'] else T[op['src1']]), 1)
            y = F.conv2d(xin.double(), e['w'], e['b'], op['stride'], op['pad'], 1, op['groups']).float()
            if op['res'] is not None:
                y = y + (F.interpolate(T[op['res']], y.shape[2:]) if op['res_up'] else T[op['res']])
            y = _act(y, op['act'], op['act_scale'])
            if op['dst'] is not None:
                T[op['dst']] = _q(y, act_scales[op['dst']])
            elif op.get('fuse'):
                fz = op['fuse']
                w2 = state_dict[fz['w'] + 'weight'].float().to(torch.bfloat16).float()
                z = F.conv2d(y.to(torch.bfloat16).float(), w2, state_dict[fz['w'] + 'bias'].float())
                outs[op['out_index']] = _act(z, fz['act'], fz['act_scale'])
            else:
                outs[op['out_index']] = y
        if _absmax is not None and op.get('dst') is not None:
            _absmax[op['dst']] = float(T[op['dst']].abs().max())
    return outs
