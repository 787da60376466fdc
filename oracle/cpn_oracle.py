"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain CPU restatement (torch-CPU fp32 for the floating-point conv graph, numpy for the integer/index work)
of the reference's CPN inference path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / reported baseline -- the product
(``celldetection_amd``) never imports it and fails loudly when its HIP library is missing.

Pinning: the reference's own tests hold NO golden vectors for this path, so the oracle is
pinned against outputs of the reference itself, generated in the build container by
``tests/golden/make_golden.py`` (committed, with the ``.npz`` fixtures in ``tests/golden/``) and checked by
``tests/test_oracle_golden.py``.  The third-party pieces the reference calls but does not contain
(torchvision ``nms`` / FPN forward / ResNet block forward) are restated from their published semantics and are
"unpinned by the reference repository" (SURVEY.md section 8c).

Every function cites the reference file:line (relative to /root/reference) it follows.
"""
import ctypes
import os
import subprocess
from collections import OrderedDict
from itertools import product

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))


# =====================================================================================================
# 1. conv graph (floating point; torch CPU fp32, unfused conv -> BN -> ReLU exactly like the reference)
# =====================================================================================================

def _bn(x, sd, p):
    """nn.BatchNorm2d eval (running stats, eps 1e-5): lookup_nn('batchnorm2d'), models/commons.py:143-148."""
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'],
                        False, 0., 1e-5)


def _conv(x, sd, p, stride=1, padding=0, groups=1):
    return F.conv2d(x, sd[p + 'weight'], sd.get(p + 'bias'), stride, padding, 1, groups)


def _two_conv_norm_relu(x, sd, p):
    """TwoConvNormRelu: conv3x3-BN-ReLU x2, models/commons.py:120-149 (Sequential idx 0,1,2,3,4,5)."""
    x = F.relu(_bn(_conv(x, sd, p + '0.', padding=1), sd, p + '1.'))
    return F.relu(_bn(_conv(x, sd, p + '3.', padding=1), sd, p + '4.'))


def _cd_res_block(x, sd, p):
    """celldetection ``ResBlock`` (models/commons.py:259-359; the block of ResUNet, unet.py:434-464): conv3x3(no bias)-BN-ReLU-
    conv3x3(no bias)-BN + identity (a 1x1 ConvNorm without bias when the channel counts differ), ReLU."""
    identity = x
    if (p + 'downsample.0.weight') in sd:
        identity = _bn(_conv(x, sd, p + 'downsample.0.'), sd, p + 'downsample.1.')
    out = F.relu(_bn(_conv(x, sd, p + 'block.0.', padding=1), sd, p + 'block.1.'))
    out = _bn(_conv(out, sd, p + 'block.3.', padding=1), sd, p + 'block.4.')
    return F.relu(out + identity)


def _unet_block(x, sd, p):
    """The U-Net's block class, recognised from the keys: TwoConvNormRelu (``0.`` / ``1.`` / ``3.`` / ``4.``) or ResBlock
    (``block.*`` / ``downsample.*``)."""
    return _cd_res_block(x, sd, p) if (p + 'block.0.weight') in sd else _two_conv_norm_relu(x, sd, p)


def _has(sd, prefix):
    return any(k.startswith(prefix) for k in sd)


def _unet_encoder(x, sd, p):
    """UNetEncoder (U22 body): models/unet.py:29-58; block i>0 = Sequential(MaxPool2d(2,2), TwoConvNormRelu)."""
    feats = OrderedDict()
    i = 0
    while _has(sd, f'{p}{i}.'):
        if i == 0:
            x = _unet_block(x, sd, f'{p}0.')
        else:
            x = F.max_pool2d(x, 2, 2)
            x = _unet_block(x, sd, f'{p}{i}.1.')
        feats[str(i)] = x
        i += 1
    return feats


def _res_block(x, sd, p, stride):
    """torchvision BasicBlock / Bottleneck forward as used by models/resnet.py:56-116 (stride on conv1 for
    BasicBlock, on the 3x3 conv2 for Bottleneck; groups from the conv2 weight shape)."""
    identity = x
    if (p + 'conv3.weight') in sd:  # Bottleneck
        w2 = sd[p + 'conv2.weight']
        groups = w2.shape[0] // w2.shape[1] if w2.shape[1] != w2.shape[0] else 1
        groups = sd[p + 'conv1.weight'].shape[0] // w2.shape[1]
        out = F.relu(_bn(_conv(x, sd, p + 'conv1.'), sd, p + 'bn1.'))
        out = F.relu(_bn(_conv(out, sd, p + 'conv2.', stride=stride, padding=w2.shape[-1] // 2, groups=groups),
                         sd, p + 'bn2.'))
        out = _bn(_conv(out, sd, p + 'conv3.'), sd, p + 'bn3.')
    else:  # BasicBlock
        k = sd[p + 'conv1.weight'].shape[-1]
        out = F.relu(_bn(_conv(x, sd, p + 'conv1.', stride=stride, padding=k // 2), sd, p + 'bn1.'))
        out = _bn(_conv(out, sd, p + 'conv2.', padding=1), sd, p + 'bn2.')
    if (p + 'downsample.0.weight') in sd:
        identity = _bn(_conv(x, sd, p + 'downsample.0.', stride=stride), sd, p + 'downsample.1.')
    out = out + identity
    return F.relu(out)


def _resnet_body(x, sd, p):
    """ResNet with fused_initial=False (models/resnet.py:265-297, unet.py:584-587): children
    0 = Sequential(conv7x7 s2, BN, ReLU); 1 = Sequential(MaxPool(3,2,1), layer1); 2..4 = layer2..4."""
    feats = OrderedDict()
    x = F.relu(_bn(_conv(x, sd, p + '0.0.', stride=2, padding=3), sd, p + '0.1.'))
    feats['0'] = x
    x = F.max_pool2d(x, 3, 2, 1)
    stage = 1
    while _has(sd, f'{p}{stage}.'):
        sp = f'{p}{stage}.1.' if stage == 1 else f'{p}{stage}.'
        j = 0
        while _has(sd, f'{sp}{j}.'):
            x = _res_block(x, sd, f'{sp}{j}.', stride=2 if (j == 0 and stage > 1) else 1)
            j += 1
        feats[str(stage)] = x
        stage += 1
    return feats


def _generalized_unet(feats, sd, p, bridges):
    """GeneralizedUNet.forward, models/unet.py:178-249 (nearest upsample -> inner 1x1 -> cat(lateral, top_down)
    -> layer block; bridge levels have no lateral and upsample by scale_factor 2)."""
    x = list(feats.values())
    depth = 0
    while _has(sd, f'{p}layer_blocks.{depth}.'):
        depth += 1
    last_inner = x[-1]
    results = [last_inner]
    for i in range(depth - 1, -1, -1):
        has_lat = (i - bridges) >= 0
        lateral = x[i - bridges] if has_lat else None
        if lateral is not None:
            top = F.interpolate(last_inner, size=lateral.shape[2:], mode='nearest')
        else:
            top = F.interpolate(last_inner, scale_factor=2, mode='nearest')
        ip = f'{p}inner_blocks.{i}.'  # inner_blocks[i] belongs to loop index i+1 in __init__ (unet.py:118-128)
        if (ip + 'weight') in sd:
            top = _conv(top, sd, ip)
        inp = torch.cat((lateral, top), 1) if lateral is not None else top
        last_inner = _unet_block(inp, sd, f'{p}layer_blocks.{i}.')
        results.insert(0, last_inner)
    return OrderedDict((str(i), r) for i, r in enumerate(results))


def _fpn(feats, sd, p):
    """torchvision FeaturePyramidNetwork.forward with celldetection ConvNorm(norm=Identity) blocks
    (models/fpn.py:79-134): inner 1x1 (bias), top-down nearest + add, 3x3 layer conv (bias)."""
    x = list(feats.values())
    n = len(x)
    last_inner = _conv(x[-1], sd, f'{p}inner_blocks.{n - 1}.0.')
    results = [_conv(last_inner, sd, f'{p}layer_blocks.{n - 1}.0.', padding=1)]
    for idx in range(n - 2, -1, -1):
        lat = _conv(x[idx], sd, f'{p}inner_blocks.{idx}.0.')
        top = F.interpolate(last_inner, size=lat.shape[-2:], mode='nearest')
        last_inner = lat + top
        results.insert(0, _conv(last_inner, sd, f'{p}layer_blocks.{idx}.0.', padding=1))
    return OrderedDict((str(i), r) for i, r in enumerate(results))


def _hidden_act(name):
    """``lookup_nn(name)`` of the reference for the hidden activation of a ReadOut head: the torch.nn module of that name
    (case-insensitive) with default arguments (util/util.py:140-200)."""
    key = str(name).lower().replace('_', '')
    mods = {n.lower(): getattr(torch.nn, n) for n in ('ReLU', 'LeakyReLU', 'SiLU', 'GELU', 'ELU', 'Tanh', 'Sigmoid', 'Hardswish',
                                                      'Mish', 'SELU', 'Softplus', 'Identity')}
    return mods[key]()


def _readout(x, sd, p, stride=1, act='relu'):
    """ReadOut: conv kxk (bias, pad k//2, stride) -> BN -> activation (default ReLU) -> Dropout(eval: id) -> conv1x1,
    commons.py:461-511."""
    k = sd[p + 'block.0.weight'].shape[-1]
    x = _hidden_act(act)(_bn(_conv(x, sd, p + 'block.0.', stride=stride, padding=k // 2), sd, p + 'block.1.'))
    return _conv(x, sd, p + 'block.4.')


def _head_features(feats, keys, sd, fuse_prefix, fuse_kwargs=None):
    """_resolve_features (cpn.py:103-106) + Fuse2d (commons.py:640-674): a list of keys is resized (nearest) to the
    first feature's size, concatenated and passed through conv (1x1 by default) -> BN -> ReLU.  ``fuse_kwargs`` (cpn.py:173):
    kernel_size / padding / bias (read off the weights), ``norm_layer=None`` (no BN entries), ``activation`` (name or None)."""
    if not isinstance(keys, (list, tuple)):
        return feats[keys]
    xs = [feats[k] for k in keys]
    size = xs[0].shape[-2:]
    x = torch.cat([(F.interpolate(t, size) if t.shape[-2:] != size else t) for t in xs], 1)
    if len(xs) == 1:
        return x
    fk = dict(fuse_kwargs or {})
    x = _conv(x, sd, fuse_prefix + 'block.0.', padding=fk.get('padding', 0))
    if (fuse_prefix + 'block.1.running_mean') in sd:
        x = _bn(x, sd, fuse_prefix + 'block.1.')
    act = fk.get('activation', 'relu')
    return x if act is None else _hidden_act(act)(x)


def _resize(x, size, mode):
    """``_equal_size`` (models/cpn.py:109-115): F.interpolate(x, size, mode=mode, align_corners=False) -- torch rejects the
    align_corners argument for 'nearest', so that mode cannot run in the reference; 'nearest-exact' / 'area' would hit the same
    error.  Modes that do run there: bilinear, bicubic."""
    if x.shape[2:] == tuple(size):
        return x
    return F.interpolate(x, tuple(size), mode=mode, align_corners=False)


def core_forward(state_dict, x, refinement_margin=3., with_uncertainty=False, contour_head_stride=1,
                 refinement_head_stride=1, features=None, head_activations=None, refinement_interpolation='bilinear',
                 refinement_full_res=True, fuse_kwargs=None):
    """CPNCore.forward, models/cpn.py:238-283 -> (raw scores, locations, refinement, fourier), all fp32 NCHW
    (+ the sigmoid uncertainty map [N,4,h,w] or None as fifth element when ``with_uncertainty``).

    The backbone family is recognised from the state-dict keys (``unet.`` vs ``fpn.``; ``body.0.0.weight`` 7x7 =>
    ResNet stem, otherwise UNetEncoder)."""
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in state_dict.items()}
    x = x.float()
    # Normalize(mean 0, std 1, assert_range (0,1)): models/commons.py:694-700
    assert bool(torch.all(x >= 0.)) and bool(torch.all(x <= 1.)), 'Inputs should be in interval (0.0, 1.0)'
    p = 'core.backbone.'
    with torch.no_grad():
        if (p + 'body.0.0.weight') in sd and sd[p + 'body.0.0.weight'].shape[-1] == 7:
            feats = _resnet_body(x, sd, p + 'body.')
            bridges = 1
        else:
            feats = _unet_encoder(x, sd, p + 'body.')
            bridges = 0
        enc = feats
        if _has(sd, p + 'unet.'):
            feats = _generalized_unet(feats, sd, p + 'unet.', bridges)
            feats[str(len(feats))] = list(enc.values())[-1]  # the dict ends with the deepest encoder feature (unet.py:244)
            feats.update({f'encoder.{k}': v for k, v in enc.items()})  # keep_features (unet.py:247-248)
        else:
            feats = _fpn(feats, sd, p + 'fpn.')
        fk = dict(score='1', location='1', contour='1', uncertainty='1', refinement='0')
        fk.update(features or {})  # the <head>_features kwargs of CPNCore (cpn.py:135-139)
        hs = contour_head_stride
        ha = dict(score='relu', location='relu', fourier='relu', uncertainty='relu', refinement='relu')
        ha.update(head_activations or {})  # head_activation / head_activation_<head> (cpn.py:183-233)
        scores = _readout(_head_features(feats, fk['score'], sd, 'core.score_fuse.', fuse_kwargs), sd, 'core.score_head.', hs, ha['score'])
        locations = _readout(_head_features(feats, fk['location'], sd, 'core.location_fuse.', fuse_kwargs), sd, 'core.location_head.', hs,
                             ha['location'])
        fourier = _readout(_head_features(feats, fk['contour'], sd, 'core.fourier_fuse.', fuse_kwargs), sd, 'core.fourier_head.', hs,
                           ha['fourier'])
        uncertainty = None
        if _has(sd, 'core.uncertainty_head.'):  # cpn.py:209-221,266-271: ReadOut with final sigmoid
            uncertainty = torch.sigmoid(_readout(_head_features(feats, fk['uncertainty'], sd, 'core.uncertainty_fuse.', fuse_kwargs), sd,
                                                 'core.uncertainty_head.', hs, ha['uncertainty']))
        f0 = _head_features(feats, fk['refinement'], sd, 'core.refinement_fuse.', fuse_kwargs)
        if refinement_full_res:  # cpn.py:277-278
            f0 = _resize(f0, x.shape[2:], refinement_interpolation)
        refinement = torch.tanh(_readout(f0, sd, 'core.refinement_head.', refinement_head_stride, ha['refinement'])) * refinement_margin
        refinement = _resize(refinement, x.shape[2:], refinement_interpolation)  # cpn.py:279
    if with_uncertainty:
        return scores, locations, refinement, fourier, uncertainty
    return scores, locations, refinement, fourier


# =====================================================================================================
# 2. decode (numpy; index work bit-exact, float work in IEEE fp32 with the reference's operation order)
# =====================================================================================================

def sampling_table(order, samples, sampling=None):
    """sin/cos table of ops/cpn.py:66-78: t = linspace(0,1,S) (or the caller's ``sampling`` vector); c = float(pi)*2*k*t
    (fp32); cos/sin via torch CPU so that the bits equal the reference's CPU result."""
    t = torch.linspace(0, 1.0, samples) if sampling is None else torch.as_tensor(sampling, dtype=torch.float32).reshape(-1)
    c = float(np.pi) * 2 * (torch.arange(1, order + 1)[..., None]) * t[None]
    return torch.cos(c).numpy(), torch.sin(c).numpy()


def fouriers2contours(fourier, locations, samples, sampling=None):
    """ops/cpn.py:44-95: con = loc; con += sum_k(f[k,(1,3)]*sin_k); con += sum_k(f[k,(0,2)]*cos_k)
    (fp32 products, sum over k in ascending order).  ``sampling``: one sampling vector shared by all contours."""
    fourier = np.asarray(fourier, np.float32)
    locations = np.asarray(locations, np.float32)
    order = fourier.shape[-2]
    c_cos, c_sin = sampling_table(order, samples, sampling)
    samples = c_cos.shape[-1]
    con = np.zeros(fourier.shape[:-2] + (samples, 2), np.float32) + locations[..., None, :]
    for cols, tab in (((1, 3), c_sin), ((0, 2), c_cos)):
        acc = None
        for k in range(order):
            term = fourier[..., k, :][..., None, list(cols)] * tab[k][:, None]  # [..., S, 2]
            acc = term if acc is None else acc + term
        con = con + acc
    return con.astype(np.float32)


def refinement_buckets_table(samples, num_buckets):
    """ops/cpn.py:238-255 for the default sampling t = linspace(0, 1, S): three (bucket index [S], weight [S]) pairs
    for the buckets int(t*nb) - 1, int(t*nb), int(t*nb) + 1 (mod nb); weight = 1 - |j + .5 - t*nb|, 0 beyond 1."""
    base = torch.linspace(0, 1.0, samples) * num_buckets
    whole = base.long()
    out = []
    for j in (whole - 1, whole, whole + 1):
        dist = torch.abs(j + 0.5 - base)
        wgt = 1. - dist
        wgt[dist > 1] = 0
        out.append(((j % num_buckets).numpy(), wgt.numpy().astype(np.float32)))
    return out


def local_refinement(contours, refinement, b, iterations, size, num_buckets=1):
    """models/cpn.py:63-85: round-half-even, clamp, gather refinement[b,:,y,x], add.  Bucketed variant
    (cpn.py:72-82): the response of sample s is the weighted sum, in the order a, b, c, of the channel pairs of its
    three neighbouring buckets."""
    h, w = size
    c = np.asarray(contours, np.float32).copy()
    all_c = []
    table = refinement_buckets_table(c.shape[1], num_buckets) if num_buckets > 1 else None
    for _ in range(iterations):
        c = np.rint(c)
        c[..., 0] = np.clip(c[..., 0], 0, w - 1)
        c[..., 1] = np.clip(c[..., 1], 0, h - 1)
        idx = c.astype(np.int64)
        if table is None:
            resp = refinement[b[:, None], :, idx[:, :, 1], idx[:, :, 0]]  # [P, S, 2]
        else:
            resp = None
            for bi, bw in table:
                ch = np.stack((bi * 2, bi * 2 + 1), -1)  # [S, 2]
                cur = refinement[b[:, None, None], ch[None], idx[:, :, 1, None], idx[:, :, 0, None]]  # [P, S, 2]
                cur = (cur * bw[None, :, None]).astype(np.float32)
                resp = cur if resp is None else (resp + cur).astype(np.float32)
        c = (c + resp).astype(np.float32)
        all_c.append(c)
    return c, all_c


def nms_numpy(boxes, scores, thr):
    """torchvision CPU nms semantics (third-party; restated, unpinned): stable descending score order, greedy,
    suppress iff inter/(a_i+a_j-inter) > thr (NaN => keep)."""
    boxes = np.asarray(boxes, np.float32)
    scores = np.asarray(scores, np.float32)
    n = len(boxes)
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-scores.astype(np.float64), kind='stable')  # negation is exact; stable => ties in index order
    x1, y1, x2, y2 = boxes.T
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, bool)
    keep = []
    thr = np.float32(thr)
    with np.errstate(invalid='ignore', divide='ignore'):
        for _i in range(n):
            i = order[_i]
            if suppressed[i]:
                continue
            keep.append(i)
            rest = order[_i + 1:]
            w = np.maximum(np.float32(0), np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
            h = np.maximum(np.float32(0), np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, np.int64)


_NMS_C = None


def _nms_c_lib():
    """Compile/load oracle/nms_oracle.c (plain C restatement of the same greedy NMS)."""
    global _NMS_C
    if _NMS_C is None:
        so = os.path.join(_HERE, 'libnms_oracle.so')
        src = os.path.join(_HERE, 'nms_oracle.c')
        if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', src, '-o', so])
        lib = ctypes.CDLL(so)
        lib.nms_oracle.restype = ctypes.c_long
        lib.nms_oracle.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_float, ctypes.c_void_p]
        _NMS_C = lib
    return _NMS_C


def nms(boxes, scores, thr):
    """Greedy NMS through the C oracle (falls back to numpy when gcc is unavailable)."""
    boxes = np.ascontiguousarray(boxes, np.float32)
    scores = np.ascontiguousarray(scores, np.float32)
    n = len(boxes)
    if n == 0:
        return np.zeros((0,), np.int64)
    try:
        lib = _nms_c_lib()
    except Exception:
        return nms_numpy(boxes, scores, thr)
    order = np.ascontiguousarray(np.argsort(-scores.astype(np.float64), kind='stable').astype(np.int64))
    keep = np.empty(n, np.int64)
    k = lib.nms_oracle(boxes.ctypes.data, order.ctypes.data, n, ctypes.c_float(thr), keep.ctypes.data)
    return keep[:k].copy()


def batched_box_nmsi(boxes, scores, thr, batch_size=50000):
    """ops/cpn.py:189-227 incl. the chunked (> batch_size boxes) path + final NMS."""
    keeps = []
    for con, sco in zip(boxes, scores):
        n = len(con)
        if n <= batch_size:
            idx = nms(con, sco, thr)
        else:
            idx = np.zeros((0,), np.int64)
            for s in range(0, n, batch_size):
                e = min(s + batch_size, n)
                idx = np.concatenate((idx, nms(con[s:e], sco[s:e], thr) + s))
            if len(idx):
                idx = idx[nms(con[idx], sco[idx], thr)]
        keeps.append(idx)
    return keeps


def filter_by_box_voting(boxes, thresh, min_vote):
    """ops/boxes.py:52-83 (IoU = torchvision.ops.box_iou restated: inter / (area_i + area_j - inter)):
    votes = (iou * (iou > thresh)).sum(-1); keep = where(votes >= min_vote).  Returns (keep, votes[keep])."""
    b = np.asarray(boxes, np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(b[:, None, :2], b[None, :, :2])
    rb = np.minimum(b[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    with np.errstate(invalid='ignore', divide='ignore'):
        iou = inter / (area[:, None] + area[None] - inter)
        votes = (iou * (iou > np.float32(thresh)).astype(np.float32)).sum(-1, dtype=np.float32)
    keep = np.nonzero(votes >= min_vote)[0]
    return keep, votes[keep]


def _resize_bilinear(x, size):
    return F.interpolate(torch.as_tensor(x), size, mode='bilinear', align_corners=False).numpy()


def cpn_postprocess(scores_raw, locations, refinement, fourier, *, input_size, order=None, samples=32,
                    score_thresh=.9, nms_thresh=.2, refinement_iterations=4, nms=True, offsets=None,
                    scores_lower_bound=None, scores_upper_bound=None, scores_are_probabilities=False,
                    refinement_buckets=1, uncertainty=None, certainty_thresh=None, uncertainty_nms=False):
    """CPN.forward after the core, models/cpn.py:575-734 (eval mode): binary (one score plane, sigmoid) and
    multi-class (``classes`` planes, softmax/argmax, cpn.py:583-585) scores, bucketed refinement (cpn.py:72-82),
    uncertainty head (certainty filter cpn.py:617-618, ``box_uncertainties`` cpn.py:634-636, ``uncertainty_nms``
    cpn.py:723-726).

    Args are fp32 NCHW numpy arrays (or tensors).  Returns an OrderedDict of per-image lists like the reference.
    """
    to_np = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    scores_raw, locations, fourier = map(to_np, (scores_raw, locations, fourier))
    refinement = None if refinement is None else to_np(refinement)
    H, W = input_size
    n, c, h, w = fourier.shape
    multi = scores_raw.shape[1] > 2
    if scores_are_probabilities:
        scores = scores_raw.astype(np.float32)
    elif multi:
        scores = F.softmax(torch.as_tensor(scores_raw), dim=1).numpy()  # cpn.py:584
    else:
        scores = torch.sigmoid(torch.as_tensor(scores_raw)).numpy()  # cpn.py:578
    if scores_upper_bound is not None:  # cpn.py:118-123
        ub = to_np(scores_upper_bound).astype(np.float32)
        if ub.shape[2:] != scores.shape[2:]:
            ub = _resize_bilinear(ub, scores.shape[2:])
        scores = np.minimum(scores, ub)
    if scores_lower_bound is not None:
        lb = to_np(scores_lower_bound).astype(np.float32)
        if lb.shape[2:] != scores.shape[2:]:
            lb = _resize_bilinear(lb, scores.shape[2:])
        scores = np.maximum(scores, lb)
    if multi:
        classes = torch.argmax(torch.as_tensor(scores), dim=1).numpy().astype(np.int64)  # cpn.py:585
    else:
        classes = (scores > np.float32(score_thresh)).astype(np.int64)[:, 0]  # cpn.py:579
    fourier = fourier.reshape(n, c // 4, 4, h, w)  # cpn.py:594
    if order is not None and order < c // 4:
        fourier = fourier[:, :order]  # cpn.py:597-598
    # rel_location2abs_location, ops/cpn.py:15-41
    gx = np.arange(w, dtype=np.float32)[None] + np.zeros((h, 1), np.float32)
    gy = np.zeros((1, w), np.float32) + np.arange(h, dtype=np.float32)[:, None]
    locations = locations + np.stack((gx, gy), 0)
    fg = classes > 0
    unc = None if uncertainty is None else to_np(uncertainty).astype(np.float32)
    if certainty_thresh is not None and unc is not None:  # cpn.py:617-618
        fg = fg & (torch.as_tensor(unc).mean(1).numpy() < (1 - certainty_thresh))
    b, y, x = np.where(fg)  # row-major (b, y, x) order, cpn.py:620
    sel_fourier = fourier[b, :, :, y, x].astype(np.float32)  # [P, O, 4]
    sel_loc = locations[b, :, y, x].astype(np.float32)  # [P, 2]
    sel_classes = classes[b, y, x]
    sel_scores = scores[b, sel_classes, y, x] if multi else scores[b, 0, y, x]  # cpn.py:629-632
    sel_unc = None if unc is None else unc[b, :, y, x]  # cpn.py:634-636
    proposals = fouriers2contours(sel_fourier, sel_loc, samples)
    scale = np.array([W / w, H / h], np.float32)  # get_scale flip -> (x, y), ops/cpn.py:98-103
    proposals = proposals * scale
    sel_fourier = sel_fourier.copy()
    sel_fourier[..., [0, 1]] = sel_fourier[..., [0, 1]] * scale[0]
    sel_fourier[..., [2, 3]] = sel_fourier[..., [2, 3]] * scale[1]
    sel_loc = sel_loc * scale
    if refinement is not None and refinement_iterations > 0:
        contours, _ = local_refinement(proposals, refinement, b, refinement_iterations, (H, W),
                                       num_buckets=refinement_buckets)
    else:
        contours = proposals.copy()
    contours[..., 0] = np.clip(contours[..., 0], 0, W - 1)  # cpn.py:661-663 (also clamps the proposals when
    contours[..., 1] = np.clip(contours[..., 1], 0, H - 1)  # no refinement is applied: same tensor object)
    aliased = refinement is None or refinement_iterations <= 0  # cpn.py:655-656: ONE tensor object
    if aliased:
        proposals = contours
    if len(contours):
        boxes = np.concatenate((contours.min(1), contours.max(1)), 1)
    else:
        boxes = np.zeros((0, 4), np.float32)
    if offsets is not None:  # cpn.py:695-702
        offs = to_np(offsets)[b]
        if aliased:  # the two in-place `+=` of cpn.py:697-699 both hit the one tensor: offset added twice
            contours = (contours + offs[:, None].astype(np.float32)) + offs[:, None].astype(np.float32)
            proposals = contours
        else:
            contours = contours + offs[:, None].astype(np.float32)
            proposals = proposals + offs[:, None].astype(np.float32)
        boxes = boxes + np.tile(offs, (1, 2)).astype(np.float32)
        sel_loc = sel_loc + offs.astype(np.float32)
    flat = OrderedDict(contours=contours.astype(np.float32), boxes=boxes.astype(np.float32), scores=sel_scores,
                       classes=sel_classes, locations=sel_loc.astype(np.float32), fourier=sel_fourier,
                       contour_proposals=proposals.astype(np.float32))
    if sel_unc is not None:
        flat['box_uncertainties'] = sel_unc
    out = OrderedDict((k, [v[b == i] for i in range(n)]) for k, v in flat.items())  # cpn.py:42-50
    if nms:
        weights = out['scores']
        if uncertainty_nms and sel_unc is not None:  # cpn.py:723-726
            weights = [(torch.as_tensor(s_) * (1. - torch.as_tensor(u_).mean(1))).numpy()
                       for s_, u_ in zip(out['scores'], out['box_uncertainties'])]
        keeps = batched_box_nmsi(out['boxes'], weights, nms_thresh)
        out = OrderedDict((k, [v[i][keeps[i]] for i in range(n)]) for k, v in out.items())  # cpn.py:53-60
    if sel_unc is None:
        out['box_uncertainties'] = None
    return out


def cpn_forward(state_dict, x, **kw):
    """Full eval-mode CPN.forward (core + post-processing) on CPU."""
    s, l, r, f, u = core_forward(state_dict, x, refinement_margin=kw.pop('refinement_margin', 3.),
                                 with_uncertainty=True)
    return cpn_postprocess(s, l, r, f, input_size=tuple(x.shape[-2:]), uncertainty=u, **kw)


# =====================================================================================================
# 3. tiling / stitching (integer work)
# =====================================================================================================

def get_tiling_slices(size, crop_size, strides):
    """util/util.py:1305-1354 -> (list of per-tile ((h0,h1),(w0,w1)), list of per-tile overlaps, tiles per axis)."""
    nd = len(size)
    crop_size = (crop_size,) * nd if np.isscalar(crop_size) else tuple(crop_size)
    strides = (strides,) * nd if np.isscalar(strides) else tuple(strides)
    slices, shape, overlaps = [], [], []
    for ax in range(nd):
        if crop_size[ax] >= size[ax]:
            tl = [size[ax]]
        else:
            tl = list(range(crop_size[ax], 1 + crop_size[ax] + int(np.ceil((size[ax] - crop_size[ax]) / strides[ax]))
                            * strides[ax], strides[ax]))
        stops = np.minimum(tl, size[ax])
        starts = np.maximum(0, stops - crop_size[ax])
        ov_start = np.concatenate((starts[:1], stops[:-1])) - starts
        ov_end = np.concatenate((ov_start[1:], [0]))
        slices.append([(int(a), int(b_)) for a, b_ in zip(starts, stops)])
        overlaps.append([(int(a), int(b_)) for a, b_ in zip(ov_start, ov_end)])
        shape.append(len(starts))
    return list(product(*slices)), list(product(*overlaps)), shape


def remove_border_contours(contours, size, padding=1, top=True, right=True, bottom=True, left=True, offsets=None):
    """ops/cpn.py:258-290."""
    h, w = size[:2]
    c = np.asarray(contours, np.float32)
    if offsets is not None:
        c = c + np.asarray(offsets, np.float32)
    x, y = c[..., 0], c[..., 1]
    keep = np.ones(len(c), bool)
    if top:
        keep &= (y > padding).all(1)
    if right:
        keep &= (x < (w - padding)).all(1)
    if bottom:
        keep &= (y < (h - padding)).all(1)
    if left:
        keep &= (x > padding).all(1)
    return keep


def filter_contours_by_stitching_rule(contours, tile_size, overlaps, offsets=None):
    """ops/cpn.py:293-325, rule 'ex_br'."""
    c = np.asarray(contours, np.float32)
    if offsets is not None:
        c = c + np.asarray(offsets, np.float32)
    stop = (np.asarray(tile_size) - np.asarray(overlaps)[:, 1])[[1, 0]]
    right_bottom = (c >= stop).any(-1).all(-1)
    return ~right_bottom


def tiled_inference(state_dict, img, crop_size, strides, border_removal=4, **kw):
    """celldetection_scripts/cpn_inference.py:336-408 (single model, stitching_rule='nms'):
    tiles -> CPN.forward(offsets) -> remove_border_contours -> concat -> one global NMS."""
    nms_thresh = kw.get('nms_thresh', .2)
    H, W = img.shape[-2:]
    slices, overlaps, shape = get_tiling_slices((H, W), crop_size, strides)
    h_tiles, w_tiles = shape
    coll = {}
    keys = ('contours', 'boxes', 'scores', 'classes', 'locations', 'fourier', 'contour_proposals')
    for idx, ((h0, h1), (w0, w1)) in enumerate(slices):
        tile = img[..., h0:h1, w0:w1]
        offs = np.array([[w0, h0]], np.int64)
        y = cpn_forward(state_dict, tile, offsets=offs, **kw)
        h_i, w_i = np.unravel_index(idx, shape)
        keep = remove_border_contours(y['contours'][0], (h1 - h0, w1 - w0), border_removal, top=h_i > 0,
                                      right=w_i < w_tiles - 1, bottom=h_i < h_tiles - 1, left=w_i > 0,
                                      offsets=-offs[0])
        for k in keys:
            v = y[k][0][keep]
            coll[k] = np.concatenate((coll[k], v)) if k in coll else v
    keep = nms(coll['boxes'], coll['scores'], nms_thresh)
    return OrderedDict((k, v[keep]) for k, v in coll.items()), len(coll['scores'])
