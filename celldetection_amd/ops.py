"""Host-side mirror of ``celldetection.ops`` for the CPN inference path, backed by libcpn_hip.so.

Same names, argument meaning and return conventions as the reference (celldetection/ops/cpn.py, ops/boxes.py);
tensors must live on the GPU -- there is no CPU fallback in the product path (the CPU restatement lives in
``oracle/`` and is test infrastructure only).
"""
from ctypes import c_int32, c_int64, c_void_p
from typing import List

import numpy as np
import torch
from torch import Tensor

from . import _lib
from ._lib import check, ptr, stream_ptr

__all__ = ['fouriers2contours', 'local_refinement', 'nms', 'nms_binned', 'batched_box_nmsi', 'remove_border_contours',
           'remove_border_contours_batched',
           'filter_contours_by_stitching_rule', 'compact_scores', 'decode_proposals', 'sampling_tables',
           'bucket_tables', 'class_scores', 'certainty_mask', 'gather_channels', 'filter_by_box_voting',
           'NMS_BATCH_SIZE']

NMS_BATCH_SIZE = 50000  # celldetection/ops/cpn.py:12
# single-segment NMS calls with more boxes than this use the spatially binned formulation (identical keep list,
# O(P + E) memory instead of the dense P x P/64 bit mask: 32768 boxes = 134 MB of mask)
NMS_BINNED_MIN = 32768

_table_cache = {}


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('celldetection_amd ops run on the MI355X only (got a CPU tensor); '
                               'the CPU restatement is test infrastructure under oracle/.')


def sampling_tables(order: int, samples: int, device):
    """cos/sin tables exactly as the reference builds them on the CPU (celldetection/ops/cpn.py:66-78):
    t = linspace(0, 1, S); c = float(pi) * 2 * k * t; cos(c), sin(c) -> [order, samples] fp32."""
    key = (order, samples, str(device))
    if key not in _table_cache:
        t = torch.linspace(0, 1.0, samples)
        c = float(np.pi) * 2 * (torch.arange(1, order + 1)[..., None]) * t[None]
        if len(_table_cache) > 64:
            _table_cache.clear()
        _table_cache[key] = (torch.cos(c).contiguous().to(device), torch.sin(c).contiguous().to(device))
    return _table_cache[key]


def bucket_tables(samples: int, num_buckets: int, device):
    """Per-sample bucket numbers and blend weights of bucketed refinement for the default sampling
    t = linspace(0, 1, S), built on the CPU with the reference's arithmetic (resolve_refinement_buckets /
    refinement_bucket_weight, celldetection/ops/cpn.py:238-255): base = t * buckets; for j in (int(base) - 1,
    int(base), int(base) + 1): bucket j mod buckets with weight max(0, 1 - |j + 0.5 - base|) (0 beyond distance 1).
    Returns (int32 [3, S], float32 [3, S]) on ``device``."""
    key = ('buckets', samples, num_buckets, str(device))
    if key not in _table_cache:
        base = torch.linspace(0, 1.0, samples) * num_buckets
        whole = base.long()
        idx, wgt = [], []
        for j in (whole - 1, whole, whole + 1):
            dist = torch.abs(j + 0.5 - base)
            wgt.append(torch.where(dist > 1, torch.zeros_like(dist), 1. - dist))
            idx.append(j % num_buckets)
        _table_cache[key] = (torch.stack(idx).to(torch.int32).contiguous().to(device),
                             torch.stack(wgt).to(torch.float32).contiguous().to(device))
    return _table_cache[key]


def fouriers2contours(fourier: Tensor, locations: Tensor, samples: int = 64, sampling=None, cache=None):
    """celldetection/ops/cpn.py:44-95. fourier [..., order, 4], locations [..., 2] -> (contours [..., samples, 2], sampling).
    ``sampling``: optional sampling positions t shared by all contours, Tensor[samples] (or [1, ..., samples]); the cos / sin
    tables are built from it with the reference's expression (on the host, like the default table).  Per-contour samplings
    (``sampling[b]`` of the training targets, cpn.py:606,641) are a training-time feature and raise."""
    _need_cuda(fourier, locations)
    lead = fourier.shape[:-2]
    order = fourier.shape[-2]
    f = fourier.reshape(-1, order, 4).contiguous().float()
    loc = locations.reshape(-1, 2).contiguous().float()
    P = f.shape[0]
    if sampling is not None:
        sampling = torch.as_tensor(sampling)
        if sampling.numel() != sampling.shape[-1]:
            raise NotImplementedError('per-contour sampling tensors are a training-time feature (out of scope); pass one '
                                      'sampling vector shared by all contours')
        t = sampling.detach().reshape(-1).float().cpu()
        samples = int(t.shape[0])
        c = float(np.pi) * 2 * (torch.arange(1, order + 1)[..., None]) * t[None]  # ops/cpn.py:66-78
        cos_t, sin_t = torch.cos(c).contiguous().to(f.device), torch.sin(c).contiguous().to(f.device)
    else:
        cos_t, sin_t = sampling_tables(order, samples, f.device)
    out = torch.empty((P, samples, 2), dtype=torch.float32, device=f.device)
    check(_lib.load().cpn_fouriers2contours(ptr(f), ptr(loc), P, order, samples, ptr(cos_t), ptr(sin_t), ptr(out),
                                            stream_ptr()), 'fouriers2contours')
    return out.reshape(*lead, samples, 2), (sampling if sampling is not None else torch.linspace(0, 1.0, samples, device=f.device))


def local_refinement(contours: Tensor, refinement: Tensor, num_loops: int, b: Tensor, original_size=None,
                     num_buckets: int = 1):
    """celldetection/models/cpn.py:63-85 (default sampling). contours [P,S,2], refinement [N,2*buckets,H,W], b [P]."""
    _need_cuda(contours, refinement, b)
    c = contours.contiguous().float().clone()
    r = refinement.contiguous().float()
    N, ch, H, W = r.shape
    if ch != 2 * num_buckets:
        raise ValueError(f'refinement tensor must have 2 * num_buckets = {2 * num_buckets} channels, got {ch}')
    if original_size is not None and tuple(original_size) != (H, W):
        raise ValueError('refinement tensor must have the original size')
    bi = b.to(torch.int32).contiguous()
    bidx, bw = bucket_tables(c.shape[1], num_buckets, c.device) if num_buckets > 1 else (None, None)
    check(_lib.load().cpn_local_refinement(ptr(c), ptr(bi), c.shape[0], c.shape[1], ptr(r), N, H, W, int(num_loops),
                                           int(num_buckets), ptr(bidx), ptr(bw), stream_ptr()), 'local_refinement')
    return c


def _nms_segments(boxes: Tensor, scores: Tensor, seg_offsets: List[int], thresh: float):
    """-> (keep int64 [P] (per segment: global indices, written from the segment start), keep_counts list)."""
    lib = _lib.load()
    P = int(boxes.shape[0])
    nseg = len(seg_offsets) - 1
    dev = boxes.device
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    max_seg = max([seg_offsets[i + 1] - seg_offsets[i] for i in range(nseg)] + [0])
    ws_bytes = int(lib.cpn_nms_workspace_bytes(P, max_seg, nseg))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    keep = torch.empty(max(P, 1), dtype=torch.int64, device=dev)
    keep_counts = torch.empty(nseg, dtype=torch.int32, device=dev)
    host = (c_int64 * (nseg + 1))(*seg_offsets)
    dev_off = torch.tensor(seg_offsets, dtype=torch.int64, device=dev)
    check(lib.cpn_nms(ptr(boxes), ptr(scores), P, host, ptr(dev_off), nseg, float(thresh), ptr(keep), ptr(keep_counts),
                      ptr(ws), ws_bytes, stream_ptr()), 'nms')
    return keep, keep_counts.cpu().tolist()


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """torch.ops.torchvision.nms semantics (greedy, stable descending score order, NaN IoU never suppresses);
    returns kept indices (int64) in descending-score order.  Replaces celldetection/ops/cpn.py:211 and
    celldetection_scripts/cpn_inference.py:407.  Large sets (slide-level NMS) run through ``nms_binned`` -- the same
    keep list without the dense P x P/64 mask."""
    _need_cuda(boxes, scores)
    P = int(boxes.shape[0])
    if P == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if P >= NMS_BINNED_MIN and iou_threshold >= 0:
        return nms_binned(boxes, scores, iou_threshold)
    keep, counts = _nms_segments(boxes, scores, [0, P], iou_threshold)
    return keep[:counts[0]]


def nms_binned(boxes: Tensor, scores: Tensor, iou_threshold: float, return_stats: bool = False):
    """Spatially binned greedy NMS (csrc/nms_binned.hip): identical result to ``nms`` for ``iou_threshold >= 0``
    with O(P + E) workspace (E = overlapping higher-ranked pairs).  Synchronises the stream (edge count, convergence,
    keep count are read back).  ``return_stats``: also returns dict(edges, sweeps, workspace_bytes)."""
    _need_cuda(boxes, scores)
    lib = _lib.load()
    P = int(boxes.shape[0])
    dev = boxes.device
    if P == 0:
        empty = torch.empty((0,), dtype=torch.int64, device=dev)
        return (empty, dict(edges=0, sweeps=0, workspace_bytes=0)) if return_stats else empty
    bx = boxes.contiguous().float()
    sc = scores.contiguous().float()
    keep = torch.empty(P, dtype=torch.int64, device=dev)
    cnt, need, sweeps = c_int64(0), c_int64(0), c_int32(0)
    max_edges = max(32 * P, 1 << 16)
    for attempt in range(2):
        ws_bytes = int(lib.cpn_nms_binned_workspace_bytes(P, max_edges))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rc = lib.cpn_nms_binned(ptr(bx), ptr(sc), P, float(iou_threshold), max_edges, ptr(keep), None, cnt, need,
                                sweeps, ptr(ws), ws_bytes, stream_ptr())
        if rc == _lib.E_WORKSPACE and attempt == 0 and need.value > max_edges:
            max_edges = int(need.value)  # exact size reported by the count pass
            continue
        check(rc, 'nms_binned')
        break
    out = keep[:cnt.value]
    if return_stats:
        return out, dict(edges=int(need.value), sweeps=int(sweeps.value), workspace_bytes=ws_bytes)
    return out


def batched_box_nmsi(boxes: List[Tensor], scores: List[Tensor], iou_threshold: float, batch_size: int = None
                     ) -> List[Tensor]:
    """celldetection/ops/cpn.py:189-227 incl. the chunked path for > batch_size boxes.  All images that fit one
    batch are processed by ONE segmented NMS launch sequence."""
    if len(scores) != len(boxes):
        raise AssertionError(f'got {len(boxes)} box tensors but {len(scores)} score tensors')
    batch_size = NMS_BATCH_SIZE if batch_size is None else batch_size
    keeps = [None] * len(boxes)
    small = [i for i, b in enumerate(boxes) if b.shape[0] <= batch_size]
    if small:
        _need_cuda(*[boxes[i] for i in small])
        sizes = [int(boxes[i].shape[0]) for i in small]
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        if offs[-1] > 0:
            keep, counts = _nms_segments(torch.cat([boxes[i] for i in small]), torch.cat([scores[i] for i in small]),
                                         offs, iou_threshold)
            for j, i in enumerate(small):
                keeps[i] = keep[offs[j]:offs[j] + counts[j]] - offs[j]
        else:
            for i in small:
                keeps[i] = torch.empty((0,), dtype=torch.int64, device=boxes[i].device)
    for i in range(len(boxes)):
        if keeps[i] is not None:
            continue
        # image with more than batch_size boxes: NMS per chunk of batch_size boxes, then one NMS over the survivors
        # (the reference's memory-saving procedure, ops/cpn.py:212-224 -- NOT equivalent to one global NMS)
        total = int(boxes[i].shape[0])
        survivors = [nms(boxes[i][lo:lo + batch_size], scores[i][lo:lo + batch_size], iou_threshold) + lo
                     for lo in range(0, total, batch_size)]
        cand = torch.cat(survivors) if survivors else torch.zeros(0, dtype=torch.long, device=boxes[i].device)
        if cand.numel():
            cand = cand[nms(boxes[i][cand], scores[i][cand], iou_threshold)]
        keeps[i] = cand
    return keeps


def remove_border_contours(contours: Tensor, size, padding=1, top=True, right=True, bottom=True, left=True,
                           offsets=None) -> Tensor:
    """celldetection/ops/cpn.py:258-290 -> bool keep mask [num_contours]."""
    _need_cuda(contours)
    P, S = int(contours.shape[0]), int(contours.shape[1])
    keep = torch.empty(P, dtype=torch.uint8, device=contours.device)
    if P == 0:
        return keep.bool()
    h, w = size[:2]
    ox = oy = 0.
    if offsets is not None:
        o = torch.as_tensor(offsets).flatten().tolist()
        ox, oy = float(o[0]), float(o[1])
    sides = (1 if top else 0) | (2 if right else 0) | (4 if bottom else 0) | (8 if left else 0)
    c = contours.contiguous().float()
    check(_lib.load().cpn_border_keep(ptr(c), P, S, ox, oy, float(h), float(w), float(padding), sides, ptr(keep),
                                      stream_ptr()), 'remove_border_contours')
    return keep.bool()


def remove_border_contours_batched(contours: Tensor, image_index: Tensor, sides: Tensor, offsets: Tensor, size,
                                   padding=1) -> Tensor:
    """``remove_border_contours`` for all detections of a forwarded batch in ONE launch (the per-tile loop of
    celldetection_scripts/cpn_inference.py:370-380): contours [P,S,2]; image_index int32 [P] (tile of each contour);
    sides int32 [N] bit masks (1 top, 2 right, 4 bottom, 8 left = sides that have a neighbouring tile); offsets float
    [N,2] (xy) ADDED to the coordinates (the reference passes ``-tile_offset``).  -> uint8 keep mask [P] (device)."""
    _need_cuda(contours, image_index, sides, offsets)
    P, S = int(contours.shape[0]), int(contours.shape[1])
    keep = torch.empty(P, dtype=torch.uint8, device=contours.device)
    if P == 0:
        return keep
    h, w = size[:2]
    c = contours.contiguous().float()
    check(_lib.load().cpn_border_keep_batched(ptr(c), P, S, ptr(image_index.to(torch.int32).contiguous()),
                                              ptr(sides.to(torch.int32).contiguous()),
                                              ptr(offsets.to(torch.float32).contiguous()), int(sides.shape[0]),
                                              float(h), float(w), float(padding), ptr(keep), stream_ptr()),
          'remove_border_contours_batched')
    return keep


def windows_any(mask: Tensor, windows) -> list:
    """``[bool(mask[y0:y1, x0:x1].any()) for (y0, y1, x0, x1) in windows]`` in ONE launch and ONE read-back: the tile
    pre-filter of the slide loop (TileLoader skips tiles whose mask crop is empty, celldetection_scripts/cpn_inference.py:88-100;
    1849 windows on a 16384^2 slide).  ``mask``: [H, W] on the GPU, any dtype (non-float32 / non-byte masks are compared
    with zero first)."""
    _need_cuda(mask)
    assert mask.ndim == 2
    if mask.dtype == torch.bool:
        m = mask.contiguous().view(torch.uint8)
    elif mask.dtype in (torch.uint8, torch.float32):
        m = mask.contiguous()
    else:
        m = (mask != 0).contiguous().view(torch.uint8)
    win = torch.as_tensor(windows, dtype=torch.int32).reshape(-1, 4)
    n = int(win.shape[0])
    if n == 0:
        return []
    H, W = int(m.shape[0]), int(m.shape[1])
    if int(win[:, 0].min()) < 0 or int(win[:, 2].min()) < 0 or int(win[:, 1].max()) > H or int(win[:, 3].max()) > W:
        raise ValueError('windows_any: window outside the mask')
    win = win.to(m.device)
    out = torch.zeros(n, dtype=torch.int32, device=m.device)
    check(_lib.load().cpn_window_any(ptr(m), 0 if m.dtype == torch.float32 else 1, H, W, ptr(win), n, ptr(out), stream_ptr()),
          'window_any')
    return [bool(v) for v in out.cpu().tolist()]


def filter_contours_by_stitching_rule(contours: Tensor, tile_size, overlaps, rule='ex_br', offsets=None,
                                      indices=False):
    """celldetection/ops/cpn.py:293-325 (plain tensor arithmetic; not a hot spot)."""
    if not isinstance(tile_size, Tensor):
        tile_size = torch.as_tensor(tile_size, device=contours.device)
    overlaps = torch.as_tensor(overlaps, device=contours.device)
    if offsets is not None:
        contours = contours + torch.as_tensor(offsets, device=contours.device)
    if 'ex_br' in rule.split(','):
        stop = (tile_size - overlaps[:, 1])[[1, 0]]
        keep = ~((contours >= stop).any(-1).all(-1))
    else:
        raise ValueError(f'Unknown stitching rule: {rule}')
    if indices:
        keep, = torch.where(keep)
    return keep


def filter_by_box_voting(boxes: Tensor, thresh: float, min_vote: float, return_votes: bool = False):
    """celldetection/ops/boxes.py:61-83: a box receives as vote the IoU of every box (itself included) it overlaps with
    IoU > ``thresh``; boxes with ``votes >= min_vote`` are kept.  Returns keep indices (int32) [, their votes]."""
    _need_cuda(boxes)
    bx = boxes.contiguous().float()
    P = int(bx.shape[0])
    votes = torch.empty((P,), dtype=torch.float32, device=bx.device)
    check(_lib.load().cpn_box_votes(ptr(bx), P, float(thresh), ptr(votes), stream_ptr()), 'box_votes')
    mask = votes >= min_vote
    keep = torch.arange(P, device=bx.device, dtype=torch.int)[mask]
    if return_votes:
        return keep, votes[mask]
    return keep


def compact_scores(scores: Tensor, thresh: float, extra_flag: Tensor = None):
    """Ordered (b, y, x) indices of ``scores > thresh`` (replaces torch.where, celldetection/models/cpn.py:616-620).
    scores [N,1,h,w] fp32 -> (indices int32 [P] (device), per-image counts (host list), flag value or None).
    One host sync (the D2H copy of the counts), like the reference's torch.where."""
    _need_cuda(scores)
    if scores.ndim != 4 or scores.shape[1] != 1:
        raise ValueError(f'compact_scores expects a [N, 1, h, w] score map, got shape {tuple(scores.shape)}')
    lib = _lib.load()
    N, _, h, w = scores.shape
    s = scores.contiguous().float()
    dev = s.device
    indices = torch.empty(N * h * w, dtype=torch.int32, device=dev)
    counts = torch.empty(N + 1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.cpn_compact_workspace_bytes(N, h, w)), dtype=torch.uint8, device=dev)
    check(lib.cpn_compact(ptr(s), N, h, w, float(thresh), ptr(indices), ptr(counts), ptr(ws), stream_ptr()), 'compact')
    if extra_flag is not None:
        host = torch.cat((counts, extra_flag.reshape(1).to(torch.int32))).cpu().tolist()
        flag = host.pop()
    else:
        host, flag = counts.cpu().tolist(), None
    total = host[N]
    return indices[:total], host[:N], flag


def class_scores(logits: Tensor, lower=None, upper=None, return_probs: bool = False):
    """Multi-class score path (celldetection/models/cpn.py:583-585): softmax over the class planes, score bounds,
    argmax.  logits [N,C,h,w]; lower/upper [N,1,h,w] or None -> (selected [N,1,h,w] = probability of the argmax
    class, classes int32 [N,h,w], foreground float [N,1,h,w] (1 where class > 0), probs [N,C,h,w] or None)."""
    _need_cuda(logits, lower, upper)
    x = logits.contiguous().float()
    N, C, h, w = x.shape
    f32 = dict(dtype=torch.float32, device=x.device)
    sel, fg = torch.empty((N, 1, h, w), **f32), torch.empty((N, 1, h, w), **f32)
    cls = torch.empty((N, h, w), dtype=torch.int32, device=x.device)
    probs = torch.empty((N, C, h, w), **f32) if return_probs else None
    lo = None if lower is None else lower.contiguous().float()
    up = None if upper is None else upper.contiguous().float()
    check(_lib.load().cpn_class_scores(ptr(x), N, C, h, w, ptr(lo), ptr(up), ptr(probs), ptr(sel), ptr(cls), ptr(fg),
                                       stream_ptr()), 'class_scores')
    return sel, cls, fg, probs


def certainty_mask(scores: Tensor, uncertainty: Tensor, certainty_thresh: float):
    """scores where ``uncertainty.mean(1) < 1 - certainty_thresh`` else -1 (celldetection/models/cpn.py:617-618)."""
    _need_cuda(scores, uncertainty)
    s, u = scores.contiguous().float(), uncertainty.contiguous().float()
    N, C, h, w = u.shape
    out = torch.empty_like(s)
    check(_lib.load().cpn_certainty_mask(ptr(s), ptr(u), N, C, h, w, float(1 - certainty_thresh), ptr(out),
                                         stream_ptr()), 'certainty_mask')
    return out


def gather_channels(maps: Tensor, indices: Tensor):
    """maps [N,C,h,w] fp32, indices int32 [P] (linear (b,y,x) pixel indices of ``compact_scores``) -> [P,C]."""
    _need_cuda(maps, indices)
    m = maps.contiguous().float()
    _, C, h, w = m.shape
    P = int(indices.shape[0])
    out = torch.empty((P, C), dtype=torch.float32, device=m.device)
    check(_lib.load().cpn_gather_channels(ptr(m), ptr(indices.contiguous()), P, C, h, w, ptr(out), stream_ptr()),
          'gather_channels')
    return out


# score-gated heads: above this fraction of proposal pixels the dense head convs are run instead (same values; the
# gathered kernel re-reads the k x k neighbourhood per proposal, the dense one shares it between neighbouring pixels).
# Measured crossover on the MI355X (profiles/r03_sparse_density_sweep.txt, 16 x 256^2 head grid, 7x7 256->256): the
# gathered kernel is 1.76x faster at density 0.5 and 0.92x at 1.0 -> break-even at ~0.9
SPARSE_HEADS_MAX_DENSITY = .8


def dense_head(op, features_ptr: int, channel_stride: int, grid, weights: Tensor, bias: Tensor):
    """One fused ReadOut head conv (``_lib.OpDesc``, also a deferred one) densely over the NHWC bf16 tensor at
    ``features_ptr`` -> fp32 [N, cout, h, w]: what ``cpn_plan_run`` computes for a head that is not deferred."""
    _need_cuda(weights, bias)
    n, h, w = (int(v) for v in grid)
    out = torch.empty((n, int(op.fuse_cout), h, w), dtype=torch.float32, device=weights.device)
    check(_lib.load().cpn_conv2d(op, c_void_p(int(features_ptr)), int(channel_stride), ptr(None), 0, ptr(None), 0, ptr(out),
                                 0, n, h, w, ptr(weights), ptr(bias), stream_ptr()), 'conv2d')
    return out


def sparse_heads(op_a, op_b, features_ptr: int, channel_stride: int, grid, indices: Tensor, weights: Tensor,
                 bias: Tensor):
    """Score-gated ReadOut heads (csrc/sparse_heads.hip): the two fused head convs ``op_a`` / ``op_b`` (``_lib.OpDesc``
    of one plan, same NHWC bf16 source tensor at device address ``features_ptr`` with ``grid`` = (N, h, w)) evaluated only at
    the proposal pixels ``indices`` (``compact_scores``).  -> (out_a [P, cout_a], out_b [P, cout_b]) fp32: bit-identical to
    the dense head maps gathered at those pixels (CPN.forward reads nothing else of them, cpn.py:613-637)."""
    _need_cuda(indices, weights, bias)
    n, h, w = (int(v) for v in grid)
    P = int(indices.shape[0])
    f32 = dict(dtype=torch.float32, device=indices.device)
    out_a, out_b = torch.empty((P, int(op_a.fuse_cout)), **f32), torch.empty((P, int(op_b.fuse_cout)), **f32)
    if P:
        check(_lib.load().cpn_sparse_heads(op_a, op_b, c_void_p(int(features_ptr)), int(channel_stride), n, h, w,
                                           ptr(indices.contiguous()), P, ptr(weights), ptr(bias), ptr(out_a), ptr(out_b),
                                           stream_ptr()), 'sparse_heads')
    return out_a, out_b


def decode_proposals(indices: Tensor, scores: Tensor, locations: Tensor, fourier: Tensor, refinement, *, size,
                     order: int, samples: int, iterations: int, offsets=None, num_buckets: int = 1, gathered=False):
    """Fused proposal decode (celldetection/models/cpn.py:613-702): gather + rel->abs locations + Fourier synthesis +
    rescale + local refinement + clamp + boxes (+ offsets).  Returns a dict of flat [P, ...] tensors + 'b' [P].
    ``gathered``: ``locations`` [P, 2] / ``fourier`` [P, 4 * order_total] hold the head values of the proposals
    (``sparse_heads``) instead of dense [N, C, h, w] maps; the head grid is taken from ``scores``."""
    _need_cuda(indices, scores, locations, fourier)
    lib = _lib.load()
    H, W = size
    if gathered:
        N, (h, w), c4 = scores.shape[0], scores.shape[-2:], fourier.shape[1]
    else:
        N, c4, h, w = fourier.shape
    order_total = c4 // 4
    P = int(indices.shape[0])
    dev = fourier.device
    f32 = dict(dtype=torch.float32, device=dev)
    out = dict(contours=torch.empty((P, samples, 2), **f32), contour_proposals=torch.empty((P, samples, 2), **f32),
               boxes=torch.empty((P, 4), **f32), scores=torch.empty((P,), **f32), locations=torch.empty((P, 2), **f32),
               fourier=torch.empty((P, order, 4), **f32), b=torch.empty((P,), dtype=torch.int32, device=dev))
    if P == 0:
        return out
    cos_t, sin_t = sampling_tables(order, samples, dev)
    offs = None
    if offsets is not None:  # the reference adds the (int64) offsets to fp32 tensors: fp32 add of the converted value
        offs = torch.as_tensor(offsets).to(device=dev, dtype=torch.float32).contiguous()
        assert offs.shape == (N, 2), 'offsets must be Tensor[N, 2] (xy)'
    ref = None if refinement is None else refinement.contiguous().float()
    bidx, bw = bucket_tables(samples, num_buckets, dev) if (num_buckets > 1 and ref is not None) else (None, None)
    if ref is not None and ref.shape[1] != 2 * num_buckets:
        raise ValueError(f'refinement tensor must have {2 * num_buckets} channels, got {ref.shape[1]}')
    check((lib.cpn_decode_gathered if gathered else lib.cpn_decode)(
                         ptr(indices), P, ptr(scores.contiguous()), ptr(locations.contiguous()),
                         ptr(fourier.contiguous()), ptr(ref), N, h, w, H, W, order_total, order, samples,
                         int(iterations), ptr(cos_t), ptr(sin_t), ptr(offs), ptr(out['contours']),
                         ptr(out['contour_proposals']), ptr(out['boxes']), ptr(out['scores']), ptr(out['locations']),
                         ptr(out['fourier']), ptr(out['b']), int(num_buckets), ptr(bidx), ptr(bw), stream_ptr()),
          'decode')
    return out
