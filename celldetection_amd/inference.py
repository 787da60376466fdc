"""Tiled large-slide CPN inference, sharded over the GPUs of one node.

Mirrors the tiling loop of the reference script (celldetection_scripts/cpn_inference.py:311-429, ``apply_model``):
``get_tiling_slices`` -> per-tile ``CPN.forward(offsets=...)`` -> ``remove_border_contours`` (+ optional stitching
rule) -> concatenate -> gather over ranks -> ONE global NMS over all detections of the slide.

MI355X-first differences (same results, different mechanics):
  * the slide stays resident in HBM as uint8; tiles are cropped on the device and converted to bf16 NHWC by the
    input kernel (no per-batch H2D copy, no DataLoader workers);
  * tiles are independent units: rank r processes tiles r, r+W, r+2W, ... (strided, like Lightning's distributed
    sampler in the reference) with NO data-path collective;
  * the only exchange is ONE packed variable-length all-gather (counts, then a padded [max_count, D] float32 payload
    of ~0.6 KB per detection) over RCCL/xGMI instead of the reference's per-key sequential send/recv to rank 0
    (cpn_inference.py:257-308); the global NMS then runs redundantly on every rank (the payload is KB..MB:
    latency-bound, so ring/all-reduce bandwidth considerations do not apply).
"""
from collections import OrderedDict
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .util import get_tiling_slices

__all__ = ['shard_tiles', 'pack_detections', 'unpack_detections', 'gather_detections', 'tiled_inference',
           'stitch_rule_batched',
           'ensemble_inference', 'forward_tiled', 'KEYS']

KEYS = ('contours', 'boxes', 'scores', 'classes', 'locations', 'fourier', 'contour_proposals')


def shard_tiles(num_tiles: int, rank: int, world_size: int) -> List[int]:
    """Strided tile -> rank assignment (every tile exactly once)."""
    return list(range(rank, num_tiles, world_size))


def pack_detections(d: Dict[str, torch.Tensor]) -> torch.Tensor:
    """dict of [K, ...] tensors -> one float32 [K, D] struct-of-rows buffer (classes stored as float, exact)."""
    k = d['scores'].shape[0]
    cols = [d[key].reshape(k, int(np.prod(d[key].shape[1:]))).to(torch.float32) for key in KEYS]
    return torch.cat(cols, 1)


def unpack_detections(buf: torch.Tensor, samples: int, order: int) -> 'OrderedDict[str, torch.Tensor]':
    k = buf.shape[0]
    widths = dict(contours=samples * 2, boxes=4, scores=1, classes=1, locations=2, fourier=order * 4,
                  contour_proposals=samples * 2)
    shapes = dict(contours=(k, samples, 2), boxes=(k, 4), scores=(k,), classes=(k,), locations=(k, 2),
                  fourier=(k, order, 4), contour_proposals=(k, samples, 2))
    out, o = OrderedDict(), 0
    for key in KEYS:
        t = buf[:, o:o + widths[key]].reshape(shapes[key])
        out[key] = t.to(torch.int64) if key == 'classes' else t
        o += widths[key]
    return out


def gather_detections(buf: torch.Tensor, group=None, _force: bool = False) -> torch.Tensor:
    """Variable-length all-gather of per-rank [K_r, D] buffers -> [sum K_r, D] in rank order (on every rank).
    Two collectives: all_gather of the counts, all_gather of the payload padded to the maximum count.
    (``_force``: run the collectives on a one-rank communicator too -- the single-GPU RCCL smoke test.)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return buf
    world = dist.get_world_size(group)
    if world == 1 and not _force:
        return buf
    count = torch.tensor([buf.shape[0]], dtype=torch.int64, device=buf.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if mx == 0:
        return buf
    padded = torch.zeros((mx, buf.shape[1]), dtype=buf.dtype, device=buf.device)
    padded[:buf.shape[0]] = buf
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded.contiguous(), group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


def _default_ops():
    from . import ops
    return ops.remove_border_contours_batched, stitch_rule_batched, ops.nms


def stitch_rule_batched(contours: torch.Tensor, image_index: torch.Tensor, stops: torch.Tensor,
                        offsets: torch.Tensor) -> torch.Tensor:
    """``filter_contours_by_stitching_rule(rule='ex_br')`` (celldetection/ops/cpn.py:293-325) for all detections of a
    batch: contour p (of tile ``image_index[p]``) is dropped iff every point has x >= stops[tile, 0] or
    y >= stops[tile, 1] after adding ``offsets[tile]`` (stops = (tile_size - overlap_with_next_tile) in xy order)."""
    b = image_index.long()
    c = contours + offsets[b][:, None].to(contours.dtype)
    return ~((c >= stops[b][:, None].to(contours.dtype)).any(-1).all(-1))


def _lists_to_flat(y, device):
    """Per-image lists (``CPN.forward`` contract) -> (flat dict incl. 'b', counts)."""
    counts = [int(t.shape[0]) for t in y['scores']]
    flat = {k: torch.cat(list(y[k])) for k in KEYS}
    flat['b'] = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32), torch.tensor(counts)).to(device)
    return flat, counts


COMPACT_EVERY = 8  # batches whose pre-filter detections are kept alive before the kept rows are selected (one host sync)


def _compact(pending, keys):
    """[(flat dict, keep mask, *extra per-row tensors)] -> ONE entry holding only the kept rows (bounds the memory of the
    slide loop: without it every pre-filter detection of the slide stays alive until the end)."""
    if not pending:
        return []
    sel = torch.cat([p[1] for p in pending]).nonzero().squeeze(1)  # the host sync of this group of batches
    flat = {k: torch.cat([p[0][k] for p in pending]).index_select(0, sel) for k in keys}
    extra = tuple(torch.cat([p[i] for p in pending]).index_select(0, sel) for i in range(2, len(pending[0])))
    return [(flat, torch.ones(sel.shape[0], dtype=torch.bool, device=sel.device)) + extra]


_warned_double_offset = [False]


def _undo_double_offset(model, flat, offs, dev):
    """Without refinement (``refinement_iterations == 0``) the reference's ``contours`` IS its ``contour_proposals``
    tensor, so the two in-place ``+= offsets`` of CPN.forward hit it twice (models/cpn.py:655-656,697-699) while boxes and
    locations receive the offset once.  ``CPN.forward`` reproduces that on purpose (golden ``noref_offs.*``); in the TILE
    LOOPS the doubled offset would displace every contour by its tile origin and make the border / stitching rules test
    shifted coordinates, so here one offset is taken back (documented divergence from the reference script, which returns
    the displaced contours for such models)."""
    if getattr(model, 'refinement', True) and getattr(model, 'refinement_iterations', 1) > 0:
        return flat
    if not _warned_double_offset[0]:
        import warnings
        warnings.warn('CPN without refinement in a tile loop: the reference adds the tile offset to the contours twice '
                      '(models/cpn.py:655-699); celldetection_amd corrects this in tiled_inference / forward_tiled.',
                      RuntimeWarning, stacklevel=3)
        _warned_double_offset[0] = True
    d = offs.to(torch.float32).to(dev)[flat['b'].long()][:, None]
    flat = dict(flat)
    flat['contours'] = flat['contours'] - d
    flat['contour_proposals'] = flat['contour_proposals'] - d
    return flat


def _non_empty_tiles(mask, slices, tile_ids):
    """Tiles whose mask crop holds a non-zero value (TileLoader, cpn_inference.py:88-100).  GPU masks: ONE window-any launch over
    the tiling table and one read-back (``ops.windows_any``) instead of a host synchronisation per tile (1849 on a 16384^2 slide);
    host masks (the CPU tests of the sharding logic) are indexed in place."""
    if not tile_ids:
        return []
    if mask.is_cuda:
        from . import ops
        flags = ops.windows_any(mask, [[slices[i][0].start, slices[i][0].stop, slices[i][1].start, slices[i][1].stop]
                                       for i in tile_ids])
        return [i for i, f in zip(tile_ids, flags) if f]
    return [i for i in tile_ids if bool(torch.any(mask[slices[i]]))]


def _sync(dev):
    if dev.type == 'cuda':
        torch.cuda.synchronize(dev)


@torch.no_grad()
def tiled_inference(model, img: torch.Tensor, crop_size=(1024, 1024), strides=(768, 768), batch_size: int = 8,
                    border_removal: int = 4, stitching_rule: str = 'nms', rank: Optional[int] = None,
                    world_size: Optional[int] = None, group=None, nms_thresh: Optional[float] = None,
                    forward_fn: Optional[Callable] = None, ops_fns=None, mask: Optional[torch.Tensor] = None,
                    point_mask: Optional[torch.Tensor] = None, point_mask_exclusive: bool = False,
                    timings: Optional[dict] = None, bug_compatible_offsets: bool = False):
    """Slide-level CPN inference.

    Args:
        model: ``celldetection_amd.models.CPN`` (on the GPU of this rank).
        img: slide as Tensor[C, H, W] or [1, C, H, W], uint8 or float in [0, 1], ideally already on the device.  Any
            size: a slide smaller than the crop is one tile of the slide's own size.
        crop_size, strides: tiling (celldetection_scripts/cpn_inference.py:451-452 defaults 1024 / 768).
        batch_size: tiles per forward.
        border_removal: contours touching the outer ``border_removal`` px of a tile side that has a neighbouring
            tile are dropped (cpn_inference.py:370-380).
        stitching_rule: 'nms' (global NMS) and/or 'ex_br' (cpn_inference.py:382-388,405-408).
        rank, world_size, group: tile sharding; default = torch.distributed state (single process if uninitialised).
        forward_fn / ops_fns: injection points used by the CPU (gloo) tests of the sharding/gather logic:
            ``forward_fn(tiles, offsets, **kw)`` -> per-image lists; ``ops_fns`` = (border_keep_batched(contours,
            image_index, sides, offsets, size, pad) -> mask, stitch_rule_batched, nms).
        bug_compatible_offsets: models WITHOUT refinement only.  The reference adds the tile offset to such a model's
            contours twice (models/cpn.py:655-656,697-699), so its script filters and returns contours displaced by their
            tile origin; by default this loop takes one offset back (``_undo_double_offset``).  True = bit-parity with the
            reference script's output for such models (a caller-supplied ``forward_fn`` is never corrected).
        mask: optional [H, W] (or [1, H, W]) foreground mask: tiles whose mask crop is empty are skipped and the crop
            is passed as ``scores_upper_bound`` (TileLoader semantics, cpn_inference.py:94-100).
        point_mask: optional [H, W] map of seed points: tiles without seeds are skipped, ``clip(crop, 0, 1)`` is passed
            as ``scores_lower_bound`` and, with ``point_mask_exclusive``, also as ``scores_upper_bound``
            (cpn_inference.py:102-113).
        timings: optional dict that receives wall-clock seconds of the phases ('tiles', 'gather', 'nms'; the device
            is synchronised at the phase boundaries when given) and the detection counts.

    Returns:
        OrderedDict of flat tensors (contours [K,S,2], boxes [K,4], scores [K], classes [K], locations [K,2],
        fourier [K,O,4], contour_proposals [K,S,2]) in global slide coordinates, identical on every rank.

    Host work is O(batches): every forwarded batch is filtered by ONE border-rule launch over all its detections; the
    kept rows of all batches are gathered once at the end (one host synchronisation for the whole filter).
    """
    import time
    import torch.distributed as dist
    if world_size is None:
        world_size = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if rank is None:
        rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
    border_fn, stitch_fn, nms_fn = ops_fns if ops_fns is not None else _default_ops()
    if img.ndim == 4:
        assert img.shape[0] == 1, 'one slide at a time'
        img = img[0]
    dev = img.device
    H, W = img.shape[-2:]
    crop_size = (crop_size,) * 2 if np.isscalar(crop_size) else tuple(crop_size)
    strides = (strides,) * 2 if np.isscalar(strides) else tuple(strides)
    slices, overlaps, shape = get_tiling_slices((H, W), crop_size, strides, return_overlaps=True)
    slices, overlaps = list(slices), list(overlaps)
    h_tiles, w_tiles = shape
    tile_ids = list(range(len(slices)))
    if mask is not None:
        mask = mask.reshape(mask.shape[-2:])
        tile_ids = _non_empty_tiles(mask, slices, tile_ids)
    if point_mask is not None:
        point_mask = point_mask.reshape(point_mask.shape[-2:])
        tile_ids = _non_empty_tiles(point_mask, slices, tile_ids)
    mine = [tile_ids[j] for j in shard_tiles(len(tile_ids), rank, world_size)]
    nms_thresh = model.nms_thresh if nms_thresh is None else nms_thresh
    rules = stitching_rule.split(',')
    samples = order = None
    meta = []  # FIFO of (tile ids, offsets, tile size) of the batches handed to the model
    t_start = time.perf_counter()

    def batches():
        for b0 in range(0, len(mine), batch_size):
            idxs = mine[b0:b0 + batch_size]
            tiles = torch.stack([img[(...,) + slices[i]] for i in idxs])  # cropped on the device
            offs = torch.tensor([[slices[i][1].start, slices[i][0].start] for i in idxs], dtype=torch.int64)
            kw = dict(offsets=offs)
            if mask is not None:
                kw['scores_upper_bound'] = torch.stack([mask[slices[i]] for i in idxs])[:, None].to(
                    device=tiles.device, dtype=torch.float32)
            if point_mask is not None:
                lb = torch.stack([point_mask[slices[i]] for i in idxs])[:, None].to(
                    device=tiles.device, dtype=torch.float32).clamp(0., 1.)
                kw['scores_lower_bound'] = lb
                if point_mask_exclusive:
                    kw['scores_upper_bound'] = lb
            meta.append((idxs, offs, tuple(tiles.shape[-2:])))
            yield tiles, kw

    if forward_fn is None:  # conv graph of batch i+1 overlaps the post-processing / border filtering of batch i
        results = model.forward_pipelined(batches(), flat_output=True)
    else:
        results = (_lists_to_flat(forward_fn(t, kw.pop('offsets'), **kw), dev) for t, kw in batches())
    pending = []  # (flat tensors of a batch, keep mask): rows are selected once, after the loop
    for flat, _ in results:
        idxs, offs, size = meta.pop(0)
        if timings is not None:  # host time at which this batch's detections were handed over (no extra synchronisation)
            timings.setdefault('batch_done_s', []).append(time.perf_counter() - t_start)
        samples, order = flat['contours'].shape[1], flat['fourier'].shape[1]
        if flat['scores'].shape[0] == 0:
            continue
        if forward_fn is None and not bug_compatible_offsets:
            flat = _undo_double_offset(model, flat, offs, dev)
        sides = []
        for i in idxs:  # sides that have a neighbouring tile (cpn_inference.py:372-380)
            h_i, w_i = np.unravel_index(i, shape)
            sides.append((1 if h_i > 0 else 0) | (2 if w_i < (w_tiles - 1) else 0) |
                         (4 if h_i < (h_tiles - 1) else 0) | (8 if w_i > 0 else 0))
        sides_t = torch.tensor(sides, dtype=torch.int32).to(dev, non_blocking=True)
        neg = (-offs).to(torch.float32).to(dev, non_blocking=True)
        keep = border_fn(flat['contours'], flat['b'], sides_t, neg, size, border_removal).bool()
        if 'ex_br' in rules:
            ov = torch.tensor([[overlaps[i][1][1], overlaps[i][0][1]] for i in idxs], dtype=torch.float32)
            stops = (torch.tensor([size[1], size[0]], dtype=torch.float32) - ov).to(dev, non_blocking=True)
            keep = keep & stitch_fn(flat['contours'], flat['b'], stops, neg)
        pending.append((flat, keep))
        if len(pending) >= COMPACT_EVERY:
            pending = _compact(pending, KEYS)
    if samples is None:  # this rank had no tiles: shapes from the model
        samples, order = model.samples, min(model.order, model.core.order)
    local = OrderedDict()
    shapes = dict(contours=(0, samples, 2), boxes=(0, 4), scores=(0,), classes=(0,), locations=(0, 2),
                  fourier=(0, order, 4), contour_proposals=(0, samples, 2))
    if pending:
        local.update(_compact(pending, KEYS)[0][0])  # (one host sync per COMPACT_EVERY batches)
    else:
        for k in KEYS:
            local[k] = torch.zeros(shapes[k], device=dev)
    if timings is not None:
        _sync(dev)
        timings['tiles'] = time.perf_counter() - t_start
        timings['tiles_local'] = len(mine)
        timings['detections_local'] = int(local['scores'].shape[0])
        t_start = time.perf_counter()
    buf = gather_detections(pack_detections(local), group=group)
    res = unpack_detections(buf, samples, order)
    if timings is not None:
        _sync(dev)
        timings['gather'] = time.perf_counter() - t_start
        timings['detections_gathered'] = int(res['scores'].shape[0])
        t_start = time.perf_counter()
    if 'nms' in rules and res['scores'].shape[0]:
        keep = nms_fn(res['boxes'], res['scores'], nms_thresh)
        res = OrderedDict((k, v[keep]) for k, v in res.items())
    if timings is not None:
        _sync(dev)
        timings['nms'] = time.perf_counter() - t_start
        timings['detections_final'] = int(res['scores'].shape[0])
    return res


@torch.no_grad()
def ensemble_inference(models, img: torch.Tensor, min_vote: float = 1, nms_thresh: Optional[float] = None, **kwargs):
    """Multi-model slide inference (celldetection_scripts/cpn_inference.py:311-427 with several ``models``): every
    model runs the tiled loop (incl. its own stitching NMS), the results are concatenated in model order, filtered by
    box voting when ``min_vote > 1`` (``filter_by_box_voting``, adds the ``votes`` key) and de-duplicated by one final
    NMS with ``nms_thresh`` (default: the last model's, like the reference)."""
    from . import ops
    models = list(models) if isinstance(models, (list, tuple)) else [models]
    parts = [tiled_inference(m, img, nms_thresh=nms_thresh, **kwargs) for m in models]
    if len(parts) == 1:
        return parts[0]
    res = OrderedDict((k, torch.cat([p[k] for p in parts])) for k in KEYS)
    thr = models[-1].nms_thresh if nms_thresh is None else nms_thresh
    if res['scores'].shape[0]:
        if min_vote > 1:
            keep, votes = ops.filter_by_box_voting(res['boxes'], thr, min_vote, return_votes=True)
            res = OrderedDict((k, v[keep.long()]) for k, v in res.items())
            res['votes'] = votes
        keep = ops.nms(res['boxes'], res['scores'], thr)
        res = OrderedDict((k, v[keep]) for k, v in res.items())
    return res


def _small_box_keep(boxes: torch.Tensor, min_size: float) -> torch.Tensor:
    """torchvision ``remove_small_boxes`` as a mask: both sides >= min_size (lightning_cpn.py:129)."""
    return ((boxes[:, 2] - boxes[:, 0]) >= min_size) & ((boxes[:, 3] - boxes[:, 1]) >= min_size)


@torch.no_grad()
def forward_tiled(model, inputs: torch.Tensor, crop_size=1024, stride=512, border_removal: int = 6,
                  min_box_size: float = 1., nms_thresh: Optional[float] = None, inputs_mask=None, batch_size: int = 8,
                  extra_keys=(), extra_nms=None, forward_fn: Optional[Callable] = None, ops_fns=None, **kwargs):
    """In-model tiling variant (``LitCpn.forward_tiled``, celldetection/models/lightning_cpn.py:88-177; pinned to outputs of the
    imported reference method by tests/golden/forward_tiled.npz): for a batch of large images, tiles (default 1024 / 512) ->
    forward (NO offsets: everything but contours / boxes stays tile-local, as there) -> ``remove_small_boxes(min 1.0)`` ->
    border removal (6 px) -> tile origin added to contours and boxes -> concat in tile order -> ONE NMS per image.
    ``inputs_mask``: a tile is skipped -- for every image of the batch -- iff its mask crop is empty in ALL images
    (``torch.any(crop_m)`` over the batch, lightning_cpn.py:114-117).  ``extra_keys``: further ``CPN.forward`` outputs to carry
    along (filtered like the others; ``extra_nms={key: False}`` exempts one from the final NMS selection); remaining ``kwargs`` go
    to the model's forward, like there.  Returns ``OrderedDict(contours, scores, boxes, *extra_keys)`` of per-image lists.
    ``forward_fn(tiles, **kw)`` -> per-image lists / ``ops_fns`` = (border_keep_batched, nms): injection points of the CPU tests.

    MI355X mechanics: tiles of all images are batched (``batch_size`` per conv-graph run, pipelined against the filtering of the
    previous batch); every forwarded batch is filtered by ONE border-rule launch; rows are selected once per COMPACT_EVERY
    batches."""
    n_img = inputs.shape[0]
    H, W = inputs.shape[-2:]
    crop = (crop_size,) * 2 if np.isscalar(crop_size) else tuple(crop_size)
    strd = (stride,) * 2 if np.isscalar(stride) else tuple(stride)
    assert (np.array(crop) <= np.array(strd) * 2).all()
    kwargs.pop('max_imsize', None)
    kwargs.pop('targets', None)
    slices, shape = get_tiling_slices((H, W), crop, strd)
    slices = list(slices)
    h_tiles, w_tiles = shape
    nms_thresh = getattr(model, 'nms_thresh', None) if nms_thresh is None else nms_thresh
    assert nms_thresh is not None, 'Could not retrieve nms_thresh from model. Please specify it in forward method.'
    extra_keys, extra_nms = tuple(extra_keys), dict(extra_nms or {})
    out_keys = ('contours', 'scores', 'boxes') + extra_keys
    if ops_fns is None:
        from . import ops
        border_fn, nms_fn = ops.remove_border_contours_batched, ops.nms
    else:
        border_fn, nms_fn = ops_fns
    dev = inputs.device
    tile_ids = list(range(len(slices)))
    if inputs_mask is not None:
        any_img = (inputs_mask != 0).reshape((-1,) + tuple(inputs_mask.shape[-2:])).any(0)
        tile_ids = _non_empty_tiles(any_img, slices, tile_ids)
    jobs = [(j, i) for i in tile_ids for j in range(n_img)]
    meta = []

    def batches():
        for b0 in range(0, len(jobs), batch_size):
            chunk = jobs[b0:b0 + batch_size]
            tiles = torch.stack([inputs[j][(...,) + slices[i]] for j, i in chunk])
            meta.append((chunk, tuple(tiles.shape[-2:])))
            yield tiles, dict(kwargs)

    if forward_fn is None:
        results = model.forward_pipelined(batches(), flat_output=True)
    else:
        def lists_to_flat(y):
            counts = [int(t.shape[0]) for t in y['scores']]
            flat = {k: torch.cat(list(y[k])) for k in out_keys}
            flat['b'] = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32), torch.tensor(counts)).to(dev)
            return flat, counts
        results = (lists_to_flat(forward_fn(t, **kw)) for t, kw in batches())
    pending = []
    for flat, _ in results:
        chunk, size = meta.pop(0)
        if flat['scores'].shape[0] == 0:
            continue
        sides = []
        for _, i in chunk:
            h_i, w_i = np.unravel_index(i, shape)
            sides.append((1 if h_i > 0 else 0) | (2 if w_i < (w_tiles - 1) else 0) |
                         (4 if h_i < (h_tiles - 1) else 0) | (8 if w_i > 0 else 0))
        sides_t = torch.tensor(sides, dtype=torch.int32).to(dev, non_blocking=True)
        zero = torch.zeros((len(chunk), 2), dtype=torch.float32, device=dev)
        keep = _small_box_keep(flat['boxes'], min_box_size)
        keep = keep & border_fn(flat['contours'], flat['b'], sides_t, zero, size, border_removal).bool()
        b = flat['b'].long()
        origin = torch.tensor([[slices[i][1].start, slices[i][0].start] for _, i in chunk], dtype=torch.float32).to(
            dev, non_blocking=True)[b]                      # (w_start, h_start) per detection
        rows = {k: flat[k] for k in out_keys}
        rows['contours'] = flat['contours'] + origin[:, None]             # lightning_cpn.py:139-142
        rows['boxes'] = flat['boxes'] + torch.cat((origin, origin), 1)
        img = torch.tensor([j for j, _ in chunk], dtype=torch.int64).to(dev, non_blocking=True)[b]
        pending.append((rows, keep, img))
        if len(pending) >= COMPACT_EVERY:
            pending = _compact(pending, out_keys)
    coll = [None] * n_img
    if pending:
        (cat, _, img_all), = _compact(pending, out_keys)
        # group the kept rows by image with ONE stable sort (tile order within an image is preserved) and one count
        # read-back, instead of a mask + nonzero (= a host sync) per image
        order_ = torch.sort(img_all, stable=True).indices
        counts = torch.bincount(img_all, minlength=n_img).tolist()
        o = 0
        for j in range(n_img):
            if counts[j]:
                sel = order_[o:o + counts[j]]
                coll[j] = {k: cat[k].index_select(0, sel) for k in cat}
            o += counts[j]
    final = OrderedDict((k, []) for k in out_keys)
    samples = getattr(model, 'samples', 32)
    order = min(getattr(model, 'order', 5), getattr(getattr(model, 'core', model), 'order', 5))
    empty = dict(contours=(0, samples, 2), scores=(0,), boxes=(0, 4), classes=(0,), locations=(0, 2), fourier=(0, order, 4),
                 contour_proposals=(0, samples, 2))
    for j in range(n_img):
        if coll[j] is not None:
            keep = nms_fn(coll[j]['boxes'], coll[j]['scores'], nms_thresh)
            for k in out_keys:
                final[k].append(coll[j][k][keep] if extra_nms.get(k, True) else coll[j][k])
        else:
            for k in out_keys:
                final[k].append(torch.zeros(empty.get(k, (0,)), device=dev, dtype=torch.int64 if k == 'classes' else torch.float32))
    return final
