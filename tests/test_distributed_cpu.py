"""world_size-2 gloo test (CPU) of the multi-GPU path: tile sharding + packed variable-length all-gather + global
NMS must reproduce the single-process result.  The per-tile forward and the ops are injected (CPU oracle), so the
collective/sharding logic is exercised without a GPU."""
import os
import socket
from collections import OrderedDict

import numpy as np
import torch
import torch.multiprocessing as mp

import cpn_oracle as orc
from celldetection_amd import inference

S, O = 8, 3


def fake_forward(tiles, offsets):
    """Deterministic synthetic detections per tile (a function of the tile offset only)."""
    out = OrderedDict((k, []) for k in inference.KEYS)
    for n in range(tiles.shape[0]):
        ox, oy = int(offsets[n, 0]), int(offsets[n, 1])
        rng = np.random.default_rng(ox * 7919 + oy)
        k = int(rng.integers(0, 9))
        ctr = rng.uniform(0, tiles.shape[-1], (k, 1, 2)).astype(np.float32)
        con = ctr + rng.uniform(-6, 6, (k, S, 2)).astype(np.float32) + np.array([ox, oy], np.float32)
        boxes = np.concatenate((con.min(1), con.max(1)), 1)
        out['contours'].append(torch.as_tensor(con))
        out['contour_proposals'].append(torch.as_tensor(con + 1))
        out['boxes'].append(torch.as_tensor(boxes))
        out['scores'].append(torch.as_tensor(rng.random(k).astype(np.float32)))
        out['classes'].append(torch.ones(k, dtype=torch.int64))
        out['locations'].append(torch.as_tensor(ctr[:, 0] + np.array([ox, oy], np.float32)))
        out['fourier'].append(torch.as_tensor(rng.standard_normal((k, O, 4)).astype(np.float32)))
    return out


def cpu_ops():
    """CPU stand-ins with the batched signatures of the product ops, built from the oracle's per-tile functions."""
    def border(contours, image_index, sides, offsets, size, pad):
        con, b = contours.numpy(), image_index.numpy()
        keep = np.zeros(len(con), bool)
        for n in np.unique(b):
            m = b == n
            sd = int(sides[n])
            keep[m] = orc.remove_border_contours(con[m], size, pad, top=bool(sd & 1), right=bool(sd & 2),
                                                 bottom=bool(sd & 4), left=bool(sd & 8),
                                                 offsets=offsets[n].numpy().astype(np.float32))
        return torch.as_tensor(keep)

    nms = lambda b, s, t: torch.as_tensor(orc.nms(b.numpy(), s.numpy(), t))
    return border, inference.stitch_rule_batched, nms


def reference_loop(img, crop, strides, border, thr):
    """The reference's per-tile loop (cpn_inference.py:357-408) restated with the oracle's per-tile functions."""
    slices, overlaps, shape = orc.get_tiling_slices(tuple(img.shape[-2:]), crop, strides)
    coll = {}
    for idx, ((h0, h1), (w0, w1)) in enumerate(slices):
        offs = torch.tensor([[w0, h0]])
        y = fake_forward(img[..., h0:h1, w0:w1], offs)
        h_i, w_i = np.unravel_index(idx, shape)
        con = y['contours'][0].numpy()
        neg = -offs[0].numpy().astype(np.float32)
        keep = orc.remove_border_contours(con, crop, border, top=h_i > 0, right=w_i < shape[1] - 1,
                                          bottom=h_i < shape[0] - 1, left=w_i > 0, offsets=neg)
        keep &= orc.filter_contours_by_stitching_rule(con, crop, np.array(overlaps[idx]), offsets=neg)
        for k in inference.KEYS:
            v = y[k][0].numpy()[keep]
            coll[k] = np.concatenate((coll[k], v)) if k in coll else v
    keep = orc.nms(coll['boxes'], coll['scores'], thr)
    return {k: v[keep] for k, v in coll.items()}


class _Model:
    nms_thresh, samples, order = .3, S, O

    class core:
        order = O


def run(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    img = torch.zeros(1, 3, 200, 328)
    res = inference.tiled_inference(_Model(), img, crop_size=(64, 96), strides=(48, 64), batch_size=3,
                                    forward_fn=fake_forward, ops_fns=cpu_ops(), stitching_rule='nms,ex_br')
    if rank == 0:
        q.put({k: v.numpy() for k, v in res.items()})
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_rank_gloo_matches_single_process():
    img = torch.zeros(1, 3, 200, 328)
    single = inference.tiled_inference(_Model(), img, crop_size=(64, 96), strides=(48, 64), batch_size=3,
                                       forward_fn=fake_forward, ops_fns=cpu_ops(), stitching_rule='nms,ex_br',
                                       rank=0, world_size=1)
    assert single['scores'].shape[0] > 10
    ref = reference_loop(img, (64, 96), (48, 64), 4, _Model.nms_thresh)  # batched filtering == the per-tile loop
    for k, v in single.items():
        np.testing.assert_array_equal(v.numpy(), ref[k], err_msg=k)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    multi = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # same detection SET (the concatenation order differs between strided sharding and one process; scores are
    # distinct, so the global NMS output order -- descending score -- is identical)
    for k, v in single.items():
        np.testing.assert_array_equal(multi[k], v.numpy(), err_msg=k)


def test_shard_tiles_partition():
    for n in (0, 1, 7, 1849):
        for w in (1, 2, 8):
            allt = sorted(i for r in range(w) for i in inference.shard_tiles(n, r, w))
            assert allt == list(range(n))


def test_pack_unpack_roundtrip():
    y = fake_forward(torch.zeros(1, 3, 64, 96), torch.tensor([[5, 9]]))
    d = {k: v[0] for k, v in y.items()}
    back = inference.unpack_detections(inference.pack_detections(d), S, O)
    for k in inference.KEYS:
        assert torch.equal(back[k], d[k]) and back[k].dtype == d[k].dtype


def test_slide_loop_reproduces_reference_stitching_with_duplicates():
    """Host logic of the slide loop (tiling, sharding x3, batched border rule, packing, global NMS) on the fixture with
    cross-tile duplicates: the result equals what the reference's functions produced (CPU stand-ins for the kernels)."""
    import stitch_fixture as sf
    g = sf.load()
    H, W = (int(i) for i in g['size'])
    img = torch.zeros(1, 3, H, W)
    kw = dict(crop_size=tuple(int(i) for i in g['crop']), strides=tuple(int(i) for i in g['stride']), batch_size=5,
              border_removal=int(g['border']), forward_fn=sf.forward_fn(g, 'cpu'), ops_fns=cpu_ops())
    res = inference.tiled_inference(sf.StubModel(), img, rank=0, world_size=1, **kw)
    assert res['scores'].shape[0] <= 0.9 * int(g['pre_nms_count'])
    for k in inference.KEYS:
        np.testing.assert_array_equal(res[k].numpy(), g[f'final.{k}'], err_msg=k)
    # sharded: the union of the three ranks' local results, gathered by hand, gives the same final set
    parts = [inference.tiled_inference(sf.StubModel(), img, rank=r, world_size=3, stitching_rule='', **kw) for r in range(3)]
    boxes = torch.cat([p['boxes'] for p in parts])
    scores = torch.cat([p['scores'] for p in parts])
    assert scores.shape[0] == int(g['pre_nms_count'])
    keep = orc.nms(boxes.numpy(), scores.numpy(), float(g['nms_thresh']))
    np.testing.assert_array_equal(boxes.numpy()[keep], g['final.boxes'])


def test_mid_loop_compaction_keeps_the_result(monkeypatch):
    """The slide loop selects the kept rows every COMPACT_EVERY batches (bounded memory: ADVICE r2); the result must not
    depend on where those compactions fall."""
    img = torch.zeros(1, 3, 200, 328)
    kw = dict(crop_size=(64, 96), strides=(48, 64), batch_size=3, forward_fn=fake_forward, ops_fns=cpu_ops(),
              stitching_rule='nms,ex_br', rank=0, world_size=1)
    ref = reference_loop(img, (64, 96), (48, 64), 4, _Model.nms_thresh)
    for every in (1, 2, 3, 1000):
        monkeypatch.setattr(inference, 'COMPACT_EVERY', every)
        got = inference.tiled_inference(_Model(), img, **kw)
        for k, v in got.items():
            np.testing.assert_array_equal(v.numpy(), ref[k], err_msg=f'{k} (COMPACT_EVERY={every})')
