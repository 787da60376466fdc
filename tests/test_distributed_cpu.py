"""world_size-2 gloo test (CPU) of the multi-GPU path: tile sharding + packed variable-length all-gather + global
NMS must reproduce the single-process result.  The per-tile forward and the ops are injected (CPU oracle), so the
collective/sharding logic is exercised without a GPU."""
import os
import socket
from collections import OrderedDict

import numpy as np
import pytest
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
    rb = lambda con, size, pad, **kw: torch.as_tensor(orc.remove_border_contours(
        con.numpy(), size, pad, offsets=kw.pop('offsets').numpy().astype(np.float32), **kw))
    sf = lambda con, size, ov, rule, offsets: torch.as_tensor(orc.filter_contours_by_stitching_rule(
        con.numpy(), size, ov.numpy(), offsets=offsets.numpy().astype(np.float32)))
    nms = lambda b, s, t: torch.as_tensor(orc.nms(b.numpy(), s.numpy(), t))
    return rb, sf, nms


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
