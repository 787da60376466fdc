"""Host side of the score-gated ReadOut heads (csrc/sparse_heads.hip; opt-in ``model.sparse_heads = True``): plan
construction, packing and the executor's planning -- everything that needs no GPU."""
from ctypes import c_int32, c_int64, c_void_p

import pytest
import torch

import celldetection_amd as cda
from celldetection_amd import _lib, graph


def _plans(backbone='ResNeXt101UNet', **kw):
    return graph.build_plan(backbone, 3, **kw), graph.build_plan(backbone, 3, sparse_heads=True, **kw)


def test_plan_defers_location_and_fourier_heads_only():
    dense, sparse = _plans()
    assert dense.meta['sparse_heads'] is None and all(not op.get('deferred') for op in dense.ops)
    meta = sparse.meta['sparse_heads']
    ia, ib = meta['ops']
    assert [i for i, op in enumerate(sparse.ops) if op.get('deferred')] == [ia, ib]
    a, b = sparse.ops[ia], sparse.ops[ib]
    assert a['w'] == 'core.location_head.block.0.' and b['w'] == 'core.fourier_head.block.0.'
    assert a['src0'] == b['src0'] == meta['src'] and a['fuse']['cout'] == 2 and b['fuse']['cout'] == 20
    # same parameters in the same order (one checkpoint serves both plans), same ops otherwise
    assert sparse.entries == dense.entries
    strip = lambda ops: [{k: v for k, v in op.items() if k != 'deferred'} for op in ops]
    assert strip(sparse.ops) == strip(dense.ops)
    # the algorithmic (reference-graph) FLOP count does not change
    assert graph.reference_flops(sparse, 512, 512) == graph.reference_flops(dense, 512, 512)


@pytest.mark.parametrize('backbone,kw', [('U22', dict(contour_head_channels=64)),  # hidden width 64
                                         ('ResNet18FPN', dict(contour_head_stride=2)),
                                         ('ResNeXt101UNet', dict(features=dict(contour='0'))),
                                         ('ResNeXt101UNet', dict(kernel_sizes=dict(location=3))),
                                         ('ResNet18FPN', dict(fuse_readout=False))])
def test_plans_that_do_not_qualify_stay_dense(backbone, kw):
    plan = graph.build_plan(backbone, 3, sparse_heads=True, **kw)
    assert plan.meta['sparse_heads'] is None and all(not op.get('deferred') for op in plan.ops)


def _native(plan):
    sd = {}
    g = torch.Generator().manual_seed(0)
    for key, shape, kind in plan.entries:
        sd[key] = torch.zeros(shape, dtype=torch.long) if kind == 'long' else torch.rand(shape, generator=g) + .5
    tens, ops, wblob, bblob = graph.pack(plan, sd, 'cpu')
    lib = _lib.load()
    handle = c_void_p()
    _lib.check(lib.cpn_plan_create(handle, tens, len(tens), ops, len(ops), _lib.ptr(wblob), wblob.numel() * 2,
                                   _lib.ptr(bblob), bblob.numel(), _lib.PRECISION_BF16), 'plan_create')
    return lib, handle, tens, ops, (wblob, bblob)


def test_executor_keeps_the_head_source_alive_and_skips_the_heads():
    dense, sparse = _plans('ResNet18FPN')
    lib, hd, _, ops_d, keep_d = _native(dense)
    lib, hs, tens, ops_s, keep_s = _native(sparse)
    try:
        ia, ib = sparse.meta['sparse_heads']['ops']
        assert ops_s[ia].op == ops_s[ib].op == _lib.OP_CONV_DEFERRED and ops_d[ia].op == _lib.OP_CONV
        n, h, w = 2, 96, 160
        # the deferred heads are not executed ...
        fd, fs = lib.cpn_plan_executed_flops(hd, n, h, w), lib.cpn_plan_executed_flops(hs, n, h, w)
        heads = sum(2. * n * (h // 4) * (w // 4) * 256 * 256 * 49 for _ in range(2))
        assert fd > fs > 0 and abs((fd - fs) - heads) < 1e-6 * fd
        # ... but still report their output size
        oh, ow = c_int32(0), c_int32(0)
        _lib.check(lib.cpn_plan_output_dims(hs, h, w, _lib.OUT_FOURIER, oh, ow), 'output_dims')
        assert (oh.value, ow.value) == (h // 4, w // 4)

        def info(handle, t):
            off, th, tw, cs = c_int64(0), c_int32(0), c_int32(0), c_int32(0)
            _lib.check(lib.cpn_plan_tensor_info(handle, n, h, w, t, off, th, tw, cs), 'tensor_info')
            return off.value, th.value, tw.value, cs.value

        src = sparse.meta['sparse_heads']['src']
        off, th, tw, cs = info(hs, src)
        assert (th, tw, cs) == (h // 4, w // 4, 256)
        lo, hi = off, off + n * th * tw * cs * 2
        assert hi <= lib.cpn_plan_workspace_bytes(hs, n, h, w)
        # no tensor written after the head source may share its bytes (it is read after the run)
        written_after = {op['dst'] for op in sparse.ops[[i for i, op in enumerate(sparse.ops) if op.get('dst') == src][0] + 1:]
                         if op.get('dst') is not None}
        for t in written_after:
            o, a, b, c = info(hs, t)
            assert o >= hi or o + n * a * b * c * 2 <= lo, f'tensor {t} overlaps the deferred heads\' source'
        with pytest.raises(RuntimeError):
            info(hs, len(tens))
    finally:
        lib.cpn_plan_destroy(hd)
        lib.cpn_plan_destroy(hs)


def test_model_switch_selects_the_plan():
    m = cda.models.CpnResNet18FPN(3)
    assert m.plan_for('bf16').meta['sparse_heads'] is None
    m.sparse_heads = True
    assert m.plan_for('bf16').meta['sparse_heads'] is not None
    assert m.plan_for('fp8').meta['sparse_heads'] is None and m.plan_for('fp32').meta['sparse_heads'] is None
    assert 'sparse_heads' not in m.hparams  # a run-time switch, not a constructor argument of the reference


def test_postprocess_routes_gated_heads(monkeypatch):
    """Control flow of CPN.postprocess with score-gated heads, with the GPU ops replaced by recording stubs: below
    ops.SPARSE_HEADS_MAX_DENSITY the gathered kernel + gathered decode, above it the two dense convs + plain decode."""
    from celldetection_amd import ops
    m = cda.models.CpnResNet18FPN(3)
    n, h, w = 2, 8, 10
    scores = torch.rand(n, 1, h, w)
    calls = []

    def compact(select_map, thresh, extra_flag=None):
        idx = (select_map.reshape(-1) > thresh).nonzero().squeeze(1).to(torch.int32)
        b = idx // (h * w)
        return idx, [int((b == i).sum()) for i in range(n)], 0

    def sparse_heads(op_a, op_b, ptr_, cs, grid, indices, weights, bias):
        calls.append(('sparse', ptr_, cs, tuple(grid), int(indices.numel())))
        return torch.zeros(indices.numel(), 2), torch.zeros(indices.numel(), 20)

    def dense_head(op, ptr_, cs, grid, weights, bias):
        calls.append(('dense', op, ptr_, cs, tuple(grid)))
        return torch.zeros(n, 2 if op == 'A' else 20, h, w)

    def decode(indices, scores_, locations, fourier, refinement, *, size, order, samples, iterations, offsets, num_buckets,
               gathered):
        calls.append(('decode', gathered, tuple(locations.shape), tuple(fourier.shape)))
        P = int(indices.numel())
        z = lambda *s: torch.zeros((P,) + s)
        return dict(contours=z(samples, 2), contour_proposals=z(samples, 2), boxes=z(4), scores=z(), locations=z(2),
                    fourier=z(order, 4), b=(indices // (h * w)).to(torch.int32))

    monkeypatch.setattr(ops, 'compact_scores', compact)
    monkeypatch.setattr(ops, 'sparse_heads', sparse_heads)
    monkeypatch.setattr(ops, 'dense_head', dense_head)
    monkeypatch.setattr(ops, 'decode_proposals', decode)
    ctx = dict(op_a='A', op_b='B', features_ptr=1234, channel_stride=256, grid=(n, h, w), weights=None, bias=None)
    for thresh, route in ((.9, 'sparse'), (.1, 'dense')):
        calls.clear()
        m.score_thresh = thresh
        out = m.postprocess(scores, None, None, None, (32, 40), nms=False, sparse=ctx)
        P = int((scores > thresh).sum())
        assert sum(len(v) for v in out['scores']) == P
        if route == 'sparse':
            assert calls[0] == ('sparse', 1234, 256, (n, h, w), P) and calls[1] == ('decode', True, (P, 2), (P, 20))
        else:
            assert [c[:2] for c in calls[:2]] == [('dense', 'A'), ('dense', 'B')]
            assert calls[2] == ('decode', False, (n, 2, h, w), (n, 20, h, w))
    with pytest.raises(ValueError):
        m.postprocess(scores, None, None, None, (32, 40), nms=False)  # maps missing and no gated-head context
