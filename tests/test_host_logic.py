"""CPU tests of the host-side mirror: state-dict compatibility, tiling tables, weight packing layout, C-ABI surface,
loud failure without a GPU.  No compute call goes through the HIP library here."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import celldetection_amd as cda
from celldetection_amd import _lib, graph
from model_specs import ALL_SPECS, ref_template_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')


@pytest.mark.parametrize('name', list(ALL_SPECS))
def test_state_dict_keys_match_reference(name):
    spec = ALL_SPECS[name]
    model = getattr(cda.models, spec['cls'])(**spec['kwargs'])
    ref = ref_template_state_dict(name)
    mine = model.state_dict()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
        assert mine[k].dtype == ref[k].dtype, k


def test_full_size_models_param_counts_and_flops():
    # parameters and per-tile conv GFLOP of the BASELINE configs
    for cls, params, size, gflop in (('CpnU22', 31.57, 256, 201.67), ('CpnResNeXt101UNet', 224.64, 512, 2392.83),
                                     ('CpnResNet18FPN', 27.24, 512, 2118.5), ('CpnResNet50FPN', 40.31, 512, 2145.3)):
        plan = graph.build_plan(cls[3:], 3)
        n = sum(int(np.prod(s)) for _, s, kind in plan.entries if kind == 'param') / 1e6
        assert abs(n - params) < 0.01, (cls, n)
        assert abs(graph.reference_flops(plan, size, size) / 1e9 - gflop) < 0.1 * (1 if 'FPN' not in cls else 1), cls


def test_tiling_tables_match_reference():
    t = np.load(os.path.join(G, 'tiling.npz'))
    for tag in 'abcdef':
        slices, overlaps, shape = cda.get_tiling_slices(tuple(int(i) for i in t[f'{tag}_size']),
                                                        tuple(int(i) for i in t[f'{tag}_crop']),
                                                        tuple(int(i) for i in t[f'{tag}_stride']), return_overlaps=True)
        sl = np.array([[[s.start, s.stop] for s in item] for item in slices])
        np.testing.assert_array_equal(sl, t[f'{tag}_slices'])
        np.testing.assert_array_equal(np.array([[list(o) for o in item] for item in overlaps]), t[f'{tag}_overlaps'])
        np.testing.assert_array_equal(np.array(shape), t[f'{tag}_shape'])


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'cpn_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(cpn_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    lib = _lib.load()  # resolves every symbol (raises otherwise); no GPU needed
    assert lib.cpn_abi_version() == _lib.ABI_VERSION
    assert lib.cpn_nms_workspace_bytes(1000, 1000, 1) > 1000 * 16 * 8


def test_clock_probe_lives_in_the_measurement_library_only():
    """include/cpn_hip.h cpn_debug_clock_probe: the product library carries no probe (the call fails with a message, without touching
    the device); libcpn_hip_clock.so is built next to it and exports the same ABI."""
    import ctypes
    lib = _lib.load()
    buf = (ctypes.c_uint64 * 15)()
    assert lib.cpn_debug_clock_probe(buf, 0) == 1
    assert b'without the clock probe' in lib.cpn_last_error()
    clock = ctypes.CDLL(os.path.join(ROOT, 'celldetection_amd', 'libcpn_hip_clock.so'), mode=ctypes.RTLD_LOCAL)
    for name in _lib.EXPORTED_SYMBOLS:
        assert hasattr(clock, name), name
    clock.cpn_abi_version.restype = ctypes.c_int
    assert clock.cpn_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    model = cda.models.CpnU22(3, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}})
    with pytest.raises(RuntimeError, match='GPU'):
        model(torch.rand(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        cda.ops.nms(torch.rand(4, 4), torch.rand(4), .5)
    with pytest.raises(NotImplementedError):
        model.train()


def test_fetchable_model_file_format(tmp_path):
    model = cda.models.CpnResNet18FPN(3, order=4, samples=16, backbone_kwargs={
        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}})
    model.score_thresh = .5  # updated attribute -> 'updated_kwargs' (celldetection/util/util.py:527-542)
    f = cda.save_fetchable_model(model, str(tmp_path / 'm'))
    raw = torch.load(f, weights_only=False)
    assert set(raw) >= {'cd.__version__', 'cd.models', 'state_dict'}
    assert raw['cd.models']['model'] == 'CpnResNet18FPN' and raw['cd.models']['updated_kwargs'] == {'score_thresh': .5}
    m2 = cda.load_model(f)
    assert m2.score_thresh == .5 and m2.samples == 16 and m2.order == 4
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    with pytest.raises(FileNotFoundError):
        cda.fetch_model('ginoro')  # offline: the hosted checkpoint is not available


def _emulate_packed_conv(op_desc, wblob, bblob, x0, x1):
    """Numerically emulates what the HIP kernel computes from the PACKED blobs (layout check on the CPU):
    weights [bundle][chunk][tap][cout_b][32], virtual concat [src0 padded | src1 padded], bias per output channel."""
    k, s, pad = op_desc.kh, op_desc.stride, op_desc.pad
    B, cin_b, cout_b = op_desc.bundles, op_desc.cin_b, op_desc.cout_b
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    items = (cin_b // 32) * k * k
    slabs = items + items % 2  # an odd item count is padded with one all-zero slab (two items per pipeline step)
    w = wblob[op_desc.weight_offset // 2: op_desc.weight_offset // 2 + B * slabs * cout_b * 32].float()
    w = w.reshape(B, slabs, cout_b, 32)
    if slabs != items:
        assert float(w[:, -1].abs().max()) == 0.
    w = w[:, :items].reshape(B, cin_b // 32, k * k, cout_b, 32).permute(0, 3, 1, 4, 2).reshape(B * cout_b, cin_b, k, k)
    b = bblob[op_desc.bias_offset: op_desc.bias_offset + B * cout_b]
    return F.conv2d(xin[:, :B * cin_b], w, b, s, pad, 1, B)


@pytest.mark.parametrize('cfg', [dict(cin=8, cout=24, k=3), dict(cin=64, cout=64, k=3, groups=32),
                                 dict(cin=256, cout=256, k=3, groups=32, stride=2),
                                 dict(cin=128, cout=128, k=3, groups=2), dict(cin=40, cout=48, k=1, cin1=20),
                                 dict(cin=96, cout=96, k=3, groups=4)])
def test_weight_packing_layout(cfg):
    g = torch.Generator().manual_seed(0)
    cin, cout, k = cfg['cin'], cfg['cout'], cfg['k']
    groups, stride, cin1 = cfg.get('groups', 1), cfg.get('stride', 1), cfg.get('cin1', 0)
    P = graph.Plan()
    s0 = P.tensor(cin, 1)
    s1 = P.tensor(cin1, 1) if cin1 else None
    P.conv(s0, cout, k, w='c.', bn='b.', bias=True, stride=stride, groups=groups, src1=s1)
    sd = {}
    for key, shape, kind in P.entries:
        sd[key] = (torch.rand(shape, generator=g) + .5) if key.endswith(('running_var', 'b.weight')) else \
            torch.randn(shape, generator=g) * .3
    tens, ops, wblob, bblob = graph.pack(P, sd, 'cpu')
    p32 = lambda c: (c + 31) // 32 * 32
    x0 = torch.zeros(1, p32(cin), 12, 12)
    x0[:, :cin] = torch.randn(1, cin, 12, 12, generator=g)
    x1 = None
    if cin1:
        x1 = torch.zeros(1, p32(cin1), 12, 12)
        x1[:, :cin1] = torch.randn(1, cin1, 12, 12, generator=g)
    got = _emulate_packed_conv(ops[0], wblob, bblob, x0, x1)[:, :cout]
    wf, bf = graph._fold(sd, P.ops[0])
    xin = x0[:, :cin] if not cin1 else torch.cat((x0[:, :cin], x1[:, :cin1]), 1)
    ref = F.conv2d(xin, wf.float(), bf.float(), stride, k // 2, 1, groups)
    # packed weights are bf16: compare against the bf16-rounded folded weights
    ref_bf = F.conv2d(xin, wf.float().to(torch.bfloat16).float(), bf.float(), stride, k // 2, 1, groups)
    assert torch.allclose(got, ref_bf, atol=1e-4, rtol=1e-4)
    assert (got - ref).abs().max() < 0.1


def test_pack_unpack_empty_detections():
    from celldetection_amd.inference import pack_detections, unpack_detections
    d = dict(contours=torch.zeros(0, 32, 2), boxes=torch.zeros(0, 4), scores=torch.zeros(0),
             classes=torch.zeros(0, dtype=torch.int64), locations=torch.zeros(0, 2), fourier=torch.zeros(0, 5, 4),
             contour_proposals=torch.zeros(0, 32, 2))
    buf = pack_detections(d)
    assert tuple(buf.shape) == (0, 32 * 2 + 4 + 1 + 1 + 2 + 20 + 64)
    out = unpack_detections(buf, 32, 5)
    assert all(tuple(out[k].shape) == tuple(d[k].shape) for k in d)


@pytest.mark.parametrize('name', list(ALL_SPECS))
def test_fp8_packing_passes_plan_validation(name):
    """fp8 weight/bias/multiplier blobs of every model: sizes and offsets satisfy the native plan's validation (host-side
    object only, no kernel runs), weight codes are finite e4m3, multipliers are positive."""
    from ctypes import c_void_p
    spec = ALL_SPECS[name]
    model = getattr(cda.models, spec['cls'])(**spec['kwargs'])
    plan = model.plan_for('fp8')
    if spec['kwargs'].get('refinement_interpolation') == 'bicubic' and any(o['op'] == 'bilinear' for o in plan.ops):
        with pytest.raises(NotImplementedError, match='bicubic'):  # bf16 / fp32 plans only (the result leaves the e4m3 range)
            graph.pack(plan, model.state_dict(), 'cpu', precision='fp8', act_scales=[0.01] * len(plan.tensors))
        return
    tens, ops, wblob, bblob, mblob, op_scales = graph.pack(plan, model.state_dict(), 'cpu', precision='fp8',
                                                           act_scales=[0.01 + 0.001 * i for i in range(len(plan.tensors))])
    assert wblob.dtype == torch.uint8 and all(t.channels % 64 == 0 for t in tens)
    assert float(mblob.min()) > 0
    lib = _lib.load()
    h = c_void_p()
    rc = lib.cpn_plan_create(h, tens, len(tens), ops, len(ops), _lib.ptr(wblob), wblob.numel(), _lib.ptr(bblob),
                             bblob.numel(), _lib.PRECISION_FP8)
    assert rc == 0, lib.cpn_last_error()
    assert lib.cpn_plan_workspace_bytes(h, 2, 64, 64) > 0
    lib.cpn_plan_destroy(h)
    for d, op in zip(ops, plan.ops):
        if d.op != _lib.OP_CONV:
            continue
        items = (d.cin_b // 64) * d.kh * d.kw
        codes = wblob[d.weight_offset:d.weight_offset + d.bundles * (items + items % 2) * d.cout_b * 64]
        assert not bool(((codes & 0x7f) == 0x7f).any()), op['w']  # no e4m3 NaN codes
        if items % 2:  # the padding slab of the last item is all zero
            assert int(codes.reshape(d.bundles, items + 1, -1)[:, -1].max()) == 0


def test_backbone_kwargs_whitelist_and_model2dict():
    """Constructor options the HIP graph does not model must fail loudly unless they carry the reference's default
    (a checkpoint trained with inputs_mean=0.5 would otherwise load and silently compute something else)."""
    u8 = {'backbone_kwargs': {'base_channels': 8}}
    cda.models.CpnU22(3, backbone_kwargs=dict(u8, inputs_mean=0., inputs_std=1., pretrained=False))
    for bad in (dict(u8, inputs_mean=.5), dict(u8, inputs_std=(.2, .2, .2)), dict(u8, interpolate='bilinear'),
                {'backbone_kwargs': {'base_channels': 8, 'block_cls': 'ResBlock'}}, dict(u8, made_up_option=1)):
        with pytest.raises(NotImplementedError):
            cda.models.CpnU22(3, backbone_kwargs=bad)
    with pytest.raises(NotImplementedError):
        cda.models.CpnResNet18FPN(3, backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8, 'fused_initial': True}})
    m = cda.models.CpnU22(3, backbone_kwargs=u8)
    m.score_thresh, m.samples = .5, 64
    d = cda.model2dict(m)
    assert d['model'] == 'CpnU22' and d['updated_kwargs'] == {'score_thresh': .5, 'samples': 64}
    assert d['kwargs']['backbone_kwargs'] == u8 and d['kwargs']['score_thresh'] == .9
    m2 = cda.dict2model(d)
    assert m2.score_thresh == .5 and m2.samples == 64 and list(m2.state_dict()) == list(m.state_dict())


def test_plans_per_precision():
    """bf16 plans fuse the bilinear resize into the refinement head's loader; fp8 / fp32 plans keep the separate op."""
    m = cda.models.CpnResNet18FPN(3, backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}})
    kinds = lambda p: [o['op'] for o in p.ops]
    assert 'bilinear' not in kinds(m.plan_for('bf16')) and 'bilinear' in kinds(m.plan_for('fp8'))
    assert any(o.get('up0') == 'bilinear' for o in m.plan_for('bf16').ops if o['op'] == 'conv')
    assert not any(o.get('fuse') for o in m.plan_for('fp32').ops if o['op'] == 'conv')
    u = cda.models.CpnU22(3, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}})
    assert 'bilinear' not in kinds(u.plan_for('fp8'))  # level 0 of a U22 has the input size by construction


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 5, 7, 6, 9), (1, 3, 4, 1, 1), (1, 8, 8, 16, 5)])
def test_subpixel_decomposition_is_exact(n, cin, cout, h, w):
    """3x3 conv over a x2 nearest-upsampled map == four 2x2 convs on the low-resolution map (celldetection_amd/subpixel.py):
    the algebra behind the planned decoder FLOP saving, incl. the zero padding at all four borders."""
    import torch.nn.functional as F
    from celldetection_amd import subpixel
    g = torch.Generator().manual_seed(n * 100 + h)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), wt, padding=1)
    got = subpixel.upsampled_conv_by_phases(x, wt)
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref, rtol=1e-12, atol=1e-12)
    assert subpixel.collapse_upsampled_taps(wt).shape == (2, 2, cout, cin, 2, 2)


def test_subpixel_triple_packing_matches_the_kernel_indexing():
    """CPU emulation of what conv_igemm does with the packed PHASE / LATERAL ops of a sub-pixel triple (bundle g = phase
    (g >> 1, g & 1) with padding (1 - py, 1 - px), K items chunk-major / tap-minor, residual read pixel-shuffled) against
    the reference's statement of the layer: conv3x3(cat(lateral, nearest_x2(top))) (models/unet.py:213-224)."""
    import torch.nn.functional as F
    from celldetection_amd import _lib, graph
    g = torch.Generator().manual_seed(0)
    n, h, w, c0, c1, cout = 1, 8, 12, 8, 40, 24
    P = graph.Plan()
    lat, top = P.tensor(c0, 1), P.tensor(c1, 2)
    kw = dict(w='c.', bn='b.', bias=True)
    x = P.conv(lat, cout, 3, act='relu', src1=top, up1=True, sub='head', **kw)
    ph = P.conv(top, cout, 2, pad=1, sub=('phase', c0), **kw)
    P.conv(lat, cout, 3, act='relu', res=ph, res_up='shuffle', sub=('lateral', c0), dst=x, **kw)
    assert len(P.entries) == 7  # one conv + one BN: the member ops add no state-dict entries
    sd = {}
    for key, shape, kind in P.entries:
        sd[key] = torch.zeros((), dtype=torch.long) if kind == 'long' else (
            torch.rand(shape, generator=g) + .5 if key.endswith('running_var') else torch.randn(shape, generator=g) * .3)
    tens, ops, wblob, bblob = graph.pack(P, sd, 'cpu')
    assert (ops[1].kh, ops[1].pad, ops[1].bundles, ops[1].bias_offset, ops[1].subpixel) == (2, 1, 4, -1, _lib.SUBPIXEL_PHASE)
    assert (ops[2].res_up, ops[2].res, ops[2].dst, ops[2].subpixel) == (2, ph, x, _lib.SUBPIXEL_LATERAL)
    assert tens[ph].channels == 4 * 32 and ops[0].dst == x and ops[0].subpixel == _lib.SUBPIXEL_HEAD
    xl, xt = torch.randn(n, c0, h, w, generator=g), torch.randn(n, c1, h // 2, w // 2, generator=g)

    def slab(op, k):  # [bundle][chunk][tap][cout_b][32] float
        items = (op.cin_b // 32) * k * k
        cnt = op.bundles * (items + (items & 1)) * op.cout_b * 32
        wv = wblob[op.weight_offset // 2: op.weight_offset // 2 + cnt].float()
        return wv.reshape(op.bundles, items + (items & 1), op.cout_b, 32)[:, :items].reshape(op.bundles, op.cin_b // 32, k * k, op.cout_b, 32)

    # PHASE: out[g][co][Y][X] = sum_{chunk, (ky, kx), c} W[g][chunk][ky*2+kx][co][c] * top[chunk*32+c][Y+ky-(1-py)][X+kx-(1-px)]
    wp = slab(ops[1], 2)
    xt_p = torch.zeros(n, ops[1].cin_b, h // 2, w // 2)
    xt_p[:, :c1] = xt
    part = torch.zeros(n, 4, ops[1].cout_b, h // 2, w // 2)
    for gph in range(4):
        py, px = gph >> 1, gph & 1
        xp = F.pad(xt_p, (1 - px, px, 1 - py, py))  # taps at offsets -(1-p), -(1-p)+1
        wk = wp[gph].permute(2, 0, 3, 1).reshape(ops[1].cout_b, ops[1].cin_b, 2, 2)  # [co][chunk*32+c][ky][kx]
        part[:, gph] = F.conv2d(xp, wk)
    # LATERAL: 3x3 on the lateral + bias + part(y >> 1, x >> 1, phase (y & 1, x & 1)) -> relu
    wl = slab(ops[2], 3)[0].permute(2, 0, 3, 1).reshape(ops[2].cout_b, ops[2].cin_b, 3, 3)
    xl_p = torch.zeros(n, ops[2].cin_b, h, w)
    xl_p[:, :c0] = xl
    out = F.conv2d(xl_p, wl, bblob[ops[2].bias_offset: ops[2].bias_offset + ops[2].cout_b], padding=1)
    shuf = part.reshape(n, 2, 2, ops[1].cout_b, h // 2, w // 2).permute(0, 3, 4, 1, 5, 2).reshape(n, ops[1].cout_b, h, w)
    out = F.relu(out + shuf)[:, :cout]
    wf, bf = graph._fold(sd, P.ops[0])
    ref = F.relu(F.conv2d(torch.cat((xl, F.interpolate(xt, scale_factor=2, mode='nearest')), 1), wf.float(), bf.float(), padding=1))
    assert (out - ref).abs().max().item() < 3e-2 * max(ref.abs().max().item(), 1.)  # bf16 weights vs fp32 weights
    assert ((out - ref).norm() / ref.norm()).item() < 5e-3


def test_subpixel_plans_keep_entries_and_flops_and_switch_per_size():
    import celldetection_amd as cda
    from celldetection_amd import _lib, graph
    from ctypes import c_void_p
    from celldetection_amd.synth import synth_state_dict
    m = cda.models.CpnU22(3, backbone_kwargs={'backbone_kwargs': {'base_channels': 32}})
    assert m.subpixel and sum(1 for op in m._plan.ops if op.get('sub') == 'head') == 4
    plain = graph.build_plan(**m._plan_kwargs)
    assert plain.entries == m._plan.entries and all(op.get('sub') is None for op in plain.ops)
    assert graph.reference_flops(plain, 256, 256) == graph.reference_flops(m._plan, 256, 256)
    assert all(op.get('sub') is None for op in m.plan_for('fp32').ops)
    # fp8 plans (round 5): the triples, never the scattered single-op form of a bridge level; same entries
    f8 = m.plan_for('fp8')
    assert sum(1 for op in f8.ops if op.get('sub') == 'head') == 4 and f8.entries == plain.entries
    assert not any(isinstance(op.get('sub'), tuple) and op['sub'][0] == 'scatter' for op in f8.ops)
    r = cda.models.CpnResNeXt101UNet(3, backbone_kwargs={'backbone_kwargs': {'base_channel': 8}})
    assert any(isinstance(op.get('sub'), tuple) and op['sub'][0] == 'scatter' for op in r.plan_for('bf16').ops)
    assert not any(isinstance(op.get('sub'), tuple) and op['sub'][0] == 'scatter' for op in r.plan_for('fp8').ops)
    r.subpixel = False
    assert all(op.get('sub') in (None,) or 'bl' in str(op.get('sub')) for op in r.plan_for('fp8').ops)
    m.subpixel = False
    assert all(op.get('sub') is None for op in m.plan_for('bf16').ops)
    m.subpixel = True
    sd = synth_state_dict(m.state_dict(), seed=0)
    lib = _lib.load()
    fl = {}
    for name, plan in (('sub', m._plan), ('plain', plain)):
        tens, ops, wblob, bblob = graph.pack(plan, sd, 'cpu')
        hdl = c_void_p()
        _lib.check(lib.cpn_plan_create(hdl, tens, len(tens), ops, len(ops), c_void_p(wblob.data_ptr()), wblob.numel() * 2,
                                       c_void_p(bblob.data_ptr()), bblob.numel(), _lib.PRECISION_BF16), 'create')
        fl[name] = {hw: lib.cpn_plan_executed_flops(hdl, 1, *hw) for hw in ((64, 96), (75, 101))}
        assert lib.cpn_plan_workspace_bytes(hdl, 2, 64, 96) > 0 and lib.cpn_plan_workspace_bytes(hdl, 2, 75, 101) > 0
        lib.cpn_plan_destroy(hdl)
    assert fl['sub'][(64, 96)] < .96 * fl['plain'][(64, 96)]      # exact x2 everywhere: the decomposition runs
    assert fl['sub'][(75, 101)] == fl['plain'][(75, 101)]        # odd sizes: every level falls back to the head conv


def test_reference_written_checkpoint_loads_on_the_host():
    """The model-file format is the reference's (util/util.py:545-560): a file written by the reference's own
    save_fetchable_model (fixture generated by make_golden.py gen_checkpoint) builds the same class with the constructor
    kwargs overridden by the attributes that were changed after construction, and every tensor of its state dict."""
    import os
    import celldetection_amd as cda
    from model_specs import G
    path = os.path.join(G, 'ref_checkpoint_CpnU22.pt')
    raw = torch.load(path, weights_only=True)  # builtins + tensors only: nothing of the reference is pickled
    assert set(raw) == {'cd.__version__', 'cd.models', 'state_dict'}
    model = cda.load_model(path)
    assert type(model).__name__ == raw['cd.models']['model'] == 'CpnU22'
    assert raw['cd.models']['updated_kwargs'] == {'score_thresh': .85, 'samples': 24}
    assert (model.score_thresh, model.samples, model.nms_thresh) == (.85, 24, raw['cd.models']['kwargs']['nms_thresh'])
    sd = model.state_dict()
    assert list(sd) == list(raw['state_dict']) and all(torch.equal(sd[k], v) for k, v in raw['state_dict'].items())
    d = cda.util.model2dict(model)  # and back: the description this build writes for that model
    assert d['model'] == 'CpnU22' and d['kwargs']['backbone_kwargs'] == raw['cd.models']['kwargs']['backbone_kwargs']


def test_stem_fast_path_plan_and_packing():
    """ResNet-family bf16 plans carry the stem alternatives (generic pair alt = 1, fast pair alt = 2; csrc/stem.hip); the
    fast conv packs the SAME state-dict weights as [ky][cout][kx 0..7][c 0..3]; the executor takes the fast pair at sizes
    where the padded 4-channel layout fits into the input tensor and reports fewer executed FLOPs there."""
    import celldetection_amd as cda
    from ctypes import c_void_p
    from celldetection_amd import _lib, graph
    from celldetection_amd.synth import synth_state_dict
    m = cda.models.CpnResNet18FPN(3, backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}})
    assert [(o['op'], o.get('alt')) for o in m._plan.ops[:4]] == [('input', 1), ('input_stem', 2), ('conv', 1), ('stem7', 2)]
    generic = graph.build_plan(**m._plan_kwargs, subpixel=True, fuse_blocks=True, bilinear_phases=True)
    assert generic.entries == m._plan.entries
    assert all(not o.get('alt') for o in m.plan_for('fp32').ops)
    assert [(o['op'], o.get('alt')) for o in m.plan_for('fp8').ops[:4]] == [('input', 1), ('input_stem', 2), ('conv', 1), ('stem7', 2)]
    # fp8 plans: the stem stays bf16 (weights + input), only its output is e4m3 -> bf16 weight bytes inside the byte blob,
    # multiplier slots of ones, output scale from the dst tensor
    sd8 = synth_state_dict(m.state_dict(), seed=0)
    p8 = m.plan_for('fp8')
    t8, o8, w8, b8, m8, sc8 = graph.pack(p8, sd8, 'cpu', precision='fp8', act_scales=[.01] * len(p8.tensors))
    st8 = o8[3]
    assert (st8.op, st8.alt, st8.cout_b, w8.dtype) == (_lib.OP_STEM7, 2, 64, torch.uint8)
    wk8 = w8[st8.weight_offset: st8.weight_offset + 7 * 64 * 32 * 2].view(torch.bfloat16).float().reshape(7, 64, 8, 4)
    wf8, _ = graph._fold(sd8, p8.ops[2])
    want8 = torch.zeros(7, 64, 8, 4)
    want8[:, :8, :7, :3] = wf8.permute(2, 0, 3, 1).float().to(torch.bfloat16).float()
    assert torch.equal(wk8, want8) and abs(sc8[3][1] - 100.) < 1e-3
    assert not any(o.get('alt') for o in cda.models.CpnU22(3).plan_for('bf16').ops)  # (3x3 stride-1 first conv: no stem)
    sd = synth_state_dict(m.state_dict(), seed=0)
    tens, ops, wblob, bblob = graph.pack(m._plan, sd, 'cpu')
    st = ops[3]
    assert (st.op, st.alt, st.cout_b, ops[1].op, ops[1].in_channels) == (_lib.OP_STEM7, 2, 32, _lib.OP_INPUT_STEM, 3)
    wk = wblob[st.weight_offset // 2: st.weight_offset // 2 + 7 * 32 * 32].float().reshape(7, 32, 8, 4)
    wf, bf = graph._fold(sd, m._plan.ops[2])
    want = torch.zeros(7, 32, 8, 4)
    want[:, :8, :7, :3] = wf.permute(2, 0, 3, 1).float().to(torch.bfloat16).float()
    assert torch.equal(wk, want)
    assert torch.equal(bblob[st.bias_offset: st.bias_offset + 8], bf.float())
    lib = _lib.load()
    fl = {}
    for name, plan in (('fast', m._plan), ('generic', generic)):
        t_, o_, w_, b_ = graph.pack(plan, sd, 'cpu')
        hdl = c_void_p()
        _lib.check(lib.cpn_plan_create(hdl, t_, len(t_), o_, len(o_), c_void_p(w_.data_ptr()), w_.numel() * 2,
                                       c_void_p(b_.data_ptr()), b_.numel(), _lib.PRECISION_BF16), 'create')
        fl[name] = lib.cpn_plan_executed_flops(hdl, 1, 64, 96)
        assert lib.cpn_plan_workspace_bytes(hdl, 2, 64, 96) > 0
        lib.cpn_plan_destroy(hdl)
    stem_generic = 2. * 32 * 48 * 32 * 32 * 49
    assert abs((fl['generic'] - fl['fast']) - stem_generic * (1 - 7 / 49.)) < 1.


@pytest.mark.parametrize('k', [3, 7])
def test_bilinear_subpixel_decomposition_is_exact(k):
    """k x k conv over a x2 BILINEAR-resized map (cpn.py:277-278 + the ReadOut conv) == four k2 x k2 convs on the low-resolution
    map (celldetection_amd/subpixel.py) wherever the conv window stays inside the resized image; the frame differs (there the
    conv's zero padding applies, not the resize's edge clamp) and is exactly ``bilinear_frame(k)`` pixels wide."""
    import torch.nn.functional as F
    from celldetection_amd import subpixel as sp
    g = torch.Generator().manual_seed(k)
    x = torch.randn(2, 5, 13, 17, generator=g, dtype=torch.float64)
    w = torch.randn(6, 5, k, k, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False), w, padding=k // 2)
    got = sp.bilinear_conv_by_phases(x, w)
    f = sp.bilinear_frame(k)
    assert f == {3: 2, 7: 4}[k] and sp.collapse_bilinear_taps(w).shape == (2, 2, 6, 5, (k + 3) // 2, (k + 3) // 2)
    d = (got - ref).abs()
    assert d[:, :, f:-f, f:-f].max() < 1e-12
    assert d[:, :, f - 1:d.shape[2] - f + 1, f - 1:d.shape[3] - f + 1].max() > 1e-3  # one pixel further out it is NOT the same
    with pytest.raises(ValueError):
        sp.collapse_bilinear_taps(torch.zeros(1, 1, 5, 5))


def test_bilinear_triple_in_fpn_plans(monkeypatch):
    """FPN models: the refinement ReadOut head over the bilinear-resized level-0 map carries its decomposition (BL_HEAD,
    BL_PHASE, BL_FRAME); the native plan accepts it and executes fewer MACs where the resize is an exact x2."""
    import celldetection_amd as cda
    from celldetection_amd import _lib, graph
    from ctypes import c_void_p
    from celldetection_amd.synth import synth_state_dict
    m = cda.models.CpnResNet18FPN(3, backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}})
    subs = [op.get('sub') for op in m._plan.ops if op.get('sub') is not None]
    assert subs == ['blhead', ('blphase', 7), ('blframe', 7)]
    plain = graph.build_plan(**m._plan_kwargs, fuse_blocks=True, stem_fast=True)
    assert plain.entries == m._plan.entries and graph.reference_flops(plain, 64, 96) == graph.reference_flops(m._plan, 64, 96)
    assert all(op.get('sub') is None for op in m.plan_for('fp32').ops)
    # fp8 plans keep the resize as an op of its own: head and frame conv read its output, the phases the map in front of it
    p8 = m.plan_for('fp8')
    i8 = next(j for j, op in enumerate(p8.ops) if op.get('sub') == 'blhead')
    rz = next(op for op in p8.ops if op['op'] == 'bilinear')
    assert rz.get('ring_for_bl') and p8.ops[i8]['src0'] == p8.ops[i8 + 2]['src0'] == rz['dst'] and p8.ops[i8 + 1]['src0'] == rz['src0']
    assert [p8.ops[i8 + j].get('sub') for j in (1, 2)] == [('blphase', 7), ('blframe', 7)] and not p8.ops[i8]['up0']
    sd = synth_state_dict(m.state_dict(), seed=0)
    lib = _lib.load()
    fl = {}
    for name, plan in (('bl', m._plan), ('plain', plain)):
        tens, ops, wblob, bblob = graph.pack(plan, sd, 'cpu')
        if name == 'bl':
            i = next(j for j, o in enumerate(ops) if o.subpixel == _lib.SUBPIXEL_BL_HEAD)
            assert (ops[i + 1].subpixel, ops[i + 1].kh, ops[i + 1].pad, ops[i + 1].bundles, ops[i + 1].up0) == (_lib.SUBPIXEL_BL_PHASE, 5, 2, 4, 0)
            assert (ops[i + 2].subpixel, ops[i + 2].kh, ops[i + 2].up0, ops[i + 2].out_index) == (_lib.SUBPIXEL_BL_FRAME, 7, 2, ops[i].out_index)
            assert ops[i + 1].fuse_cout == ops[i].fuse_cout == 2
        hdl = c_void_p()
        _lib.check(lib.cpn_plan_create(hdl, tens, len(tens), ops, len(ops), c_void_p(wblob.data_ptr()), wblob.numel() * 2,
                                       c_void_p(bblob.data_ptr()), bblob.numel(), _lib.PRECISION_BF16), 'create')
        fl[name] = [lib.cpn_plan_executed_flops(hdl, 1, h, w) for h, w in ((512, 512), (75, 101), (64, 96))]
        lib.cpn_plan_destroy(hdl)
    assert fl['bl'][1] == fl['plain'][1]  # odd size: the resize is no exact x2 -> the conv over the resized map
    assert fl['bl'][2] == fl['plain'][2]  # 64 x 96: exact, but the frame (whole 8 x 32 tiles) is most of the image -> no gain
    # 512^2: the head's 49 taps become 4 x 25 taps on 256^2 + 15 % frame tiles of the 7 x 7 conv
    head = 2. * 512 * 512 * 32 * 32 * 49
    saved = fl['plain'][0] - fl['bl'][0]
    assert 0.3 * head < saved < 0.5 * head, (saved / head)
    # the fp8 plan takes the same decision (kernel A/B switch CPN_BLPHASE=0: the conv over the resized map everywhere)
    tens, ops, wblob, bblob, mblob, _ = graph.pack(p8, sd, 'cpu', precision='fp8', act_scales=[.02] * len(p8.tensors))
    f8 = {}
    for env in ('0', None):
        if env is not None:
            monkeypatch.setenv('CPN_BLPHASE', env)
        else:
            monkeypatch.delenv('CPN_BLPHASE')
        hdl = c_void_p()
        _lib.check(lib.cpn_plan_create(hdl, tens, len(tens), ops, len(ops), _lib.ptr(wblob), wblob.numel(), _lib.ptr(bblob),
                                       bblob.numel(), _lib.PRECISION_FP8), 'create')
        f8[env] = [lib.cpn_plan_executed_flops(hdl, 1, h, w) for h, w in ((512, 512), (75, 101))]
        lib.cpn_plan_destroy(hdl)
    # closed form of the frame launch at 512^2 (m = 4, 8 x 32 tiles): the tile rows above and below the box whole (2 x 16 tiles),
    # ONE wrap tile for each of the 62 rows that cross it -> 94 of 1024 tiles run the 7x7 conv; the phases run four 5x5 convs
    # on all 256 tiles of the 256^2 map
    tile = 2. * 256 * 32 * 32
    assert abs(saved - (1024 * 49 - 94 * 49 - 4 * 256 * 25) * tile) < 1e-3 * head
    head8 = 2. * 512 * 512 * 64 * 32 * 49  # (fp8 plans pad channels to 64)
    assert f8['0'][1] == f8[None][1] and 0.3 * head8 < f8['0'][0] - f8[None][0] < 0.5 * head8


def test_unet_family_constructors_follow_the_reference():
    """CpnSlimU22 / CpnWideU22 / CpnResUNet (models/cpn.py:811-927, unet.py:434-524): state-dict keys and shapes equal the
    reference's (recorded in the golden fixtures); SlimU22 / WideU22 pass ``base_channels`` themselves, so a caller's value is the
    reference's TypeError (tests/golden/reference_behaviours.json)."""
    import json
    import celldetection_amd as cda
    from model_specs import ALL_SPECS, G, ref_template_state_dict
    for name in ('CpnSlimU22', 'CpnWideU22', 'CpnResUNet', 'CpnResNet18UNet', 'CpnResNet34UNet', 'CpnResNet18FPN_lowres'):
        spec = ALL_SPECS[name]
        m = getattr(cda.models, spec['cls'])(**spec['kwargs'])
        tmpl = ref_template_state_dict(name)
        sd = m.state_dict()
        assert list(sd.keys()) == list(tmpl.keys()), name
        assert all(tuple(sd[k].shape) == tuple(tmpl[k].shape) for k in sd), name
    with open(os.path.join(G, 'reference_behaviours.json')) as f:
        rec = json.load(f)
    assert rec['slimu22_base_channels_kwarg']['type'] == 'TypeError'
    for cls in ('CpnSlimU22', 'CpnWideU22'):
        with pytest.raises(TypeError, match="multiple values for keyword argument 'base_channels'"):
            getattr(cda.models, cls)(3, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}})


def test_every_unet_backbone_packs_into_a_valid_native_plan():
    """Plan.hoist must never split a fused pair / bridge unit -- every UNet backbone's default bf16 plan (heads
    hoisted, sub-pixel decoders, fused bridge) is accepted by cpn_plan_create (which validates the op order; no GPU needed)."""
    from ctypes import c_void_p
    import celldetection_amd as cda
    from celldetection_amd import _lib, graph
    lib = _lib.load()
    tiny = {'backbone_kwargs': {'base_channel': 8}}
    for bb in ('ResNet18UNet', 'ResNet34UNet', 'ResNet50UNet', 'ResNeXt101UNet', 'WideResNet50UNet'):
        for hoist in (True, False):
            m = getattr(cda.models, 'Cpn' + bb)(3, backbone_kwargs=tiny)
            P = graph.build_plan(bb, 3, backbone_kwargs=tiny, subpixel=True, stem_fast=True, fuse_blocks=True,
                                 bilinear_phases=True, hoist_heads=hoist)
            tens, ops, w, b = graph.pack(P, m.state_dict(), 'cpu', 'bf16')
            h = c_void_p()
            rc = lib.cpn_plan_create(h, tens, len(tens), ops, len(ops), _lib.ptr(w), w.numel() * 2, _lib.ptr(b), b.numel(),
                                     _lib.PRECISION_BF16)
            assert rc == 0, (bb, hoist, lib.cpn_last_error())
            lib.cpn_plan_destroy(h)
            for i, o in enumerate(P.ops):  # a fused op sits right behind the two convs it restates
                if o['op'] in ('conv_pair', 'conv_bridge'):
                    assert o['first'] == i - 2 and P.ops[i - 1]['src0'] == P.ops[i - 2]['dst']
            if hoist:  # the heads did move in front of the bridge level
                names = [o.get('w') for o in P.ops]
                assert names.index('core.score_head.block.0.') < names.index('core.backbone.unet.layer_blocks.0.0.')
