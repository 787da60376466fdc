"""GPU parity of the fused bottleneck head (CPN_OP_CONV_PAIR, csrc/conv_pair.hip): conv1 1x1 + BN + ReLU -> grouped conv2 3x3
+ BN + ReLU of a ResNeXt block (reference: torchvision Bottleneck.forward as built by celldetection/models/resnet.py:88-116,
119-193), through the C ABI.

* against a plain PyTorch fp32 reference of the two convs on the same bf16-rounded operands (conv1's activated output rounded
  to bf16 like the tensor the unfused graph stores);
* against the two cpn_conv2d launches it replaces: same operands, same K order, same rounding points -> bit-identical;
* inside a plan: the executor picks the fused op where the feature map is 16 / 32 / 64 wide and the two convs elsewhere.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _pad32(c):
    return (c + 31) // 32 * 32


def _pair_plan(cin, cmid, groups, seed, stride=1):
    from celldetection_amd import graph
    g = torch.Generator().manual_seed(seed)
    P = graph.Plan()
    x = P.tensor(cin, 1)
    t = P.conv(x, cmid, 1, w='conv1.', bn='bn1.', act='relu')
    P.conv(t, cmid, 3, w='conv2.', bn='bn2.', groups=groups, stride=stride, act='relu')
    assert P.conv_pair()
    sd = {}
    for key, shape, kind in P.entries:
        if key.endswith('running_var'):
            sd[key] = torch.rand(shape, generator=g) + .5
        elif key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
        elif len(shape) == 4:
            sd[key] = torch.randn(shape, generator=g) / np.sqrt(np.prod(shape[1:]))
        else:
            sd[key] = torch.randn(shape, generator=g) * .5 + (1. if key.endswith('.weight') else 0.)
    return P, sd


def run_pair(dev, *, n, h, w, cin, cmid, groups=32, seed=0, stride=1):
    from celldetection_amd import _lib, graph
    P, sd = _pair_plan(cin, cmid, groups, seed, stride)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16).float()
    cs = _pad32(cin)
    d0 = torch.zeros(n, h, w, cs, dtype=torch.bfloat16, device=dev)
    d0[..., :cin] = x.permute(0, 2, 3, 1).to(torch.bfloat16).to(dev)
    lib = _lib.load()
    cm = _pad32(cmid)
    fused = torch.full((n, ho, wo, cm), float('nan'), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.cpn_conv_pair(ops[2], _lib.ptr(d0), cs, _lib.ptr(fused), cm, n, h, w, _lib.ptr(wblob), _lib.ptr(bblob),
                                 _lib.stream_ptr()), 'conv_pair')
    mid = torch.full((n, h, w, cm), float('nan'), dtype=torch.bfloat16, device=dev)
    two = torch.full((n, ho, wo, cm), float('nan'), dtype=torch.bfloat16, device=dev)
    for op, src, dst, ss in ((ops[0], d0, mid, cs), (ops[1], mid, two, cm)):
        _lib.check(lib.cpn_conv2d(op, _lib.ptr(src), ss, None, 0, None, 0, _lib.ptr(dst), cm, n, h, w, _lib.ptr(wblob),
                                  _lib.ptr(bblob), _lib.stream_ptr()), 'conv2d')
    torch.cuda.synchronize()
    w1, b1 = graph._fold(sd, P.ops[0])
    w2, b2 = graph._fold(sd, P.ops[1])
    r = F.relu(F.conv2d(x, w1.float().to(torch.bfloat16).float(), b1.float())).to(torch.bfloat16).float()
    ref = F.relu(F.conv2d(r, w2.float().to(torch.bfloat16).float(), b2.float(), stride, 1, 1, groups))
    nchw = lambda t: t[..., :cmid].permute(0, 3, 1, 2).float().cpu()
    return nchw(fused), nchw(two), ref


PAIR_CASES = {
    # (feature-map width, channels per group) of the ResNeXt stages at 512^2 / 256^2 tiles, plus ragged heights
    'w32_cpg32_layer3': dict(n=2, h=32, w=32, cin=256, cmid=1024, groups=32),
    'w32_cpg8_tail_rows': dict(n=1, h=19, w=32, cin=64, cmid=256, groups=32),
    'w32_cpg16': dict(n=1, h=8, w=32, cin=96, cmid=512, groups=32, seed=3),
    'w32_cpg64': dict(n=1, h=16, w=32, cin=128, cmid=256, groups=4, seed=4),
    'w64_cpg16_layer2': dict(n=2, h=64, w=64, cin=128, cmid=512, groups=32, seed=5),
    'w64_cpg8_short': dict(n=1, h=5, w=64, cin=32, cmid=256, groups=32, seed=6),
    'w64_cpg64': dict(n=1, h=24, w=64, cin=64, cmid=128, groups=2, seed=7),
    'w16_cpg64_layer4': dict(n=2, h=16, w=16, cin=256, cmid=2048, groups=32, seed=8),
    'w16_cpg32_tall': dict(n=1, h=37, w=16, cin=64, cmid=256, groups=8, seed=9),
    'w16_one_chunk': dict(n=3, h=16, w=16, cin=32, cmid=256, groups=32, seed=10),
    # generic 16 x 32 tiles with a real halo on all sides: any width > 32 that is not 64 (stage 1 at 512^2: 128 wide)
    'gen_w128_cpg8_layer1': dict(n=2, h=128, w=128, cin=256, cmid=256, groups=32, seed=11),
    'gen_w128_first_block': dict(n=1, h=48, w=128, cin=64, cmid=256, groups=32, seed=12),
    'gen_w63_ragged': dict(n=2, h=37, w=63, cin=32, cmid=128, groups=16, seed=13),
    'gen_w40_cpg16': dict(n=1, h=16, w=40, cin=96, cmid=512, groups=32, seed=14),
    'gen_w100_tall': dict(n=1, h=70, w=100, cin=64, cmid=128, groups=4, seed=15),
    # stride-2 conv2 (the first block of a stage): 16 x 32 input tiles -> 8 x 16 outputs
    's2_w128_layer2_first': dict(n=2, h=128, w=128, cin=256, cmid=512, groups=32, stride=2, seed=16),
    's2_w64_layer3_first': dict(n=1, h=64, w=64, cin=128, cmid=1024, groups=32, stride=2, seed=17),
    's2_w32_cpg64_layer4_first': dict(n=2, h=32, w=32, cin=64, cmid=256, groups=4, stride=2, seed=18),
    's2_odd_sizes': dict(n=1, h=37, w=63, cin=32, cmid=128, groups=16, stride=2, seed=19),
    's2_w50_cpg8': dict(n=3, h=20, w=50, cin=64, cmid=256, groups=32, stride=2, seed=20),
}


@pytest.mark.parametrize('name', sorted(PAIR_CASES))
def test_conv_pair(dev, name):
    fused, two, ref = run_pair(dev, **PAIR_CASES[name])
    assert torch.isfinite(fused).all()
    # the two launches it replaces: same operands, K order and rounding points
    assert torch.equal(fused, two), (name, (fused - two).abs().max().item())
    err = (fused - ref).abs().max().item()
    tol = 2e-2 * max(1., ref.abs().max().item())  # bf16 output rounding (2^-8 relative) + fp32 accumulation order
    rel = ((fused - ref).norm() / (ref.norm() + 1e-12)).item()
    print(f'{name}: max err {err:.3e} rel L2 {rel:.3e}')
    assert err <= tol and rel < 5e-3, (name, err, rel)


def test_conv_pair_unsupported_width(dev):
    from celldetection_amd import _lib, graph
    P, sd = _pair_plan(64, 256, 32, 0)
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    x = torch.zeros(1, 8, 24, 64, dtype=torch.bfloat16, device=dev)
    y = torch.zeros(1, 8, 24, 256, dtype=torch.bfloat16, device=dev)
    rc = _lib.load().cpn_conv_pair(ops[2], _lib.ptr(x), 64, _lib.ptr(y), 256, 1, 8, 24, _lib.ptr(wblob), _lib.ptr(bblob),
                                   _lib.stream_ptr())
    assert rc == _lib.E_UNSUPPORTED  # 24 pixels wide: neither a strip width nor wide enough for the generic tiles


@pytest.mark.parametrize('size,fused', [((256, 256), True), ((200, 200), True), ((88, 96), False), ((128, 512), True)])
def test_plan_picks_the_fused_pair_per_input_size(dev, size, fused, monkeypatch):
    """A ResNeXt encoder plan carries both statements of every stride-1 block head; the executor runs the fused op where the
    stage's feature map is 16 / 32 / 64 pixels wide -- head maps are identical to the plan without fused ops."""
    import celldetection_amd as cda
    from celldetection_amd.synth import synth_state_dict
    monkeypatch.setenv('CPN_PAIR', '2')  # wherever the kernel applies (by default: only where the launch fills the chip)
    m = cda.models.CpnResNeXt50FPN(3, nms_thresh=.5, score_thresh=.5)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=1))
    m = m.to(dev)
    x = torch.rand(1, 3, *size, generator=torch.Generator().manual_seed(0)).to(dev)
    eng = m.engine(dev)
    n_pair = sum(o['op'] == 'conv_pair' for o in eng.plan.ops)
    assert n_pair == 3 + 4 + 6 + 3  # every block of the four stages (the three stride-2 ones included)
    prof = eng.profile(x, m.core.order, True)
    ran = [p for p in prof if p['op'] == 'conv_pair' and p['gflop'] > 0]
    assert bool(ran) == fused, [(p['name'], p['gflop']) for p in prof if p['op'] == 'conv_pair']
    got = [t.clone() for t in m.core_forward(x)]
    monkeypatch.setenv('CPN_PAIR', '0')
    m2 = cda.models.CpnResNeXt50FPN(3, nms_thresh=.5, score_thresh=.5)
    m2.load_state_dict(m.state_dict())
    m2 = m2.to(dev)
    exp = m2.core_forward(x)
    assert not any(p['op'] == 'conv_pair' and p['gflop'] > 0 for p in m2.engine(dev).profile(x, m2.core.order, True))
    for a, b in zip(got, exp):
        assert torch.equal(a, b)


def test_fused_pairs_run_by_default_where_they_fill_the_chip(dev):
    """Batch 16 of 256^2 tiles through a ResNeXt101 (32x8d) encoder: stage 1 (64 pixels wide, 256 channels) gives 16 x 8 x 2 =
    256 workgroups -> fused by default, like the stride-2 block heads of stages 2 / 3; the stride-1 blocks of the deeper stages
    (128 / 64 workgroups) keep the two convs.  Same head maps either way."""
    import os
    import celldetection_amd as cda
    from celldetection_amd.synth import synth_state_dict
    assert os.environ.get('CPN_PAIR') is None
    m = cda.models.CpnResNeXt101FPN(3, nms_thresh=.5, score_thresh=.5)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=2))
    m = m.to(dev)
    x = torch.rand(16, 3, 256, 256, generator=torch.Generator().manual_seed(0)).to(dev)
    prof = m.engine(dev).profile(x, m.core.order, True)
    ran = [p['name'] for p in prof if p['op'] == 'conv_pair' and p['gflop'] > 0]
    # stage 1 (three blocks, strips) + the stride-2 heads of stages 2 and 3 (generic tiles: 512 / 256 workgroups)
    assert len(ran) == 5 and sum('body.1.1.' in r for r in ran) == 3 and any('body.2.0.' in r for r in ran) and \
        any('body.3.0.' in r for r in ran), ran
    got = [t.clone() for t in m.core_forward(x)]
    os.environ['CPN_PAIR'] = '0'
    try:
        m2 = cda.models.CpnResNeXt101FPN(3, nms_thresh=.5, score_thresh=.5)
        m2.load_state_dict(m.state_dict())
        m2 = m2.to(dev)
        exp = m2.core_forward(x)
    finally:
        del os.environ['CPN_PAIR']
    for a, b in zip(got, exp):
        assert torch.equal(a, b)
