"""GPU parity of the fused bridge level (CPN_OP_CONV_BRIDGE, csrc/conv_igemm.hip MODE_BR): TwoConvNormRelu over the x2
nearest-upsampled map of a UNet bridge level (reference: celldetection/models/unet.py:92-107,213-217; commons.py:120-149),
through the C ABI.

* against the two cpn_conv2d launches it replaces (the scattered phase conv + the 3x3 conv): same operands, same K order, same
  rounding points -> bit-identical;
* against a plain PyTorch fp32 reference of the reference's statement of the level (conv 3x3 over F.interpolate(x, scale 2,
  nearest), BN, ReLU, conv 3x3, BN, ReLU) on the same bf16-rounded input;
* inside a plan: the executor runs the fused op on the full-width ResNet-UNets wherever the output holds a 16 x 32 tile, the two
  convs elsewhere and with CPN_BRIDGE=0 -- identical head maps either way.
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


def _bridge_plan(cin, seed):
    from celldetection_amd import graph
    g = torch.Generator().manual_seed(seed)
    P = graph.Plan()
    x = P.tensor(cin, 2)
    y = graph._two_conv_norm_relu(P, x, 64, 'blk.', bias=False, up0=True, subpixel=True)
    assert P.ops[-1]['op'] == 'conv_bridge' and P.ops[-1]['dst'] == y
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


CASES = {
    'bridge_64_full_tiles': dict(n=2, h=32, w=64, cin=64),          # output 64 x 128: whole 16 x 32 tiles
    'bridge_64_flagship_rows': dict(n=4, h=128, w=128, cin=64, seed=3),
    'bridge_32_in': dict(n=2, h=16, w=32, cin=32, seed=5),            # one input chunk
    'bridge_partial_tiles': dict(n=3, h=20, w=26, cin=64, seed=7),    # output 40 x 52: ragged tile rows / columns
    'bridge_min_size': dict(n=1, h=8, w=16, cin=64, seed=9),          # output 16 x 32: a single tile
    'bridge_odd_channels': dict(n=1, h=24, w=40, cin=40, seed=11),    # 40 real input channels in a 64-channel tensor
}


@pytest.mark.parametrize('brf', ['0', '1'], ids=['one_workgroup_per_cu', 'two_workgroups_per_cu'])
@pytest.mark.parametrize('name', list(CASES))
def test_conv_bridge_kernel(dev, name, brf, monkeypatch):
    """Both tiles of the fused kernel: MODE_BR (default: the <16,64,2,2> tile) and MODE_BRF (CPN_BRF=1: 8-row tiles on 4 waves,
    flat pitch-34 halo tiles, two workgroups per CU -- measured neutral, kept as an opt-in)."""
    from celldetection_amd import _lib, graph
    monkeypatch.setenv('CPN_BRF', brf)
    cfg = dict(seed=0)
    cfg.update(CASES[name])
    n, h, w, cin = (cfg[k] for k in ('n', 'h', 'w', 'cin'))
    P, sd = _bridge_plan(cin, cfg['seed'])
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    g = torch.Generator().manual_seed(cfg['seed'] + 1)
    x = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16).float()
    cs = _pad32(cin)
    d0 = torch.zeros(n, h, w, cs, dtype=torch.bfloat16, device=dev)
    d0[..., :cin] = x.permute(0, 2, 3, 1).to(torch.bfloat16).to(dev)
    lib = _lib.load()
    H, W = 2 * h, 2 * w
    fused = torch.full((n, H, W, 64), float('nan'), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.cpn_conv_bridge(ops[2], _lib.ptr(d0), cs, None, 0, _lib.ptr(fused), 64, n, h, w, _lib.ptr(wblob),
                                   _lib.ptr(bblob), _lib.stream_ptr()), 'conv_bridge')
    mid = torch.full((n, H, W, 64), float('nan'), dtype=torch.bfloat16, device=dev)
    two = torch.full((n, H, W, 64), float('nan'), dtype=torch.bfloat16, device=dev)
    # the scattered phase conv takes the LOW-resolution size, the 3x3 conv the full-resolution one
    _lib.check(lib.cpn_conv2d(ops[0], _lib.ptr(d0), cs, None, 0, None, 0, _lib.ptr(mid), 64, n, h, w, _lib.ptr(wblob),
                              _lib.ptr(bblob), _lib.stream_ptr()), 'conv2d scatter')
    _lib.check(lib.cpn_conv2d(ops[1], _lib.ptr(mid), 64, None, 0, None, 0, _lib.ptr(two), 64, n, H, W, _lib.ptr(wblob),
                              _lib.ptr(bblob), _lib.stream_ptr()), 'conv2d 3x3')
    torch.cuda.synchronize()
    assert torch.isfinite(fused.float()).all()
    assert torch.equal(fused, two), f'{name}: fused bridge differs from the two launches: max abs ' \
                                    f'{(fused.float() - two.float()).abs().max().item():.3e}, ' \
                                    f'{(fused != two).float().mean().item():.2e} of the outputs'
    # the reference's statement of the level, fp32 on the bf16-rounded input (the sub-pixel form rounds the collapsed tap sums once:
    # the usual tolerance)
    w1, b1 = graph._fold(sd, P.ops[0])
    w2, b2 = graph._fold(sd, P.ops[1])
    r = F.relu(F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), w1.float(), b1.float(), 1, 1)).to(torch.bfloat16).float()
    ref = F.relu(F.conv2d(r, w2.float().to(torch.bfloat16).float(), b2.float(), 1, 1))
    got = fused.float().permute(0, 3, 1, 2).cpu()
    scale = max(ref.abs().max().item(), 1.)
    assert (got - ref).abs().max().item() < 5e-2 * scale, (got - ref).abs().max().item()


def test_conv_bridge_in_the_plan_and_switch(dev, monkeypatch):
    """Full-width CpnResNet50UNet (64-channel bridge): the executor runs the fused op by default, the two convs with
    CPN_BRIDGE=0 and below a 16 x 32 output -- same head maps bit for bit."""
    import celldetection_amd as cda
    from celldetection_amd.synth import synth_state_dict
    m = cda.models.CpnResNet50UNet(3)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
    m = m.to(dev)
    assert sum(o['op'] == 'conv_bridge' for o in m.plan_for('bf16').ops) == 1
    for size in ((96, 128), (75, 101)):  # (odd input: the bridge output is 2 * ceil(H / 2) = 76 x 102 -- ragged tiles inside a plan)
        x = torch.rand(2, 3, *size, generator=torch.Generator().manual_seed(1)).to(dev)
        maps = {}
        for mode in ('1', '0'):
            monkeypatch.setenv('CPN_BRIDGE', mode)
            maps[mode] = [t.clone() for t in m.core_forward(x)]
            prof = m.engine(dev).profile(x, m.core.order, True)
            ran = [p['gflop'] > 0 for p in prof if p['op'] == 'conv_bridge']
            assert ran == [mode == '1'], (size, mode, ran)
            two = [p['gflop'] > 0 for p in prof if p['op'] == 'conv' and 'layer_blocks.0.' in (p['name'] or '')]
            assert two == [mode == '0'] * 2, (size, mode, two)
        for a, b in zip(maps['1'], maps['0']):
            assert torch.equal(a, b), size
    monkeypatch.setenv('CPN_BRIDGE', '1')
    tiny = torch.rand(1, 3, 8, 24, generator=torch.Generator().manual_seed(2)).to(dev)   # output 8 x 24: below one tile
    prof = m.engine(dev).profile(tiny, m.core.order, True)
    assert [p['gflop'] > 0 for p in prof if p['op'] == 'conv_bridge'] == [False]
