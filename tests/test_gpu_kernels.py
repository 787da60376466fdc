"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes).

* conv / pool / resize / input conversion: floating point -> compared with a plain PyTorch fp32 CPU reference of
  the same op on the same bf16-rounded operands (tolerance = bf16 output rounding + fp32 accumulation order).
* compaction / decode / refinement / NMS / border filter: compared bit-exactly with the CPU oracle and with the
  golden vectors generated from the reference.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _pad32(c):
    return (c + 31) // 32 * 32


def to_nhwc_bf16(x, cpad=None):
    """NCHW fp32 -> NHWC bf16 with zero-padded channels."""
    n, c, h, w = x.shape
    cpad = cpad or _pad32(c)
    out = torch.zeros(n, h, w, cpad, dtype=torch.bfloat16, device=x.device)
    out[..., :c] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    return out.contiguous()


def from_nhwc(y, c):
    return y[..., :c].permute(0, 3, 1, 2).float()


def run_conv(dev, *, n, h, w, cin, cout, k, stride=1, groups=1, bias=True, bn=True, act='relu', cin1=0, up0=False,
             up1=False, res=False, res_up=False, out_f32=False, act_scale=3., seed=0, fuse_cout=0, fuse_act='none',
             bilinear=False):
    """Builds a one-conv plan, runs cpn_conv2d, returns (got, ref) as fp32 NCHW CPU tensors."""
    from celldetection_amd import _lib, graph
    g = torch.Generator().manual_seed(seed)
    P = graph.Plan()
    hin, win = h, w
    s0 = P.tensor(cin, 2 if (up0 or bilinear) else 1)
    if bilinear:  # source stored at half resolution, read through the fused bilinear resize (MODE_BL)
        up0 = 'bilinear'
    s1 = P.tensor(cin1, 2 if up1 else 1) if cin1 else None
    r = P.tensor(cout, (2 if res_up else 1) * stride) if res else None
    P.conv(s0, cout, k, w='c.', bn='b.' if bn else None, bias=bias, stride=stride, groups=groups, act=act,
           act_scale=act_scale, src1=s1, up0=up0, up1=up1, res=r, res_up=res_up,
           out_index=_lib.OUT_SCORES if (out_f32 or fuse_cout) else None,
           fuse=dict(w='f.', cout=fuse_cout, act=fuse_act, act_scale=act_scale) if fuse_cout else None)
    out_f32 = out_f32 or bool(fuse_cout)
    sd = {}
    for key, shape, kind in P.entries:
        if key.endswith('running_var'):
            sd[key] = torch.rand(shape, generator=g) + .5
        elif key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
        elif len(shape) == 4:
            sd[key] = torch.randn(shape, generator=g) / np.sqrt(np.prod(shape[1:]))
        else:
            sd[key] = torch.randn(shape, generator=g) * .5 + (1. if key.endswith('b.weight') else 0.)
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    op = ops[0]

    def mk(c, hh, ww):
        return (torch.randn(n, c, hh, ww, generator=g)).to(torch.bfloat16).float()

    x0 = mk(cin, hin // 2 if up0 else hin, win // 2 if up0 else win)
    x1 = mk(cin1, hin // 2 if up1 else hin, win // 2 if up1 else win) if cin1 else None
    ho, wo = (hin + 2 * (k // 2) - k) // stride + 1, (win + 2 * (k // 2) - k) // stride + 1
    xr = mk(cout, ho // 2 if res_up else ho, wo // 2 if res_up else wo) if res else None
    d0 = to_nhwc_bf16(x0.to(dev))
    d1 = to_nhwc_bf16(x1.to(dev)) if cin1 else None
    dr = to_nhwc_bf16(xr.to(dev)) if res else None
    lib = _lib.load()
    if out_f32:
        dst = torch.full((n, fuse_cout or cout, ho, wo), float('nan'), dtype=torch.float32, device=dev)
        dstride = 0
    else:
        dst = torch.full((n, ho, wo, _pad32(cout)), float('nan'), dtype=torch.bfloat16, device=dev)
        dstride = _pad32(cout)
    _lib.check(lib.cpn_conv2d(op, _lib.ptr(d0), d0.shape[-1], _lib.ptr(d1), 0 if d1 is None else d1.shape[-1],
                              _lib.ptr(dr), 0 if dr is None else dr.shape[-1], _lib.ptr(dst), dstride, n, hin, win,
                              _lib.ptr(wblob), _lib.ptr(bblob), _lib.stream_ptr()), 'conv2d')
    torch.cuda.synchronize()
    got = dst.cpu() if out_f32 else from_nhwc(dst.cpu(), cout)
    # ---- reference: fp32 conv on the same bf16-rounded operands with the folded weights
    wf, bf = graph._fold(sd, P.ops[0])
    wf = wf.float().to(torch.bfloat16).float()
    if bilinear:  # blended values are rounded to bf16 before they enter the MFMA (like the materialised tensor was)
        xin = F.interpolate(x0, scale_factor=2, mode='bilinear', align_corners=False).to(torch.bfloat16).float()
    else:
        xin = F.interpolate(x0, scale_factor=2, mode='nearest') if up0 else x0
    if cin1:
        xin = torch.cat((xin, F.interpolate(x1, scale_factor=2, mode='nearest') if up1 else x1), 1)
    ref = F.conv2d(xin, wf, bf.float(), stride, k // 2, 1, groups)
    if res:
        ref = ref + (F.interpolate(xr, scale_factor=2, mode='nearest') if res_up else xr)
    if act == 'relu':
        ref = F.relu(ref)
    elif act == 'sigmoid':
        ref = torch.sigmoid(ref)
    elif act == 'tanh_scaled':
        ref = torch.tanh(ref) * act_scale
    if fuse_cout:  # fused ReadOut tail: bf16-rounded activations x bf16-rounded 1x1 weights, fp32 accumulate
        w2 = sd['f.weight'].to(torch.bfloat16).float()
        ref = F.conv2d(ref.to(torch.bfloat16).float(), w2, sd['f.bias'])
        if fuse_act == 'sigmoid':
            ref = torch.sigmoid(ref)
        elif fuse_act == 'tanh_scaled':
            ref = torch.tanh(ref) * act_scale
    return got, ref, out_f32


CONV_CASES = {
    '1x1_64_64': dict(n=2, h=32, w=32, cin=64, cout=64, k=1),
    '1x1_flat_narrow': dict(n=4, h=16, w=16, cin=128, cout=256, k=1, act='none'),
    '1x1_res': dict(n=1, h=32, w=64, cin=96, cout=128, k=1, res=True),
    '1x1_s2': dict(n=2, h=32, w=32, cin=64, cout=128, k=1, stride=2, act='none'),
    '3x3_64_64': dict(n=2, h=32, w=32, cin=64, cout=64, k=3),
    '3x3_odd_channels': dict(n=1, h=32, w=64, cin=8, cout=24, k=3),
    '3x3_256_big_tile': dict(n=1, h=64, w=64, cin=64, cout=256, k=3),
    '3x3_s2': dict(n=2, h=64, w=64, cin=32, cout=64, k=3, stride=2),
    '3x3_grouped_cpg8': dict(n=1, h=32, w=32, cin=256, cout=256, k=3, groups=32),
    '3x3_grouped_cpg64_s2': dict(n=1, h=32, w=32, cin=128, cout=128, k=3, groups=2, stride=2),
    '3x3_grouped_cpg1': dict(n=1, h=32, w=32, cin=32, cout=32, k=3, groups=32),
    '3x3_concat_up': dict(n=2, h=32, w=32, cin=64, cout=64, k=3, cin1=128, up1=True),
    '3x3_concat_up_pad': dict(n=1, h=32, w=32, cin=8, cout=16, k=3, cin1=16, up1=True),
    '3x3_bridge_up0': dict(n=1, h=64, w=64, cin=64, cout=64, k=3, up0=True, bias=False),
    '1x1_fpn_lateral': dict(n=2, h=32, w=32, cin=64, cout=256, k=1, res=True, res_up=True, bn=False, act='none'),
    '3x3_c64_th16_tile': dict(n=4, h=256, w=256, cin=64, cout=64, k=3),
    '7x7_c64_th16_tile': dict(n=4, h=256, w=256, cin=64, cout=64, k=7, seed=3),
    '7x7_head': dict(n=1, h=32, w=64, cin=64, cout=64, k=7),
    '7x7_head_256': dict(n=1, h=32, w=32, cin=256, cout=256, k=7),
    '7x7_stem_s2': dict(n=2, h=64, w=64, cin=3, cout=64, k=7, stride=2, bias=False),
    # sizes at which the flagship 8x256 tile is what launch_conv selects (see test_conv_register_weight_loop, too)
    '3x3_256_flagship_tile': dict(n=8, h=128, w=128, cin=64, cout=256, k=3),
    '3x3_256_flagship_concat_up': dict(n=8, h=128, w=128, cin=32, cout=256, k=3, cin1=64, up1=True, seed=7),
    'fused_head_256_flagship_tile': dict(n=8, h=128, w=128, cin=96, cout=256, k=7, fuse_cout=20, fuse_act='none', seed=9),
    # outputs exactly 16 pixels wide: MODE_N (a pixel fragment = two rows x 16 columns)
    '3x3_narrow16_flagship': dict(n=16, h=16, w=16, cin=64, cout=512, k=3, seed=21),
    '3x3_narrow16_rows24_res': dict(n=2, h=24, w=16, cin=32, cout=64, k=3, res=True, seed=22),
    '3x3_narrow16_grouped': dict(n=4, h=16, w=16, cin=256, cout=256, k=3, groups=8, seed=23),
    '3x3_narrow16_concat_up': dict(n=8, h=16, w=16, cin=64, cout=256, k=3, cin1=96, up1=True, seed=24),
    '7x7_narrow16_fused_head': dict(n=4, h=32, w=16, cin=64, cout=128, k=7, fuse_cout=20, fuse_act='none', seed=25),
    '5x5_narrow16_f32_out': dict(n=2, h=16, w=16, cin=32, cout=3, k=5, bn=False, act='tanh_scaled', out_f32=True, seed=26),
    '1x1_flagship_tile': dict(n=16, h=32, w=32, cin=96, cout=1024, k=1, seed=11),
    '1x1_flagship_tile_res': dict(n=16, h=32, w=32, cin=64, cout=1024, k=1, res=True, seed=13),
    'fused_head_256': dict(n=1, h=32, w=64, cin=64, cout=256, k=7, fuse_cout=20, fuse_act='none'),
    'fused_head_128_sigmoid': dict(n=2, h=32, w=32, cin=32, cout=128, k=3, fuse_cout=1, fuse_act='sigmoid'),
    'fused_head_64_tanh_th16': dict(n=4, h=256, w=256, cin=64, cout=64, k=7, fuse_cout=2, fuse_act='tanh_scaled', seed=5),
    'fused_head_64_tanh': dict(n=1, h=32, w=32, cin=64, cout=64, k=7, fuse_cout=2, fuse_act='tanh_scaled'),
    'fused_head_small': dict(n=1, h=32, w=64, cin=8, cout=8, k=7, fuse_cout=20, fuse_act='none'),
    '7x7_bilinear_src_256': dict(n=1, h=64, w=64, cin=64, cout=256, k=7, bilinear=True),
    '3x3_bilinear_src_32': dict(n=2, h=32, w=64, cin=16, cout=24, k=3, bilinear=True),
    'fused_head_64_bilinear': dict(n=1, h=64, w=96, cin=64, cout=64, k=7, fuse_cout=2, fuse_act='tanh_scaled', bilinear=True),
    'fused_head_256_bilinear': dict(n=1, h=32, w=64, cin=256, cout=256, k=7, fuse_cout=2, fuse_act='tanh_scaled', bilinear=True),
    'final_sigmoid': dict(n=2, h=32, w=32, cin=64, cout=1, k=1, bn=False, act='sigmoid', out_f32=True),
    'final_tanh': dict(n=1, h=64, w=32, cin=64, cout=2, k=1, bn=False, act='tanh_scaled', out_f32=True),
    'final_fourier': dict(n=1, h=32, w=32, cin=128, cout=20, k=1, bn=False, act='none', out_f32=True),
}


@pytest.mark.parametrize('name', list(CONV_CASES))
def test_conv(dev, name):
    got, ref, f32 = run_conv(dev, **CONV_CASES[name])
    assert got.shape == ref.shape
    assert torch.isfinite(got).all(), f'{name}: non-finite outputs ({(~torch.isfinite(got)).sum().item()})'
    err = (got - ref).abs()
    scale = ref.abs().max().item() + 1e-6
    tol = (2e-3 if f32 else 1e-2) * max(scale, 1.)  # bf16 output rounding: 2^-9 relative
    bad = (err > tol + (0 if f32 else 8e-3) * ref.abs()).sum().item()
    print(f'{name}: max abs err {err.max().item():.3e} (ref max {scale:.3e}), mean {err.mean().item():.3e}, bad {bad}')
    # fused heads round the intermediate activation to bf16: a value that sits on a rounding boundary may round the
    # other way than in the reference (different fp32 summation order) -> allow <= 1e-4 of the outputs up to 5e-2
    allowed = int(1e-4 * err.numel()) if name.startswith('fused_head') else 0
    assert bad <= allowed and err.max().item() < (5e-2 * max(scale, 1.) if allowed else float('inf')), \
        f'{name}: {bad} / {err.numel()} elements off; max abs err {err.max().item():.4e}, ref max {scale:.3e}'


FLAGSHIP_CASES = ['3x3_256_flagship_tile', '3x3_256_flagship_concat_up', 'fused_head_256_flagship_tile']


@pytest.mark.parametrize('name', FLAGSHIP_CASES)
def test_conv_register_weight_loop(dev, name, monkeypatch):
    """CPN_RW=1 selects MODE_S1R (weight fragments from L2 straight into registers, no weight tiles in LDS, two
    barriers per chunk) for dense KxK convs on the 8x256 tile.  Same K order and MFMA sequence as the LDS-weight loop:
    the outputs must be bit-identical, and both within the usual tolerance of the fp32 reference."""
    monkeypatch.setenv('CPN_RW', '0')
    lds, ref, f32 = run_conv(dev, **CONV_CASES[name])
    monkeypatch.setenv('CPN_RW', '1')
    rw, _, _ = run_conv(dev, **CONV_CASES[name])
    assert torch.equal(rw, lds), f'{name}: register-weight loop differs from the LDS-weight loop ' \
                                 f'(max abs {(rw - lds).abs().max().item():.3e})'
    scale = max(ref.abs().max().item(), 1.)
    assert (rw - ref).abs().max().item() < 5e-2 * scale


C64_CASES = {
    # the 64-channel tiles: <16,64,2,2> (3x3_c64_th16_tile / 7x7_c64_th16_tile: 4 x 256^2), <8,64,2,2> (small maps), a fused
    # ReadOut head with 64 hidden units, a concat source, partial tiles
    '3x3_c64_th16_tile': None, '7x7_c64_th16_tile': None, '3x3_64_64': None, '7x7_head': None,
    'c64_fused_head_th16': dict(n=4, h=256, w=256, cin=64, cout=64, k=7, fuse_cout=2, fuse_act='tanh_scaled', seed=11),
    'c64_concat_up_partial': dict(n=3, h=40, w=72, cin=32, cout=64, k=3, cin1=64, up1=True, seed=12),
    'c64_k5_three_chunks': dict(n=2, h=64, w=64, cin=96, cout=64, k=5, seed=13),
}


@pytest.mark.parametrize('name', list(C64_CASES))
def test_conv_register_weight_loop_64_channel_tiles(dev, name, monkeypatch):
    """CPN_RW bit 1: MODE_S1R on the 64-output-channel tiles (the refinement ReadOut head, commons.py:461-511 at 64 -> 64, and
    the bridge convs): weight fragments from L2 into registers, LDS holds only the halo -- half the LDS fragment reads.  Same K
    order and MFMA sequence as the LDS-weight loop: bit-identical outputs."""
    cfg = C64_CASES[name] or CONV_CASES[name]
    monkeypatch.setenv('CPN_RW', '0')
    lds, ref, f32 = run_conv(dev, **cfg)
    monkeypatch.setenv('CPN_RW', '2')
    rw, _, _ = run_conv(dev, **cfg)
    assert torch.equal(rw, lds), f'{name}: register-weight loop differs from the LDS-weight loop ' \
                                 f'(max abs {(rw - lds).abs().max().item():.3e})'
    scale = max(ref.abs().max().item(), 1.)
    assert (rw - ref).abs().max().item() < 5e-2 * scale


S1F_CASES = {
    # MODE_S1F (two 4-wave workgroups per CU, flat pitch-36 halo tile): CPN_S1F=2 forces it wherever the kernel applies
    's1f_3x3_256': dict(n=8, h=128, w=128, cin=64, cout=256, k=3),                 # the decoder's 3x3 class
    's1f_3x3_one_chunk': dict(n=2, h=64, w=64, cin=32, cout=128, k=3, seed=31),   # a single chunk: one halo buffer
    's1f_3x3_partial_tiles': dict(n=3, h=44, w=72, cin=96, cout=256, k=3, seed=32),   # ragged rows / columns, 3 chunks
    's1f_3x3_res': dict(n=2, h=64, w=96, cin=64, cout=128, k=3, res=True, seed=33),
    's1f_5x5_falls_back': dict(n=2, h=64, w=64, cin=64, cout=128, k=5, seed=34),   # two halo tiles + slabs > 80 KiB: the plain tiles run
    's1f_2x2ish_k3_cin2048': dict(n=1, h=32, w=64, cin=2048, cout=128, k=3, seed=39),   # 64 chunks through the two-buffer ring
    's1f_3x3_512_two_blocks': dict(n=4, h=64, w=64, cin=128, cout=512, k=3, seed=36),   # four cout blocks folded into x
    's1f_3x3_odd_tile_count': dict(n=3, h=24, w=40, cin=64, cout=384, k=3, seed=37),    # tiles not a multiple of 8: padded grid
    's1f_3x3_no_bias_none': dict(n=1, h=32, w=32, cin=64, cout=128, k=3, bias=False, bn=False, act='none', seed=38),
}


@pytest.mark.parametrize('name', list(S1F_CASES))
def test_conv_two_workgroups_per_cu_mode(dev, name, monkeypatch):
    """MODE_S1F against the flagship path (CPN_S1F=0): same operands, K order and MFMA sequence per output element ->
    bit-identical; and within the usual tolerance of the fp32 conv."""
    cfg = S1F_CASES[name]
    monkeypatch.setenv('CPN_S1F', '0')
    base, ref, _ = run_conv(dev, **cfg)
    monkeypatch.setenv('CPN_S1F', '2')
    got, _, _ = run_conv(dev, **cfg)
    assert torch.isfinite(got).all()
    assert torch.equal(got, base), f'{name}: MODE_S1F differs from the one-workgroup-per-CU tiles ' \
                                   f'(max abs {(got - base).abs().max().item():.3e}, {(got != base).float().mean().item():.2e} of the outputs)'
    scale = max(ref.abs().max().item(), 1.)
    assert (got - ref).abs().max().item() < 5e-2 * scale


S1Q_CASES = {
    # MODE_S1Q (four K items per pipeline step, flat pitch-40 halo tiles) on the 64-output-channel 5x5 / 7x7 convs
    's1q_7x7_64_64': dict(n=4, h=256, w=256, cin=64, cout=64, k=7, seed=41),                      # 98 items -> 100 (two read zeros)
    's1q_7x7_fused_head': dict(n=4, h=256, w=256, cin=64, cout=64, k=7, fuse_cout=2, fuse_act='tanh_scaled', seed=42),
    's1q_7x7_one_chunk': dict(n=8, h=128, w=160, cin=32, cout=64, k=7, seed=43),                   # 49 items -> 52 (three read zeros)
    's1q_5x5_three_chunks_ragged': dict(n=6, h=120, w=176, cin=96, cout=64, k=5, seed=44),         # 75 items, partial tiles
    's1q_7x7_res': dict(n=4, h=128, w=256, cin=64, cout=64, k=7, res=True, seed=45),
}


@pytest.mark.parametrize('name', list(S1Q_CASES))
def test_conv_four_items_per_step_mode(dev, name, monkeypatch):
    """MODE_S1Q against the two-items-per-step loop (CPN_S1Q=0): same K order and MFMA sequence per output element -> bit-identical;
    and within the usual tolerance of the fp32 conv."""
    cfg = S1Q_CASES[name]
    monkeypatch.setenv('CPN_S1Q', '0')
    base, ref, _ = run_conv(dev, **cfg)
    monkeypatch.setenv('CPN_S1Q', '1')
    got, _, _ = run_conv(dev, **cfg)
    assert torch.isfinite(got).all()
    assert torch.equal(got, base), f'{name}: MODE_S1Q differs from the two-item loop (max abs {(got - base).abs().max().item():.3e}, ' \
                                   f'{(got != base).float().mean().item():.2e} of the outputs)'
    scale = max(ref.abs().max().item(), 1.)
    assert (got - ref).abs().max().item() < 5e-2 * scale


SUBPIXEL_CASES = {
    'sp_64_128_64': dict(n=2, h=32, w=32, c0=64, c1=128, cout=64),
    'sp_padded_channels': dict(n=1, h=64, w=64, c0=8, c1=16, cout=16),
    'sp_partial_tiles': dict(n=2, h=40, w=48, c0=32, c1=64, cout=96, seed=3),
    'sp_flagship_tile': dict(n=8, h=128, w=128, c0=32, c1=64, cout=256, seed=5),
    'sp_two_cout_blocks': dict(n=4, h=64, w=64, c0=64, c1=64, cout=512, seed=7),
    'sp_no_bias': dict(n=1, h=32, w=64, c0=32, c1=32, cout=32, bias=False, seed=9),
    'sp_narrow_lowres': dict(n=16, h=32, w=32, c0=64, c1=128, cout=256, seed=11),  # phase conv on 16 x 16 maps: MODE_N
}


@pytest.mark.parametrize('name', list(SUBPIXEL_CASES))
def test_subpixel_decoder_conv(dev, name):
    """Sub-pixel triple of a UNet decoder conv over cat(lateral, nearest-x2-upsampled top-down map)
    (models/unet.py:213-224): PHASE (four 2x2 convs on the low-resolution map) + LATERAL (3x3 on the lateral, pixel-
    shuffled partial sums as residual) against (a) an fp32 emulation of exactly that arithmetic on the same bf16-rounded
    operands -- tap sums rounded to bf16 ONCE, partial sums rounded to bf16 -- and (b) the HEAD conv = the reference's
    statement of the layer (both approximate the fp32 conv; they differ by the rounding of the tap sums and partials)."""
    from celldetection_amd import _lib, graph
    from celldetection_amd.subpixel import collapse_upsampled_taps, phase_padding
    c = dict(bias=True, seed=0)
    c.update(SUBPIXEL_CASES[name])
    n, h, w, c0, c1, cout = (c[k] for k in ('n', 'h', 'w', 'c0', 'c1', 'cout'))
    g = torch.Generator().manual_seed(c['seed'])
    P = graph.Plan()
    lat, top = P.tensor(c0, 1), P.tensor(c1, 2)
    kw = dict(w='c.', bn='b.', bias=c['bias'])
    x = P.conv(lat, cout, 3, act='relu', src1=top, up1=True, sub='head', **kw)
    ph = P.conv(top, cout, 2, pad=1, sub=('phase', c0), **kw)
    P.conv(lat, cout, 3, act='relu', res=ph, res_up='shuffle', sub=('lateral', c0), dst=x, **kw)
    sd = {}
    for key, shape, kind in P.entries:
        if key.endswith('running_var'):
            sd[key] = torch.rand(shape, generator=g) + .5
        elif key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
        elif len(shape) == 4:
            sd[key] = torch.randn(shape, generator=g) / np.sqrt(np.prod(shape[1:]))
        else:
            sd[key] = torch.randn(shape, generator=g) * .5 + (1. if key.endswith('b.weight') else 0.)
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    assert [o.subpixel for o in ops] == [_lib.SUBPIXEL_HEAD, _lib.SUBPIXEL_PHASE, _lib.SUBPIXEL_LATERAL]
    assert tens[ph].channels == 4 * _pad32(cout)
    xl = torch.randn(n, c0, h, w, generator=g).to(torch.bfloat16).float()
    xt = torch.randn(n, c1, h // 2, w // 2, generator=g).to(torch.bfloat16).float()
    dl, dt = to_nhwc_bf16(xl.to(dev)), to_nhwc_bf16(xt.to(dev))
    cp = _pad32(cout)
    lib = _lib.load()
    nan = float('nan')
    head = torch.full((n, h, w, cp), nan, dtype=torch.bfloat16, device=dev)
    part = torch.full((n, h // 2, w // 2, 4 * cp), nan, dtype=torch.bfloat16, device=dev)
    pair = torch.full((n, h, w, cp), nan, dtype=torch.bfloat16, device=dev)
    args = (_lib.ptr(wblob), _lib.ptr(bblob), _lib.stream_ptr())
    _lib.check(lib.cpn_conv2d(ops[0], _lib.ptr(dl), dl.shape[-1], _lib.ptr(dt), dt.shape[-1], _lib.ptr(None), 0,
                              _lib.ptr(head), cp, n, h, w, *args), 'head')
    _lib.check(lib.cpn_conv2d(ops[1], _lib.ptr(dt), dt.shape[-1], _lib.ptr(None), 0, _lib.ptr(None), 0,
                              _lib.ptr(part), 4 * cp, n, h // 2, w // 2, *args), 'phase')
    _lib.check(lib.cpn_conv2d(ops[2], _lib.ptr(dl), dl.shape[-1], _lib.ptr(None), 0, _lib.ptr(part), 4 * cp,
                              _lib.ptr(pair), cp, n, h, w, *args), 'lateral')
    torch.cuda.synchronize()
    assert torch.isfinite(part.float()).all() and torch.isfinite(pair.float()).all()
    got, got_head = from_nhwc(pair.cpu(), cout), from_nhwc(head.cpu(), cout)
    wf, bf = graph._fold(sd, P.ops[0])
    wc = collapse_upsampled_taps(wf[:, c0:]).float().to(torch.bfloat16).float()  # [py, px, cout, c1, 2, 2]
    psum = torch.zeros(n, cout, h, w)
    for py in (0, 1):
        for px in (0, 1):
            pt, pl = phase_padding(py), phase_padding(px)
            psum[:, :, py::2, px::2] = F.conv2d(F.pad(xt, (pl, 1 - pl, pt, 1 - pt)), wc[py, px])
    got_part = part.cpu().float().reshape(n, h // 2, w // 2, 2, 2, cp)[..., :cout]  # [n, Y, X, py, px, c]
    got_part = got_part.permute(0, 5, 1, 3, 2, 4).reshape(n, cout, h, w)
    scale_p = max(psum.abs().max().item(), 1.)
    assert (got_part - psum).abs().max().item() < 1e-2 * scale_p, 'phase partial sums'
    ref = F.relu(F.conv2d(xl, wf[:, :c0].float().to(torch.bfloat16).float(), bf.float(), 1, 1)
                 + psum.to(torch.bfloat16).float())
    err = (got - ref).abs()
    scale = max(ref.abs().max().item(), 1.)
    # the kernel adds ITS bf16-rounded partial sum, which may sit one bf16 ulp (2^-8 relative) from the emulation's
    bad = (err > 1e-2 * scale + 8e-3 * ref.abs() + 2 ** -7 * psum.abs()).sum().item()
    print(f'{name}: vs emulation max abs err {err.max().item():.3e} (ref max {scale:.3e}); '
          f'vs head conv max {(got - got_head).abs().max().item():.3e}')
    assert bad == 0, f'{name}: {bad} / {err.numel()} elements off the emulated arithmetic'
    # (b) the decomposition against the reference's statement of the layer: same function, different bf16 roundings
    full = F.relu(F.conv2d(torch.cat((xl, F.interpolate(xt, scale_factor=2, mode='nearest')), 1), wf.float(), bf.float(), 1, 1))
    rel_pair = ((got - full).norm() / full.norm()).item()
    rel_head = ((got_head - full).norm() / full.norm()).item()
    print(f'{name}: rel L2 vs the fp32 conv: sub-pixel pair {rel_pair:.2e}, head conv {rel_head:.2e}')
    assert rel_pair < 1e-2 and rel_pair < 2.5 * rel_head + 1e-3


@pytest.mark.parametrize('n,h,w,cin,cout,bias', [(2, 32, 32, 64, 64, False), (1, 40, 72, 16, 24, True),
                                                  (4, 256, 256, 64, 64, False), (2, 64, 64, 32, 256, True)])
def test_subpixel_bridge_conv(dev, n, h, w, cin, cout, bias):
    """CPN_SUBPIXEL_SCATTER: the 3x3 conv of a GeneralizedUNet bridge level, whose only source is the x2-upsampled map
    (models/unet.py:100-107,213-217), as four 2x2 phase convs + bias + ReLU scattered to their output pixels; vs the fp32
    emulation of that arithmetic (tap sums rounded to bf16 once) and vs the fp32 conv over the upsampled map."""
    from celldetection_amd import _lib, graph
    from celldetection_amd.subpixel import collapse_upsampled_taps, phase_padding
    g = torch.Generator().manual_seed(n * 100 + cout)
    P = graph.Plan()
    top = P.tensor(cin, 2)
    dst = P.conv(top, cout, 2, w='c.', bn='b.', bias=bias, act='relu', pad=1, sub=('scatter', 0))
    assert P.tensors[dst]['down'] == 1 and [e[1] for e in P.entries if e[0] == 'c.weight'] == [(cout, cin, 3, 3)]
    sd = {}
    for key, shape, kind in P.entries:
        if key.endswith('running_var'):
            sd[key] = torch.rand(shape, generator=g) + .5
        elif key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
        elif len(shape) == 4:
            sd[key] = torch.randn(shape, generator=g) / np.sqrt(np.prod(shape[1:]))
        else:
            sd[key] = torch.randn(shape, generator=g) * .5 + (1. if key.endswith('b.weight') else 0.)
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    assert ops[0].subpixel == _lib.SUBPIXEL_SCATTER and ops[0].bundles == 4 and ops[0].kh == 2
    xt = torch.randn(n, cin, h // 2, w // 2, generator=g).to(torch.bfloat16).float()
    dt = to_nhwc_bf16(xt.to(dev))
    cp = _pad32(cout)
    out = torch.full((n, h, w, cp), float('nan'), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.load().cpn_conv2d(ops[0], _lib.ptr(dt), dt.shape[-1], _lib.ptr(None), 0, _lib.ptr(None), 0, _lib.ptr(out),
                                      cp, n, h // 2, w // 2, _lib.ptr(wblob), _lib.ptr(bblob), _lib.stream_ptr()), 'scatter')
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    got = from_nhwc(out.cpu(), cout)
    wf, bf = graph._fold(sd, P.ops[0])
    wc = collapse_upsampled_taps(wf).float().to(torch.bfloat16).float()
    ref = torch.zeros(n, cout, h, w)
    for py in (0, 1):
        for px in (0, 1):
            pt, pl = phase_padding(py), phase_padding(px)
            ref[:, :, py::2, px::2] = F.conv2d(F.pad(xt, (pl, 1 - pl, pt, 1 - pt)), wc[py, px], bf.float())
    ref = F.relu(ref)
    err = (got - ref).abs()
    scale = max(ref.abs().max().item(), 1.)
    bad = (err > 1e-2 * scale + 8e-3 * ref.abs()).sum().item()
    assert bad == 0, f'{bad} / {err.numel()} elements off; max abs err {err.max().item():.3e}'
    full = F.relu(F.conv2d(F.interpolate(xt, scale_factor=2, mode='nearest'), wf.float(), bf.float(), 1, 1))
    assert ((got - full).norm() / full.norm()).item() < 1e-2


@pytest.mark.parametrize('n,h,w,cin,cout,dtype', [(2, 64, 96, 3, 64, 0), (1, 75, 101, 3, 64, 0), (3, 33, 47, 1, 8, 1),
                                                   (2, 64, 64, 4, 32, 0), (16, 128, 128, 3, 64, 1), (1, 7, 9, 3, 64, 0)])
def test_stem_fast_path(dev, n, h, w, cin, cout, dtype):
    """csrc/stem.hip: ResNet stem conv 7x7 stride 2 pad 3 + BN + ReLU (models/resnet.py:274-284) from the padded 4-channel
    input layout, against an fp32 conv on the same bf16-rounded operands; the input conversion itself must be exact."""
    from celldetection_amd import _lib, graph
    g = torch.Generator().manual_seed(h * 1000 + w)
    P = graph.Plan()
    x = P.input(cin)
    P.conv(x, cout, 7, w='c.', bn='b.', stride=2, pad=3, act='relu')
    P.stem_fast_path(len(P.ops) - 1)
    assert [o['op'] for o in P.ops] == ['input', 'input_stem', 'conv', 'stem7'] and [o['alt'] for o in P.ops] == [1, 2, 1, 2]
    sd = {}
    for key, shape, kind in P.entries:
        if key.endswith('running_var'):
            sd[key] = torch.rand(shape, generator=g) + .5
        elif key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
        elif len(shape) == 4:
            sd[key] = torch.randn(shape, generator=g) / np.sqrt(np.prod(shape[1:]))
        else:
            sd[key] = torch.randn(shape, generator=g) * .5 + (1. if key.endswith('b.weight') else 0.)
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    lib = _lib.load()
    xin = torch.rand(n, cin, h, w, generator=g)
    src = xin.to(dev) if dtype == 0 else (xin * 255).to(torch.uint8).to(dev)
    xq = xin if dtype == 0 else (xin * 255).to(torch.uint8).float() / 255
    pad = torch.full((n, h + 6, w + 8, 4), 7., dtype=torch.bfloat16, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.cpn_convert_input_stem(_lib.ptr(src.contiguous()), dtype, _lib.ptr(pad), n, cin, h, w, _lib.ptr(flag),
                                          _lib.stream_ptr()), 'input_stem')
    torch.cuda.synchronize()
    exp = torch.zeros(n, h + 6, w + 8, 4)
    exp[:, 3:3 + h, 3:3 + w, :cin] = xq.to(torch.bfloat16).float().permute(0, 2, 3, 1)
    assert torch.equal(pad.cpu().float(), exp) and flag.item() == 0
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    cp = _pad32(cout)
    out = torch.full((n, ho, wo, cp), float('nan'), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.cpn_stem7(ops[3], _lib.ptr(pad), _lib.ptr(out), cp, n, h, w, _lib.ptr(wblob), _lib.ptr(bblob), 0.,
                             _lib.stream_ptr()), 'stem7')
    # the generic kernel on the 32-channel input, same weights: both approximate the same fp32 conv
    gen_in = to_nhwc_bf16(xq.to(dev))
    gen = torch.full((n, ho, wo, cp), float('nan'), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.cpn_conv2d(ops[2], _lib.ptr(gen_in), gen_in.shape[-1], _lib.ptr(None), 0, _lib.ptr(None), 0, _lib.ptr(gen),
                              cp, n, h, w, _lib.ptr(wblob), _lib.ptr(bblob), _lib.stream_ptr()), 'generic stem')
    torch.cuda.synchronize()
    got = from_nhwc(out.cpu(), cout)
    assert torch.isfinite(out.float()).all() and (cout == cp or out[..., cout:].abs().max().item() == 0)
    wf, bf = graph._fold(sd, P.ops[2])
    ref = F.relu(F.conv2d(xq.to(torch.bfloat16).float(), wf.float().to(torch.bfloat16).float(), bf.float(), 2, 3))
    err = (got - ref).abs()
    scale = max(ref.abs().max().item(), 1.)
    bad = (err > 1e-2 * scale + 8e-3 * ref.abs()).sum().item()
    assert bad == 0, f'{bad} / {err.numel()} elements off; max abs err {err.max().item():.3e}'
    assert (from_nhwc(gen.cpu(), cout) - got).abs().max().item() < 2e-2 * scale
    # fp8 plans: the same bf16 stem with e4m3 output codes of value / scale (channel stride 64)
    scale = float(ref.abs().max()) / 448. + 1e-12
    cp8 = (cout + 63) // 64 * 64
    tens8, ops8, wblob8, bblob8, _, _ = graph.pack(P, sd, dev, precision='fp8', act_scales=[1. / 448., scale])
    codes = torch.full((n, ho, wo, cp8), 0x7f, dtype=torch.uint8, device=dev)
    _lib.check(lib.cpn_stem7(ops8[3], _lib.ptr(pad), _lib.ptr(codes), cp8, n, h, w, _lib.ptr(wblob8), _lib.ptr(bblob8),
                             1. / scale, _lib.stream_ptr()), 'stem7 fp8')
    torch.cuda.synchronize()
    got8 = codes.cpu().view(torch.float8_e4m3fn).float() * scale
    assert torch.isfinite(got8).all() and (cout == cp8 or got8[..., cout:].abs().max().item() == 0)
    got8 = got8[..., :cout].permute(0, 3, 1, 2)
    err8 = (got8 - ref).abs()
    assert (err8 > ref.abs() * 0.0725 + scale * 2 ** -9 * 1.01 + 1e-2 * max(ref.abs().max().item(), 1.)).sum().item() == 0, \
        f'fp8 stem output: max err {err8.max().item():.3e}'  # half an e4m3 code step + the bf16 tolerance of the conv
    bad_in = xin.clone()
    bad_in[0, 0, 1, 1] = 1.5
    if dtype == 0:  # the range flag of the reference's Normalize assert (models/commons.py:694-697)
        _lib.check(lib.cpn_convert_input_stem(_lib.ptr(bad_in.to(dev)), 0, _lib.ptr(pad), n, cin, h, w, _lib.ptr(flag),
                                              _lib.stream_ptr()), 'input_stem')
        assert flag.item() == 1


@pytest.mark.parametrize('name', ['1x1_flagship_tile', '1x1_flagship_tile_res'])
def test_conv_pointwise_register_weight_loop(dev, name, monkeypatch):
    """CPN_PWR=1 selects MODE_PWR (the 1x1 convs' weight fragments from L2 straight into registers, 8x256 tile; round-3
    experiment, neutral on the MI355X: profiles/r03_kernel_experiments.txt).  Same K order and MFMA sequence as MODE_PW:
    bit-identical outputs."""
    monkeypatch.setenv('CPN_PWR', '0')
    lds, ref, _ = run_conv(dev, **CONV_CASES[name])
    monkeypatch.setenv('CPN_PWR', '1')
    rw, _, _ = run_conv(dev, **CONV_CASES[name])
    assert torch.equal(rw, lds), f'max abs {(rw - lds).abs().max().item():.3e}'
    assert (rw - ref).abs().max().item() < 5e-2 * max(ref.abs().max().item(), 1.)


def test_maxpool_bilinear_input(dev):
    from celldetection_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 40, 32, 64, generator=g).to(torch.bfloat16).float()
    d = to_nhwc_bf16(x.to(dev), 64)
    for k, s, p in ((3, 2, 1), (2, 2, 0)):
        ho, wo = (32 + 2 * p - k) // s + 1, (64 + 2 * p - k) // s + 1
        out = torch.empty(2, ho, wo, 64, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.cpn_maxpool2d(_lib.ptr(d), _lib.ptr(out), 2, 32, 64, 64, k, s, p, _lib.stream_ptr()))
        torch.cuda.synchronize()
        ref = F.max_pool2d(x, k, s, p)
        assert torch.equal(from_nhwc(out.cpu(), 40), ref), f'maxpool {k}/{s}/{p}'
    out = torch.empty(2, 64, 128, 64, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.cpn_resize_bilinear(_lib.ptr(d), _lib.ptr(out), 2, 32, 64, 64, 128, 64, _lib.stream_ptr()))
    torch.cuda.synchronize()
    ref = F.interpolate(x, (64, 128), mode='bilinear', align_corners=False)
    err = (from_nhwc(out.cpu(), 40) - ref).abs().max().item()
    assert err < 2e-2, f'bilinear max err {err}'
    # input conversion + range flag
    xin = torch.rand(2, 3, 32, 64, generator=g)
    for dtype in (0, 1):
        src = xin.to(dev) if dtype == 0 else (xin * 255).to(torch.uint8).to(dev)
        out = torch.full((2, 32, 64, 32), 7., dtype=torch.bfloat16, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.cpn_convert_input(_lib.ptr(src.contiguous()), dtype, _lib.ptr(out), 2, 3, 32, 64, 32,
                                         _lib.ptr(flag), _lib.stream_ptr()))
        torch.cuda.synchronize()
        exp = xin if dtype == 0 else (xin * 255).to(torch.uint8).float() / 255
        assert torch.equal(from_nhwc(out.cpu(), 3), exp.to(torch.bfloat16).float())
        assert out[..., 3:].abs().max().item() == 0 and flag.item() == 0
    bad = xin.clone()
    bad[1, 2, 5, 5] = 1.5
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.cpn_convert_input(_lib.ptr(bad.to(dev)), 0, _lib.ptr(out), 2, 3, 32, 64, 32, _lib.ptr(flag),
                                     _lib.stream_ptr()))
    assert flag.item() == 1


# ------------------------------------------------------------------------------------------------------------
# decode / NMS: bit-exact vs oracle + golden
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def gops():
    return np.load(os.path.join(G, 'ops.npz'))


@pytest.mark.parametrize('tag,samples', [('a', 32), ('b', 128), ('c', 7), ('d', 64)])
def test_fouriers2contours_golden(dev, gops, tag, samples):
    from celldetection_amd import ops
    out, _ = ops.fouriers2contours(torch.as_tensor(gops[f'f2c_{tag}_fourier']).to(dev),
                                   torch.as_tensor(gops[f'f2c_{tag}_loc']).to(dev), samples=samples)
    np.testing.assert_array_equal(out.cpu().numpy(), gops[f'f2c_{tag}_out'])


def test_fouriers2contours_custom_sampling_golden(dev, gops):
    """``sampling=`` of ops/cpn.py:44-95: one caller-supplied sampling vector shared by all contours (golden from the reference)."""
    from celldetection_amd import ops
    samp = torch.as_tensor(gops['f2c_s_sampling'])
    out, s_out = ops.fouriers2contours(torch.as_tensor(gops['f2c_s_fourier']).to(dev), torch.as_tensor(gops['f2c_s_loc']).to(dev),
                                       samples=5, sampling=samp.to(dev))
    np.testing.assert_array_equal(out.cpu().numpy(), gops['f2c_s_out'])
    assert torch.equal(s_out.cpu(), samp)
    with pytest.raises(NotImplementedError):
        ops.fouriers2contours(torch.zeros(9, 6, 4, device=dev), torch.zeros(9, 2, device=dev), sampling=torch.rand(9, 32))


def test_local_refinement_golden(dev, gops):
    from celldetection_amd import ops
    for iters in (1, 4):
        out = ops.local_refinement(torch.as_tensor(gops['refine_in']).to(dev), torch.as_tensor(gops['refine_map']).to(dev),
                                   iters, torch.as_tensor(gops['refine_b']).to(dev))
        np.testing.assert_array_equal(out.cpu().numpy(), gops[f'refine_out_{iters}'])


@pytest.mark.parametrize('thr', [.2, .5, 0.])
def test_nms_golden(dev, gops, thr):
    from celldetection_amd import ops
    keep = ops.nms(torch.as_tensor(gops['nms_boxes']).to(dev), torch.as_tensor(gops['nms_scores']).to(dev), thr)
    np.testing.assert_array_equal(keep.cpu().numpy(), gops[f'nms_keep_{thr}'])


def test_nmsi_golden(dev, gops):
    from celldetection_amd import ops
    b, s = torch.as_tensor(gops['nms_boxes']).to(dev), torch.as_tensor(gops['nms_scores']).to(dev)
    keep = ops.batched_box_nmsi([b], [s], .2, batch_size=128)[0]
    np.testing.assert_array_equal(keep.cpu().numpy(), gops['nmsi_chunked_keep'])
    keeps = ops.batched_box_nmsi([b, b[:50], b[:0]], [s, s[:50], s[:0]], .2)
    np.testing.assert_array_equal(keeps[0].cpu().numpy(), gops['nms_keep_0.2'])
    np.testing.assert_array_equal(keeps[1].cpu().numpy(), gops['nmsi_plain_keep'])
    assert keeps[2].numel() == 0


def test_border_golden(dev, gops):
    from celldetection_amd import ops
    con = torch.as_tensor(gops['border_in']).to(dev)
    flags = [(True, True, True, True), (False, True, False, True), (True, False, True, False)]
    for i, (top, right, bottom, left) in enumerate(flags):
        keep = ops.remove_border_contours(con, (48, 64), 4, top=top, right=right, bottom=bottom, left=left,
                                          offsets=torch.as_tensor(gops['border_offsets']))
        np.testing.assert_array_equal(keep.cpu().numpy(), gops[f'border_keep_{i}'])
    keep = ops.filter_contours_by_stitching_rule(con, (48, 64), torch.as_tensor(gops['stitch_overlaps']),
                                                 offsets=torch.as_tensor(gops['border_offsets']).to(dev))
    np.testing.assert_array_equal(keep.cpu().numpy(), gops['stitch_keep'])


@pytest.mark.parametrize('P,nseg', [(1, 1), (63, 1), (64, 1), (65, 2), (1000, 3), (5000, 4), (20000, 1)])
def test_nms_random_vs_oracle(dev, P, nseg):
    """Random boxes with many overlaps / ties / degenerate boxes; segmented NMS vs per-segment oracle."""
    import cpn_oracle as orc
    from celldetection_amd import ops
    rng = np.random.default_rng(P)
    xy = rng.uniform(0, 200 if P > 2000 else 60, (P, 2)).astype(np.float32)
    wh = rng.uniform(0, 20, (P, 2)).astype(np.float32)
    boxes = np.concatenate((xy, xy + wh), 1)
    boxes[::17, 2:] = boxes[::17, :2]  # zero-area
    scores = rng.random(P).astype(np.float32)
    scores[::5] = scores[0]  # ties
    cuts = sorted(rng.integers(0, P + 1, nseg - 1).tolist())
    offs = [0] + cuts + [P]
    bl = [torch.as_tensor(boxes[offs[i]:offs[i + 1]]).to(dev) for i in range(nseg)]
    sl = [torch.as_tensor(scores[offs[i]:offs[i + 1]]).to(dev) for i in range(nseg)]
    keeps = ops.batched_box_nmsi(bl, sl, .3)
    for i in range(nseg):
        exp = orc.nms(boxes[offs[i]:offs[i + 1]], scores[offs[i]:offs[i + 1]], .3)
        np.testing.assert_array_equal(keeps[i].cpu().numpy(), exp, err_msg=f'segment {i}')


@pytest.mark.parametrize('shape,density', [((2, 32, 48), .1), ((3, 17, 23), .5), ((1, 64, 64), 0.), ((2, 16, 16), 1.),
                                           ((16, 256, 256), .02)])
def test_compact_and_decode_vs_oracle(dev, shape, density):
    """Synthetic head maps -> HIP compaction + fused decode (+offsets) == oracle post-processing, bit-exact."""
    import cpn_oracle as orc
    from celldetection_amd import ops
    n, h, w = shape
    H, W = 2 * h, 2 * w
    g = torch.Generator().manual_seed(h * w)
    order, samples = 6, 33
    scores = torch.rand(n, 1, h, w, generator=g)
    thr = 1. - density if density > 0 else 2.
    if density >= 1.:
        thr = -1.
    loc = torch.randn(n, 2, h, w, generator=g)
    fourier = torch.randn(n, 4 * order, h, w, generator=g) * 4
    ref = (torch.rand(n, 2, H, W, generator=g) * 2 - 1) * 3
    offsets = torch.randint(-50, 5000, (n, 2), generator=g)
    idx, counts, _ = ops.compact_scores(scores.to(dev), thr)
    flat = ops.decode_proposals(idx, scores.to(dev), loc.to(dev), fourier.to(dev), ref.to(dev), size=(H, W),
                                order=order - 1, samples=samples, iterations=3, offsets=offsets)
    exp = orc.cpn_postprocess(scores, loc, ref, fourier, input_size=(H, W), order=order - 1, samples=samples,
                              score_thresh=thr, refinement_iterations=3, nms=False, offsets=offsets.numpy(),
                              scores_are_probabilities=True)
    assert counts == [len(e) for e in exp['scores']]
    for k in ('contours', 'boxes', 'scores', 'locations', 'fourier', 'contour_proposals'):
        np.testing.assert_array_equal(flat[k].cpu().numpy(), np.concatenate(exp[k]), err_msg=k)
    # no refinement: proposals alias clamped contours
    flat = ops.decode_proposals(idx, scores.to(dev), loc.to(dev), fourier.to(dev), None, size=(H, W), order=order,
                                samples=samples, iterations=0)
    exp = orc.cpn_postprocess(scores, loc, None, fourier, input_size=(H, W), order=order, samples=samples,
                              score_thresh=thr, refinement_iterations=0, nms=False, scores_are_probabilities=True)
    for k in ('contours', 'boxes', 'contour_proposals'):
        np.testing.assert_array_equal(flat[k].cpu().numpy(), np.concatenate(exp[k]), err_msg='norefine ' + k)


def test_box_voting_vs_oracle(dev):
    """filter_by_box_voting (ops/boxes.py:52-83) vs the numpy restatement: votes to 1e-5 relative, same keep set
    (boxes whose vote is within 1e-4 of the threshold are excluded from the set comparison)."""
    import cpn_oracle as orc
    from celldetection_amd import ops
    g = torch.Generator().manual_seed(5)
    xy = torch.rand(900, 2, generator=g) * 120
    wh = torch.rand(900, 2, generator=g) * 30 + 1
    boxes = torch.cat((xy, xy + wh), 1)
    boxes[7] = boxes[6]
    for thr, min_vote in ((.2, 2.), (.5, 1.5), (.1, 1.)):
        keep, votes = ops.filter_by_box_voting(boxes.to(dev), thr, min_vote, return_votes=True)
        ekeep, evotes = orc.filter_by_box_voting(boxes.numpy(), thr, min_vote)
        all_votes = orc.filter_by_box_voting(boxes.numpy(), thr, -1.)[1]
        sure = np.abs(all_votes - min_vote) > 1e-4
        got = np.zeros(len(boxes), bool)
        got[keep.cpu().numpy()] = True
        exp = np.zeros(len(boxes), bool)
        exp[ekeep] = True
        np.testing.assert_array_equal(got[sure], exp[sure])
        assert keep.dtype == torch.int32
        both = got & exp
        gv = np.zeros(len(boxes), np.float32)
        gv[keep.cpu().numpy()] = votes.cpu().numpy()
        np.testing.assert_allclose(gv[both], all_votes[both], rtol=1e-5, atol=1e-5)
    assert ops.filter_by_box_voting(boxes[:0].to(dev), .2, 1.).numel() == 0


# ---- fp8 (e4m3) conv kernel: v_mfma_scale_f32_32x32x64_f8f6f4 ---------------------------------------------------------
def _to_nhwc_fp8(x, scale, cpad):
    """NCHW fp32 -> (NHWC e4m3 codes as uint8 with zero-padded channels, dequantised NCHW fp32)."""
    n, c, h, w = x.shape
    q = (x / scale).clamp(-448, 448).to(torch.float8_e4m3fn)
    out = torch.zeros(n, h, w, cpad, dtype=torch.uint8, device=x.device)
    out[..., :c] = q.view(torch.uint8).permute(0, 2, 3, 1)
    return out.contiguous(), q.float() * scale


FP8_CASES = {
    '3x3_64_64': dict(n=2, h=32, w=32, cin=64, cout=64, k=3),
    '1x1_128_256': dict(n=2, h=32, w=32, cin=128, cout=256, k=1),
    '1x1_odd_chunks_res': dict(n=1, h=32, w=64, cin=192, cout=128, k=1, res=True),
    '3x3_odd_channels': dict(n=1, h=32, w=64, cin=24, cout=40, k=3),
    '3x3_s2': dict(n=2, h=64, w=64, cin=64, cout=128, k=3, stride=2),
    '7x7_256_big_tile': dict(n=1, h=64, w=64, cin=128, cout=256, k=7),
    '3x3_concat_up': dict(n=2, h=32, w=32, cin=64, cout=64, k=3, cin1=128, up1=True),
    '3x3_grouped_cpg8': dict(n=1, h=32, w=32, cin=256, cout=256, k=3, groups=32),
    '7x7_stem_s2': dict(n=2, h=64, w=64, cin=3, cout=64, k=7, stride=2, bias=False),
    'final_f32': dict(n=1, h=32, w=32, cin=128, cout=20, k=1, bn=False, act='none', out_f32=True),
    'fused_head_64': dict(n=1, h=32, w=64, cin=64, cout=64, k=7, fuse_cout=20, fuse_act='none'),
}


@pytest.mark.parametrize('name', list(FP8_CASES))
def test_conv_fp8_vs_dequantised_reference(dev, name):
    """fp8 conv kernel vs an fp32 PyTorch conv on the SAME e4m3-quantised operands (products of e4m3 values are exact
    in fp32, so only the summation order and the output rounding differ): fp32 outputs to 2e-3 of the output range,
    e4m3 outputs within one code step (2^-3 relative) + the subnormal step."""
    from celldetection_amd import _lib, graph
    cfg = dict(FP8_CASES[name])
    n, h, w, cin, cout, k = (cfg[x] for x in ('n', 'h', 'w', 'cin', 'cout', 'k'))
    stride, groups, cin1, up1 = cfg.get('stride', 1), cfg.get('groups', 1), cfg.get('cin1', 0), cfg.get('up1', False)
    res, act, out_f32, fuse_cout = cfg.get('res', False), cfg.get('act', 'relu'), cfg.get('out_f32', False), \
        cfg.get('fuse_cout', 0)
    g = torch.Generator().manual_seed(11)
    P = graph.Plan()
    s0 = P.tensor(cin, 1)
    s1 = P.tensor(cin1, 2 if up1 else 1) if cin1 else None
    r = P.tensor(cout, stride) if res else None
    P.conv(s0, cout, k, w='c.', bn='b.' if cfg.get('bn', True) else None, bias=cfg.get('bias', True), stride=stride,
           groups=groups, act=act, src1=s1, up1=up1, res=r,
           out_index=_lib.OUT_SCORES if (out_f32 or fuse_cout) else None,
           fuse=dict(w='f.', cout=fuse_cout, act=cfg.get('fuse_act', 'none'), act_scale=3.) if fuse_cout else None)
    out_f32 = out_f32 or bool(fuse_cout)
    sd = {}
    for key, shape, kind in P.entries:
        if key.endswith('running_var'):
            sd[key] = torch.rand(shape, generator=g) + .5
        elif key.endswith('num_batches_tracked'):
            sd[key] = torch.zeros((), dtype=torch.long)
        elif len(shape) == 4:
            sd[key] = torch.randn(shape, generator=g) / np.sqrt(np.prod(shape[1:]))
        else:
            sd[key] = torch.randn(shape, generator=g) * .5 + (1. if key.endswith('b.weight') else 0.)
    x0 = torch.randn(n, cin, h, w, generator=g)
    x1 = torch.randn(n, cin1, h // 2 if up1 else h, w // 2 if up1 else w, generator=g) * 2 if cin1 else None
    ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    xr = torch.randn(n, cout, ho, wo, generator=g) if res else None
    p64 = lambda c: (c + 63) // 64 * 64
    scales = {s0: float(x0.abs().max()) / 448}
    if cin1:
        scales[s1] = float(x1.abs().max()) / 448
    if res:
        scales[r] = float(xr.abs().max()) / 448
    d0, x0q = _to_nhwc_fp8(x0.to(dev), scales[s0], p64(cin))
    d1, x1q = _to_nhwc_fp8(x1.to(dev), scales[s1], p64(cin1)) if cin1 else (None, None)
    dr, xrq = _to_nhwc_fp8(xr.to(dev), scales[r], p64(cout)) if res else (None, None)
    # reference on the dequantised operands; the weight codes come from the packer (dequantised through mult)
    wf, bf = graph._fold(sd, P.ops[0])
    xin = x0q.cpu()
    if cin1:
        xin = torch.cat((xin, F.interpolate(x1q.cpu(), scale_factor=2, mode='nearest') if up1 else x1q.cpu()), 1)
    # output scale from an fp32 dry run
    ref_full = F.conv2d(xin, wf.float(), bf.float(), stride, k // 2, 1, groups)
    if res:
        ref_full = ref_full + xrq.cpu()
    if act == 'relu':
        ref_full = F.relu(ref_full)
    dst_id = P.ops[0]['dst']
    if dst_id is not None:
        scales[dst_id] = float(ref_full.abs().max()) / 448
    tens, ops, wblob, bblob, mblob, op_scales = graph.pack(P, sd, dev, precision='fp8', act_scales=scales)
    op = ops[0]
    lib = _lib.load()
    if out_f32:
        dst = torch.full((n, fuse_cout or cout, ho, wo), float('nan'), dtype=torch.float32, device=dev)
        dstride = 0
    else:
        dst = torch.full((n, ho, wo, p64(cout)), 0x7f, dtype=torch.uint8, device=dev)
        dstride = p64(cout)
    _lib.check(lib.cpn_conv2d_fp8(op, _lib.ptr(d0), d0.shape[-1], _lib.ptr(d1), 0 if d1 is None else d1.shape[-1],
                                  _lib.ptr(dr), 0 if dr is None else dr.shape[-1], _lib.ptr(dst), dstride, n, h, w,
                                  _lib.ptr(wblob), _lib.ptr(bblob), _lib.ptr(mblob), op_scales[0][0], op_scales[0][1],
                                  _lib.stream_ptr()), 'conv2d_fp8')
    torch.cuda.synchronize()
    # dequantised weights exactly as the kernel sees them: e4m3(w * s_in / s_w) * s_w / s_in
    cig = (cin + cin1) // groups
    wq = torch.empty_like(wf)
    for co in range(cout):
        row = wf[co].clone()
        if cin1:
            row[:cin] *= scales[s0]
            row[cin:] *= scales[s1]
        else:
            row *= scales[s0]
        sw = max(float(row.abs().max()) / 448., 1e-30)
        # per-(bundle, cout) scale = max over the whole packed row, which equals the row max here
        rq = (row / sw).float().to(torch.float8_e4m3fn).float() * sw
        if cin1:
            rq[:cin] /= scales[s0]
            rq[cin:] /= scales[s1]
        else:
            rq /= scales[s0]
        wq[co] = rq
    ref = F.conv2d(xin.double(), wq.double(), bf.double(), stride, k // 2, 1, groups).float()
    if res:
        ref = ref + xrq.cpu()
    if act == 'relu':
        ref = F.relu(ref)
    if fuse_cout:
        w2 = sd['f.weight'].to(torch.bfloat16).float()
        ref = F.conv2d(ref.to(torch.bfloat16).float(), w2, sd['f.bias'])
    if out_f32:
        got = dst.cpu()
        tol = (5e-2 if fuse_cout else 2e-3) * float(ref.abs().max())
        assert float((got - ref).abs().max()) <= tol, (name, float((got - ref).abs().max()), tol)
    else:
        codes = dst.cpu()[..., :cout].permute(0, 3, 1, 2).contiguous()
        got = codes.view(torch.float8_e4m3fn).float() * scales[dst_id]
        assert not bool(torch.isnan(got).any()), name
        err = (got - ref).abs()
        bound = ref.abs() * 0.0725 + scales[dst_id] * 2 ** -9 * 1.01  # half a code step (RNE) + margin, subnormal step
        bad = float((err > bound).float().mean())
        assert bad <= 1e-3, (name, bad, float(err.max()))
        if p64(cout) > cout:  # padded output channels must be exact zeros
            assert int(dst.cpu()[..., cout:].max()) in (0, 0x80) or int(dst.cpu()[..., cout:].max()) == 0


@pytest.mark.parametrize('nb', [6, 3])
def test_bucketed_local_refinement_vs_reference_golden(dev, nb):
    """ops.local_refinement(num_buckets=...) (refine_kernel with the bucket tables) bit-exact vs the reference's
    bucketed local_refinement (cpn.py:72-82) on the golden vectors."""
    from celldetection_amd import ops
    g = np.load(os.path.join(G, 'ops.npz'))
    got = ops.local_refinement(torch.as_tensor(g['refine_in']).to(dev), torch.as_tensor(g[f'refine_bucket_map_{nb}']).to(dev),
                               3, torch.as_tensor(g['refine_b']).to(dev), num_buckets=nb)
    np.testing.assert_array_equal(got.cpu().numpy(), g[f'refine_bucket_out_{nb}'])


@pytest.mark.parametrize('P,extent', [(1, 50), (700, 100), (5000, 200), (40000, 800), (40000, 120)])
def test_nms_binned_equals_dense_and_oracle(dev, P, extent):
    """Spatially binned NMS (slide-level path) = dense bit-mask NMS = CPU oracle: identical keep lists incl. ties,
    zero-area boxes (NaN IoU), duplicates and dense clusters (extent 120: ~thousands of overlaps per box)."""
    import cpn_oracle as orc
    from celldetection_amd import ops
    rng = np.random.default_rng(P + extent)
    xy = rng.uniform(0, extent, (P, 2)).astype(np.float32)
    wh = rng.uniform(0, 20, (P, 2)).astype(np.float32)
    boxes = np.concatenate((xy, xy + wh), 1)
    boxes[::17, 2:] = boxes[::17, :2]  # zero-area
    if P > 100:
        boxes[5] = boxes[4]
        boxes[50, :] = np.nan
    scores = rng.random(P).astype(np.float32)
    scores[::5] = scores[0]  # ties
    b, s = torch.as_tensor(boxes).to(dev), torch.as_tensor(scores).to(dev)
    for thr in (.2, 0.):
        got, st = ops.nms_binned(b, s, thr, return_stats=True)
        dense, cnt = ops._nms_segments(b, s, [0, P], thr)
        np.testing.assert_array_equal(got.cpu().numpy(), dense[:cnt[0]].cpu().numpy())
        if P <= 40000 and thr == .2:
            np.testing.assert_array_equal(got.cpu().numpy(), orc.nms(boxes, scores, thr))
        print(f'nms_binned P={P} extent={extent} thr={thr}: kept {got.numel()}, edges {st["edges"]}, sweeps {st["sweeps"]}, '
              f'workspace {st["workspace_bytes"] / 1e6:.1f} MB')


@pytest.mark.parametrize('n_big,chain', [(1, 0), (7, 0), (300, 0), (3, 400)])
def test_nms_binned_outliers_and_long_chains(dev, n_big, chain):
    """a few tile-sized outlier boxes among 40 000 cell-sized ones must not collapse the grid (they go through the
    oversized list: every box tests them, they scan every cell they reach), more outliers than the list holds fall back to
    the coarse grid, and a suppression chain of hundreds of links (box i suppresses i+1 only) resolves without one host
    round trip per four sweeps.  Keep lists = dense path = oracle in every case."""
    import cpn_oracle as orc
    from celldetection_amd import ops
    P = 40000
    rng = np.random.default_rng(n_big * 1000 + chain)
    xy = rng.uniform(0, 4000, (P, 2)).astype(np.float32)
    wh = rng.uniform(2, 20, (P, 2)).astype(np.float32)
    boxes = np.concatenate((xy, xy + wh), 1)
    scores = rng.random(P).astype(np.float32)
    big = rng.choice(P, n_big, replace=False)
    c = rng.uniform(500, 3500, (n_big, 2)).astype(np.float32)
    half = rng.uniform(300, 2000, (n_big, 2)).astype(np.float32)
    boxes[big] = np.concatenate((c - half, c + half), 1)
    scores[big[::2]] = 2. + rng.random(len(big[::2])).astype(np.float32)  # some outliers outrank everything, some do not
    if chain:  # boxes shifted by 45 % of their width: IoU(i, i+1) = 0.38 > thr, IoU(i, i+2) = 0.05 < thr; descending scores
        idx = rng.choice(np.setdiff1d(np.arange(P), big), chain, replace=False)
        x0 = 5000. + 4.5 * np.arange(chain, dtype=np.float32)
        boxes[idx] = np.stack((x0, np.full(chain, 5000., np.float32), x0 + 10., np.full(chain, 5010., np.float32)), 1)
        scores[idx] = 1.9 - 1e-3 * np.arange(chain, dtype=np.float32)
    b, s = torch.as_tensor(boxes).to(dev), torch.as_tensor(scores).to(dev)
    got, st = ops.nms_binned(b, s, .2, return_stats=True)
    dense, cnt = ops._nms_segments(b, s, [0, P], .2)
    np.testing.assert_array_equal(got.cpu().numpy(), dense[:cnt[0]].cpu().numpy())
    np.testing.assert_array_equal(got.cpu().numpy(), orc.nms(boxes, scores, .2))
    print(f'outliers {n_big} chain {chain}: kept {got.numel()}, edges {st["edges"]}, sweeps {st["sweeps"]}')
    if chain:
        assert st['sweeps'] >= chain // 2  # (the chain really is sequential: alternate links survive)
    if n_big <= 7:
        assert st['edges'] < 40 * P  # the grid did not collapse: a handful of candidates per box, not thousands


def test_nms_binned_slide_scale(dev):
    """10^6 detections spread like cells on a slide: < 1 GB of workspace (the dense mask would need 125 GB), the keep
    list is sorted by score, idempotent, and agrees with the dense path on a 60 000-box crop of the same set."""
    from celldetection_amd import ops
    P = 1_000_000
    g = torch.Generator().manual_seed(5)
    ctr = torch.rand(P, 2, generator=g) * 16384
    wh = torch.rand(P, 2, generator=g) * 24 + 4
    boxes = torch.cat((ctr - wh / 2, ctr + wh / 2), 1).to(dev)
    scores = torch.rand(P, generator=g).to(dev)
    keep, st = ops.nms_binned(boxes, scores, .2, return_stats=True)
    print(f'slide-scale NMS: kept {keep.numel()} of {P}, edges {st["edges"]}, sweeps {st["sweeps"]}, '
          f'workspace {st["workspace_bytes"] / 1e6:.0f} MB')
    assert st['workspace_bytes'] < 1e9
    ks = scores[keep]
    assert bool((ks[:-1] >= ks[1:]).all()) and 0 < keep.numel() < P
    again = ops.nms_binned(boxes[keep], ks, .2)
    assert torch.equal(again, torch.arange(keep.numel(), device=dev))
    sub = (ctr[:, 0] < 4000) & (ctr[:, 1] < 4000)
    idx = sub.nonzero().squeeze(1)[:60000].to(dev)
    a = ops.nms_binned(boxes[idx], scores[idx], .2)
    d, c = ops._nms_segments(boxes[idx], scores[idx], [0, idx.numel()], .2)
    assert torch.equal(a, d[:c[0]])


def test_border_batched_equals_per_tile(dev, gops):
    from celldetection_amd import ops
    con = torch.as_tensor(gops['border_in']).to(dev)  # [40, 12, 2]
    P = con.shape[0]
    img = torch.arange(P, dtype=torch.int32, device=dev) % 3
    sides = torch.tensor([15, 10, 5], dtype=torch.int32, device=dev)  # all / right+left / top+bottom
    offs = torch.tensor([[-3., 5.], [0., 0.], [2., -1.]], device=dev)
    got = ops.remove_border_contours_batched(con, img, sides, offs, (48, 64), 4).bool()
    for n, (top, right, bottom, left) in enumerate([(True, True, True, True), (False, True, False, True),
                                                    (True, False, True, False)]):
        m = img == n
        exp = ops.remove_border_contours(con[m], (48, 64), 4, top=top, right=right, bottom=bottom, left=left, offsets=offs[n])
        assert torch.equal(got[m], exp)
    np.testing.assert_array_equal(got[img == 0].cpu().numpy(), gops['border_keep_0'][(np.arange(P) % 3) == 0])


@pytest.mark.parametrize('shape,size', [((2, 1, 16, 24), (64, 96)), ((1, 3, 19, 26), (75, 101)), ((2, 2, 76, 102), (75, 101)),
                                        ((1, 1, 9, 9), (9, 31))])
def test_resize_bilinear_f32_vs_torch_cpu(dev, shape, size):
    """fp32 NCHW bilinear resize (score-bound masks, `_equal_size`) vs torch CPU's F.interpolate."""
    from celldetection_amd.cpn import _equal_size
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(1))
    ref = F.interpolate(x, size, mode='bilinear', align_corners=False)
    got = _equal_size(x.to(dev), torch.empty(1, 1, *size)).cpu()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-6)  # a few ulp: torch may contract to FMA
    m = (x > .5).float()  # 0/1 masks: what the slide loop passes as score bounds
    np.testing.assert_array_equal(_equal_size(m.to(dev), torch.empty(1, 1, *size)).cpu().numpy() > .9,
                                  F.interpolate(m, size, mode='bilinear', align_corners=False).numpy() > .9)


@pytest.mark.parametrize('dtype', [torch.bool, torch.uint8, torch.float32, torch.int64, torch.float16])
def test_windows_any_equals_per_tile_any(dev, dtype):
    """Tile pre-filter of the slide loop (TileLoader skips tiles with an empty mask crop, cpn_inference.py:88-100): ONE window-any
    launch over the tiling table == ``mask[slices].any()`` per tile -- sparse seeds, seeds on window borders, -0.0 / NaN floats."""
    from celldetection_amd import ops, util
    H, W = 700, 1000
    slices = list(util.get_tiling_slices((H, W), (128, 160), (96, 100))[0])
    win = [[s[0].start, s[0].stop, s[1].start, s[1].stop] for s in slices]
    g = torch.Generator().manual_seed(5)
    mask = torch.zeros(H, W)
    ys, xs = torch.randint(0, H, (12,), generator=g), torch.randint(0, W, (12,), generator=g)
    mask[ys, xs] = 1.
    mask[127, 159] = 1.   # last pixel of the first window
    mask[H - 1, W - 1] = 1.
    m = mask.to(dtype).to(dev)
    if dtype == torch.float32:
        m[300, 500] = -0.      # zero
        m[5, 700] = float('nan')  # non-zero
    got = ops.windows_any(m, win)
    exp = [bool(torch.any(m[s])) for s in slices]
    assert got == exp and 0 < sum(exp) < len(exp)
    assert ops.windows_any(torch.zeros(H, W, dtype=dtype, device=dev), win) == [False] * len(win)
    assert ops.windows_any(m, []) == []
    with pytest.raises(ValueError):
        ops.windows_any(m, [[0, H + 1, 0, 10]])
