"""Layer plans of the CPN model families and weight packing for the HIP conv engine.

A *plan* is built from constructor hyper-parameters only (no weights): an ordered list of state-dict entries
(names + shapes identical to the reference's ``state_dict()``, so reference checkpoints load unchanged) and a small
IR of fused ops (conv + folded BN + activation (+ residual / virtual concat / nearest upsample)).  ``pack`` turns a
state dict into the bf16 weight blob + fp32 bias blob + ``cpn_op_desc`` array the native executor consumes.

Reference structures mirrored (file:line relative to the reference repository):
  ResNet / ResNeXt encoders   celldetection/models/resnet.py:56-193,265-297,300-460
  UNetEncoder (U22)           celldetection/models/unet.py:29-58
  GeneralizedUNet decoder     celldetection/models/unet.py:62-249
  FeaturePyramidNetwork       celldetection/models/fpn.py:79-134 (+ torchvision FPN.forward)
  ReadOut heads / CPNCore     celldetection/models/commons.py:461-511, celldetection/models/cpn.py:126-283
"""
import math
import os

import torch

from . import _lib

__all__ = ['Plan', 'build_plan', 'BACKBONES']


def _pad32(c):
    return (int(c) + 31) // 32 * 32


class Plan:
    def __init__(self):
        self.entries = []      # (key, shape, kind) kind in {'param', 'buffer', 'long'}
        self.tensors = []      # dict(c=real channels, down=...)
        self.ops = []          # IR dicts
        self.meta = {}

    # ---- state-dict entries
    def conv_keys(self, prefix, cout, cin_g, k, bias):
        self.entries.append((prefix + 'weight', (cout, cin_g, k, k), 'param'))
        if bias:
            self.entries.append((prefix + 'bias', (cout,), 'param'))

    def bn_keys(self, prefix, c):
        self.entries += [(prefix + 'weight', (c,), 'param'), (prefix + 'bias', (c,), 'param'),
                         (prefix + 'running_mean', (c,), 'buffer'), (prefix + 'running_var', (c,), 'buffer'),
                         (prefix + 'num_batches_tracked', (), 'long')]

    # ---- IR
    def tensor(self, c, down, phases=1):
        """``phases`` = 4: sub-pixel phase tensor, channel layout [phase (py, px)][padded c] (see ``conv(sub='phase')``)."""
        self.tensors.append(dict(c=int(c), down=int(down), phases=int(phases)))
        return len(self.tensors) - 1

    def conv(self, src0, cout, k, *, w, bn=None, bias=False, stride=1, pad=None, groups=1, act='none', act_scale=0.,
             src1=None, up0=False, up1=False, res=None, res_up=False, out_index=None, fuse=None, deferred=False,
             sub=None, dst=None, share=None):
        """Adds conv(+BN)(+residual)(+act); returns the destination tensor id (None for external outputs).
        ``sub``: member of a sub-pixel triple (``_subpixel_pair``): 'head' = the conv as the reference states it, ('phase',
        c0) / ('lateral', c0) = its decomposition (they share the head's state-dict entries and add none); ``dst``: write
        into an existing tensor (the lateral op writes the head's destination).  ``share`` = (first, last input channel of
        the conv stated by the keys ``w`` / ``bn``, with_bias): the op applies only that channel range of a conv whose state-dict
        entries the caller registers itself (``_head_input``: Fuse2d over three and more features as partial 1x1 convs); a
        fourth element c = the first source is the running sum of the parts so far (c channels, identity weights)."""
        pad = k // 2 if pad is None else pad
        t0 = self.tensors[src0]
        if up0 == 'bilinear':  # source read through a bilinear resize to the input size (nominal down factor 1)
            assert src1 is None and stride == 1 and k > 1
            down_in = 1
        else:
            down_in = t0['down'] // (2 if up0 else 1)
            assert t0['down'] % (2 if up0 else 1) == 0
        cin = t0['c'] + (self.tensors[src1]['c'] if src1 is not None else 0)
        if src1 is not None and not up1:  # (a resized second source takes the size of the first one: any ratio)
            assert self.tensors[src1]['down'] == down_in
        down_out = down_in * stride
        scatter = isinstance(sub, tuple) and sub[0] == 'scatter'  # stands for a 3x3 conv over the x2-upsampled source
        if isinstance(sub, tuple) and sub[0] == 'blphase':  # bilinear phases: low-resolution source, full-resolution output
            down_out = max(t0['down'] // 2, 1)
        if scatter:
            assert t0['down'] % 2 == 0 and src1 is None and not up0 and k == 2
            down_out = t0['down'] // 2
        member = (isinstance(sub, tuple) and not scatter) or share is not None  # no state-dict entries of its own
        if dst is None:
            dst = self.tensor(cout, down_out, phases=4 if (isinstance(sub, tuple) and sub[0] == 'phase') else 1) if out_index is None else None
        if not member:
            self.conv_keys(w, cout, cin // groups, 3 if scatter else k, bias)
            if bn is not None:
                self.bn_keys(bn, cout)
            if fuse is not None:  # fused ReadOut tail: final 1x1 conv (bias) applied to this conv's activated output
                self.conv_keys(fuse['w'], fuse['cout'], cout, 1, True)
        self.ops.append(dict(op='conv', src0=src0, src1=src1, res=res, dst=dst, up0=up0, up1=up1, res_up=res_up, k=k,
                             stride=stride, pad=pad, groups=groups, cin=cin, cout=cout, w=w, bn=bn, bias=bias, act=act,
                             act_scale=act_scale, out_index=out_index, fuse=fuse, deferred=bool(deferred), sub=sub,
                             share=share))
        return dst

    def conv_pair(self):
        """Restates the last two conv ops -- conv1 1x1 + BN + ReLU -> grouped conv2 3x3 (stride 1 | 2) + BN + ReLU, the head of a
        ResNeXt bottleneck block (models/resnet.py:88-116,119-193) -- as ONE fused op (include/cpn_hip.h CPN_OP_CONV_PAIR,
        csrc/conv_pair.hip).  The executor runs it instead of the pair wherever the kernel's tiles fit and fill the chip;
        no state-dict entries, no weights of its own.  Returns False (and adds nothing) when the pair does not qualify."""
        c1, c2 = self.ops[-2], self.ops[-1]
        cpg = c2['cin'] // c2['groups']
        bw = cpg if cpg % 32 == 0 else (32 if 32 % cpg == 0 else 0)  # channels per packed bundle (_bundle_geometry)
        ok = (c1['op'] == c2['op'] == 'conv' and c1['k'] == 1 and c1['stride'] == 1 and c1['groups'] == 1 and
              c1['src1'] is None and c1['res'] is None and not c1['up0'] and c1['act'] == 'relu' and c1.get('sub') is None and
              c2['src0'] == c1['dst'] and c2['k'] == 3 and c2['stride'] in (1, 2) and c2['pad'] == 1 and c2['groups'] > 1 and
              c2['src1'] is None and c2['res'] is None and not c2['up0'] and c2['act'] == 'relu' and c2['cin'] == c2['cout'] and
              bw in (32, 64) and c2['cout'] % 128 == 0 and c1['dst'] is not None and c2['dst'] is not None)
        if ok:
            self.ops.append(dict(op='conv_pair', src0=c1['src0'], dst=c2['dst'], first=len(self.ops) - 2,
                                 w=c1['w'] + '+' + c2['w'].rsplit('.', 2)[-2] + '.'))
        return ok

    def conv_bridge(self):
        """Restates the last two conv ops -- the scattered bridge conv (``sub=('scatter', 0)``, 32 | 64 -> 64 channels, ReLU) and
        the 3x3 conv 64 -> 64 behind it = TwoConvNormRelu of a UNet bridge level (models/unet.py:92-107, commons.py:120-149) -- as
        ONE fused op (include/cpn_hip.h CPN_OP_CONV_BRIDGE, csrc/conv_igemm.hip MODE_BR): the full-resolution tensor between
        them stays in LDS.  The executor runs it instead of the pair wherever the output holds a 16 x 32 tile; no state-dict
        entries, no weights of its own.  Returns False (and adds nothing) when the pair does not qualify."""
        c1, c2 = self.ops[-2], self.ops[-1]
        p32 = lambda c: (c + 31) // 32 * 32
        ok = (c1['op'] == c2['op'] == 'conv' and isinstance(c1.get('sub'), tuple) and c1['sub'][0] == 'scatter' and
              c1['act'] == 'relu' and p32(c1['cout']) == 64 and p32(c1['cin']) in (32, 64) and c1['dst'] is not None and
              c2['src0'] == c1['dst'] and c2['src1'] is None and not c2['up0'] and c2['k'] == 3 and c2['stride'] == 1 and
              c2['pad'] == 1 and c2['groups'] == 1 and p32(c2['cout']) == 64 and c2.get('sub') is None and c2['dst'] is not None and
              c2['fuse'] is None and c2['res_up'] in (False, 0, 'shuffle'))
        if ok:
            self.ops.append(dict(op='conv_bridge', src0=c1['src0'], dst=c2['dst'], res=c2['res'], first=len(self.ops) - 2,
                                 w=c1['w'] + '+' + c2['w'].rsplit('.', 2)[-2] + '.'))
        return ok

    def act(self, src, name):
        """Elementwise activation as an op of its own (CPN_OP_ACT): the hidden activation of a ReadOut head other than ReLU."""
        t = self.tensors[src]
        dst = self.tensor(t['c'], t['down'])
        self.ops.append(dict(op='act', src0=src, dst=dst, act=name))
        return dst

    def maxpool(self, src, k, stride, pad):
        t = self.tensors[src]
        dst = self.tensor(t['c'], t['down'] * stride)
        self.ops.append(dict(op='maxpool', src0=src, dst=dst, k=k, stride=stride, pad=pad))
        return dst

    def bilinear_to_input(self, src, mode='bilinear'):
        """``_equal_size(features, inputs, mode)`` (models/cpn.py:109-115,277-278): bilinear | bicubic resize (align_corners=False) to
        the input size; the executor aliases source and destination when the sizes already agree."""
        dst = self.tensor(self.tensors[src]['c'], 1)
        self.ops.append(dict(op='bilinear', src0=src, dst=dst, mode=mode))
        return dst

    def input(self, in_channels):
        dst = self.tensor(in_channels, 1)
        self.ops.append(dict(op='input', dst=dst, in_channels=in_channels))
        return dst

    def hoist(self, begin, end):
        """Moves ops [begin, end) to the earliest position at which every tensor they read from outside the block has been written
        (execution order only: tensors, state-dict entries and results are untouched; the arena planner follows the new
        lifetimes).  Returns the number of positions the block moved up."""
        block = self.ops[begin:end]
        written = {o['dst'] for o in block if o.get('dst') is not None}
        reads = {o[k] for o in block for k in ('src0', 'src1', 'res') if o.get(k) is not None} - written
        ready = 0  # first position behind the last producer of a tensor the block reads
        for i, o in enumerate(self.ops[:begin]):
            if o.get('dst') in reads:
                ready = i + 1
        # (members of a sub-pixel triple / alternatives / fused pairs that restate the ops in front of them stay together)
        moved = True
        while moved:
            moved = False
            while ready < begin and (self.ops[ready].get('op') in ('conv_pair', 'conv_bridge') or self.ops[ready].get('alt') == 2 or
                                     (isinstance(self.ops[ready].get('sub'), tuple) and self.ops[ready]['sub'][0] != 'scatter')):
                ready += 1  # (a scatter conv OPENS a unit -- the block may go in front of it)
            # a fused op + the two convs it restates ([first, first + 1, fused]) are one unit: never insert inside it (the
            # ResNet18/34-UNets have no inner_blocks.0, so their bridge's scatter conv directly follows the heads' producer)
            for i, o in enumerate(self.ops[:begin]):
                if o['op'] in ('conv_pair', 'conv_bridge') and o['first'] < ready <= i:
                    ready, moved = i + 1, True
        if ready >= begin:
            return 0
        self.ops[ready:end] = block + self.ops[ready:begin]
        for i, o in enumerate(self.ops):
            if o['op'] in ('conv_pair', 'conv_bridge'):
                o['first'] = i - 2
        return begin - ready

    def stem_fast_path(self, conv_index):
        """Marks the plan's input op + the conv op ``conv_index`` (a 7x7 stride-2 pad-3 stem conv on the input tensor,
        ResNet ``body.0``, models/resnet.py:274-284) as the GENERIC alternative (alt = 1) and adds the fast alternative
        (alt = 2; csrc/stem.hip): the input converted into a padded 4-channel layout inside the same tensor's storage and
        the dedicated stem kernel writing the same destination.  The executor picks per input size.  No new state-dict
        entries: the fast conv packs the same weights differently."""
        conv = self.ops[conv_index]
        inp_i = next(i for i, o in enumerate(self.ops) if o['op'] == 'input')
        inp = self.ops[inp_i]
        assert conv['op'] == 'conv' and conv['src0'] == inp['dst'] and (conv['k'], conv['stride'], conv['pad']) == (7, 2, 3)
        inp['alt'], conv['alt'] = 1, 1
        self.ops.insert(inp_i + 1, dict(op='input_stem', dst=inp['dst'], in_channels=inp['in_channels'], alt=2))
        fast = dict(conv, op='stem7', alt=2)
        self.ops.insert(self.ops.index(conv) + 1, fast)


# ---------------------------------------------------------------------------------------------------------------------
# encoders
# ---------------------------------------------------------------------------------------------------------------------

def _two_conv_norm_relu(P, x, cout, prefix, bias=True, src1=None, up0=False, up1=False, subpixel=False):
    """TwoConvNormRelu (commons.py:120-149): Sequential(conv, bn, relu, conv, bn, relu) -> indices 0,1,3,4.
    ``subpixel``: the first conv reads cat(x, nearest-upsampled src1) -- emit it as a sub-pixel triple (see
    include/cpn_hip.h, CPN_SUBPIXEL_*): the conv itself (runs at sizes where the upsampling is not an exact x2), then
    its decomposition into four 2 x 2 phase convs on the low-resolution map + the lateral conv with the pixel-shuffled
    partial sums as residual (4/9 of the multiply-accumulates on the upsampled channels).  ``subpixel='triples'`` (fp8
    plans): the triples only -- the bridge level keeps the conv as stated (its scattered single-op form has no HEAD op for the
    executor's fallback sizes and the e4m3 simulator to follow)."""
    if subpixel is True and src1 is None and up0 is True:
        # bridge level: the conv's ONLY source is the x2-upsampled map (scale_factor=2: always exact) -> one op, the four
        # 2 x 2 phase convs + bias + ReLU scattered to their pixels (CPN_SUBPIXEL_SCATTER)
        x = P.conv(x, cout, 2, w=prefix + '0.', bn=prefix + '1.', bias=bias, act='relu', pad=1, sub=('scatter', 0))
        x = P.conv(x, cout, 3, w=prefix + '3.', bn=prefix + '4.', bias=bias, act='relu')
        P.conv_bridge()  # both convs as one kernel where the output holds a 16 x 32 tile (64-channel bridges: the ResNet-UNets)
        return x
    x = _first_conv(P, x, cout, prefix + '0.', prefix + '1.', bias, src1, up0, up1, subpixel)
    return P.conv(x, cout, 3, w=prefix + '3.', bn=prefix + '4.', bias=bias, act='relu')


def _first_conv(P, lat, cout, w, bn, bias, src1, up0, up1, subpixel):
    """conv 3x3 + BN + ReLU over cat(lat, nearest-upsampled src1) (or over ``lat`` alone), as a sub-pixel triple where asked."""
    sub = bool(subpixel) and src1 is not None and up1 and not up0
    x = P.conv(lat, cout, 3, w=w, bn=bn, bias=bias, act='relu', src1=src1, up0=up0, up1=up1, sub='head' if sub else None)
    if sub:
        c0 = P.tensors[lat]['c']
        ph = P.conv(src1, cout, 2, w=w, bn=bn, bias=bias, pad=1, sub=('phase', c0))
        P.conv(lat, cout, 3, w=w, bn=bn, bias=bias, act='relu', res=ph, res_up='shuffle', sub=('lateral', c0), dst=x)
    return x


def _cd_res_block(P, x, cin, cout, prefix, src1=None, up1=False, subpixel=False):
    """celldetection ``ResBlock`` (models/commons.py:259-359), the block class of ResUNet (unet.py:434-464), stride 1:
    conv3x3(no bias)-BN-ReLU-conv3x3(no bias)-BN + identity, ReLU; the identity is a 1x1 ConvNorm (no bias) when the channel
    counts differ -- over the same virtual concat [x | upsampled src1] in the decoder.  State-dict order: downsample.*, block.*."""
    if cin != cout:
        idt = P.conv(x, cout, 1, w=prefix + 'downsample.0.', bn=prefix + 'downsample.1.', src1=src1, up1=up1)
    elif src1 is None:
        idt = x
    else:
        raise NotImplementedError('ResBlock over a concat with as many input as output channels (identity shortcut over a cat)')
    t = _first_conv(P, x, cout, prefix + 'block.0.', prefix + 'block.1.', False, src1, False, up1, subpixel)
    return P.conv(t, cout, 3, w=prefix + 'block.3.', bn=prefix + 'block.4.', res=idt, act='relu')


# constructor options of the reference backbones that the HIP graph does not model: accepted only with the listed
# (default) values -- anything else must fail loudly instead of silently building a different network than the one the
# checkpoint was trained with (e.g. inputs_mean / inputs_std: Normalize has no parameters, so a state dict cannot tell)
_DEFAULT_ONLY = dict(pretrained=(False, None), inputs_mean=(0., None), inputs_std=(1., None), fused_initial=(False,),
                     interpolate=('nearest',), nd=(2,), block_cls=(None,), secondary_block=(None,), pool=(True,),
                     block=(None,), block_kwargs=(None, {}), bridge_strides=(True,), bridge_block_cls=(None,),
                     bridge_block_kwargs=(None, {}), block_interpolate=(False,), bridge_block_interpolate=(False,),
                     extra_blocks=(None,), norm_layer=(None, 'BatchNorm2d', 'batchnorm2d'), activation=('relu', 'ReLU'),
                     cat_order=(0,), final_activation=(None,), final_interpolate=('nearest',), anchor=(None,),
                     anchor_kwargs=(None, {}), kernel_size=(3,), padding=(1,), stride=(1,), out_layer=(False, None),
                     keep_features=(False,), groups=(None,), width_per_group=(None,), replace_stride_with_dilation=(None,),
                     layers=(None,), final_layer=(None,), final_pool=(None,))


def _check_kwargs(kw: dict, allowed: tuple, where: str):
    for k, v in kw.items():
        if k in allowed:
            continue
        ok = _DEFAULT_ONLY.get(k)
        if ok is not None and any((v is o) or (type(v) is type(o) and v == o) or
                                  (isinstance(v, (int, float)) and isinstance(o, (int, float)) and
                                   not isinstance(v, bool) and not isinstance(o, bool) and v == o)
                                  or (isinstance(v, (tuple, list)) and isinstance(o, (int, float)) and
                                      all(float(i) == float(o) for i in v)) for o in ok):
            continue
        raise NotImplementedError(f'{where}: option {k}={v!r} is not supported by the HIP engine '
                                  f'(supported: {sorted(allowed)}; default-valued reference options are accepted)')


def _unet_encoder(P, x, in_channels, prefix, depth=5, base_channels=64, factor=2, res_blocks=False, **unused):
    """UNetEncoder (unet.py:29-58) with pool=True; TwoConvNormRelu blocks (U22 family) or ResBlocks (ResUNet)."""
    _check_kwargs(unused, (), 'UNetEncoder')
    feats, channels = [], []
    in_c = in_channels
    for i in range(depth):
        out_c = base_channels * (factor ** i)
        if i > 0:
            x = P.maxpool(x, 2, 2, 0)
        bp = f'{prefix}0.' if i == 0 else f'{prefix}{i}.1.'
        x = _cd_res_block(P, x, in_c, out_c, bp) if res_blocks else _two_conv_norm_relu(P, x, out_c, bp)
        feats.append(x)
        channels.append(out_c)
        in_c = out_c
    return feats, channels, [2 ** i for i in range(depth)]


_RESNETS = {
    # name: (block, layers, groups, base_width)
    'ResNet18': ('basic', (2, 2, 2, 2), 1, 64), 'ResNet34': ('basic', (3, 4, 6, 3), 1, 64),
    'ResNet50': ('bottle', (3, 4, 6, 3), 1, 64), 'ResNet101': ('bottle', (3, 4, 23, 3), 1, 64),
    'ResNet152': ('bottle', (3, 8, 36, 3), 1, 64),
    'ResNeXt50': ('bottle', (3, 4, 6, 3), 32, 4), 'ResNeXt101': ('bottle', (3, 4, 23, 3), 32, 8),
    'ResNeXt152': ('bottle', (3, 8, 36, 3), 32, 8),
    'WideResNet50': ('bottle', (3, 4, 6, 3), 1, 128), 'WideResNet101': ('bottle', (3, 4, 23, 3), 1, 128),
}


def _resnet(P, x, in_channels, prefix, kind, base_channel=64, stem_fast=False, fuse_blocks=False, **unused):
    """ResNet(fused_initial=False) (resnet.py:265-297): body.0 = conv7x7 s2 + BN + ReLU (feature '0'),
    body.1 = Sequential(MaxPool(3,2,1), layer1), body.2..4 = layer2..4; blocks per torchvision forward."""
    _check_kwargs(unused, (), kind)
    block, layers, groups, base_width = _RESNETS[kind]
    bc = base_channel
    xin = x
    x = P.conv(x, bc, 7, w=prefix + '0.0.', bn=prefix + '0.1.', stride=2, pad=3, act='relu')
    if stem_fast and P.tensors[xin]['c'] <= 4 and _pad32(bc) in (32, 64) and P.ops[-2]['op'] == 'input':
        P.stem_fast_path(len(P.ops) - 1)
    feats, channels = [x], [bc]
    x = P.maxpool(x, 3, 2, 1)
    inplanes = bc
    expansion = 4 if block == 'bottle' else 1
    for si in range(4):
        planes = bc * (2 ** si)  # oc[si] // expansion
        stage_prefix = f'{prefix}1.1.' if si == 0 else f'{prefix}{si + 1}.'
        for j in range(layers[si]):
            stride = 2 if (j == 0 and si > 0) else 1
            p = f'{stage_prefix}{j}.'
            has_ds = j == 0 and (stride != 1 or inplanes != planes * expansion)
            if block == 'bottle':
                width = int(planes * (base_width / 64.0)) * groups
                t = P.conv(x, width, 1, w=p + 'conv1.', bn=p + 'bn1.', act='relu')
                t = P.conv(t, width, 3, w=p + 'conv2.', bn=p + 'bn2.', stride=stride, groups=groups, act='relu')
                if fuse_blocks and groups > 1:
                    P.conv_pair()  # (bf16 plans) conv1 -> grouped conv2 as one kernel where the feature map allows it
                # key order in the reference: conv3, bn3, then downsample -> emit conv3 keys before downsample keys
                n_before = len(P.entries)
                out_c = planes * expansion
                if has_ds:
                    # IR order: downsample must exist before conv3 (residual); entries order fixed afterwards
                    idt = P.conv(x, out_c, 1, w=p + 'downsample.0.', bn=p + 'downsample.1.', stride=stride)
                    ds_entries = P.entries[n_before:]
                    del P.entries[n_before:]
                else:
                    idt, ds_entries = x, []
                x = P.conv(t, out_c, 1, w=p + 'conv3.', bn=p + 'bn3.', res=idt, act='relu')
                P.entries += ds_entries
            else:
                out_c = planes
                n_before = len(P.entries)
                if has_ds:
                    idt = P.conv(x, out_c, 1, w=p + 'downsample.0.', bn=p + 'downsample.1.', stride=stride)
                    ds_entries = P.entries[n_before:]
                    del P.entries[n_before:]
                else:
                    idt, ds_entries = x, []
                t = P.conv(x, planes, 3, w=p + 'conv1.', bn=p + 'bn1.', stride=stride, act='relu')
                x = P.conv(t, planes, 3, w=p + 'conv2.', bn=p + 'bn2.', res=idt, act='relu')
                P.entries += ds_entries
            inplanes = planes * expansion
        feats.append(x)
        channels.append(inplanes)
    return feats, channels, [2, 4, 8, 16, 32]


# ---------------------------------------------------------------------------------------------------------------------
# decoders
# ---------------------------------------------------------------------------------------------------------------------

def _generalized_unet(P, feats, channels, strides, prefix, subpixel=False, res_blocks=False):
    """GeneralizedUNet (unet.py:62-249), default kwargs: nearest interpolation, cat_order 0, TwoConvNormRelu blocks,
    bridge blocks (bias=False) for the log2(first stride) missing levels.  The 1x1 ``inner`` conv is applied BEFORE
    the nearest upsample (bit-identical per pixel, 4x fewer MACs); the upsample itself and the channel concat are
    folded into the consumer conv's loader (up flag + two sources)."""
    bridges = int(math.log2(strides[0]))
    in_list = [0] * bridges + list(channels)
    out_list = list(channels)  # out_channels_list is NOT extended for one bridge (unet.py:100-107)
    if len(out_list) < len(channels) + bridges - 1:
        out_list = [out_list[0]] * (len(channels) + bridges - 1 - len(out_list)) + out_list
    n = len(in_list)
    # entries: all inner_blocks first, then layer_blocks (torchvision FPN registers inner_blocks before layer_blocks)
    inner = {}
    for i in range(1, n):
        ouc = out_list[i - 1]
        inc = out_list[i] if i < n - 1 else in_list[i]
        inner[i - 1] = (inc, ouc) if (inc > 0 and ouc < inc) else None
    depth = n - 1
    last = feats[-1]
    results = {}
    entries_inner, entries_layer = {}, {}
    for i in range(depth - 1, -1, -1):
        lat = feats[i - bridges] if (i - bridges) >= 0 else None
        top = last
        mark = len(P.entries)
        if inner[i] is not None:
            top = P.conv(top, inner[i][1], 1, w=f'{prefix}inner_blocks.{i}.', bias=True)
        entries_inner[i] = P.entries[mark:]
        del P.entries[mark:]
        ouc = out_list[i]
        if lat is not None and res_blocks:
            last = _cd_res_block(P, lat, P.tensors[lat]['c'] + P.tensors[top]['c'], ouc, f'{prefix}layer_blocks.{i}.', src1=top,
                                 up1=True, subpixel=subpixel)
        elif lat is not None:
            last = _two_conv_norm_relu(P, lat, ouc, f'{prefix}layer_blocks.{i}.', bias=True, src1=top, up1=True,
                                       subpixel=subpixel)
        else:
            last = _two_conv_norm_relu(P, top, ouc, f'{prefix}layer_blocks.{i}.', bias=False, up0=True, subpixel=subpixel)
        entries_layer[i] = P.entries[mark:]
        del P.entries[mark:]
        results[i] = last
    results[depth] = feats[-1]  # the dict of the reference ends with the deepest encoder feature itself (unet.py:207-249)
    out_list = list(out_list) + [channels[-1]] if len(out_list) <= depth else list(out_list)
    for i in sorted(entries_inner):
        P.entries += entries_inner[i]
    for i in sorted(entries_layer):
        P.entries += entries_layer[i]
    return results, out_list


def _fpn(P, feats, channels, prefix, fpn_channels, live=(0, 1)):
    """FeaturePyramidNetwork (fpn.py:79-134; forward = torchvision): ConvNorm(norm=Identity) 1x1 laterals with bias,
    top-down nearest upsample + add (fused as an upsampled residual), 3x3 output convs.  Levels whose outputs the CPN
    never reads (layer_blocks 2..4, 'pool') are skipped but their parameters stay in the state dict."""
    n = len(feats)
    lat_entries, out_entries = {}, {}
    last = None
    outs = {}
    for idx in range(n - 1, -1, -1):
        mark = len(P.entries)
        last = P.conv(feats[idx], fpn_channels, 1, w=f'{prefix}inner_blocks.{idx}.0.', bias=True,
                      res=last, res_up=last is not None)
        lat_entries[idx] = P.entries[mark:]
        del P.entries[mark:]
        if idx in live:
            outs[idx] = P.conv(last, fpn_channels, 3, w=f'{prefix}layer_blocks.{idx}.0.', bias=True)
        else:  # dead level: keep the parameters only
            P.conv_keys(f'{prefix}layer_blocks.{idx}.0.', fpn_channels, fpn_channels, 3, True)
        out_entries[idx] = P.entries[mark:]
        del P.entries[mark:]
    for idx in range(n):
        P.entries += lat_entries[idx]
    for idx in range(n):
        P.entries += out_entries[idx]
    return outs


# ---------------------------------------------------------------------------------------------------------------------
# full CPN
# ---------------------------------------------------------------------------------------------------------------------

FUSE_READOUT = True  # fuse the ReadOut tail (1x1 conv + activation) into the kxk head conv kernel

BACKBONES = {}
for _k in _RESNETS:
    BACKBONES[f'{_k}UNet'] = ('unet', _k)
    BACKBONES[f'{_k}FPN'] = ('fpn', _k)
BACKBONES['U22'] = ('unet', 'U22')
# the other UNetEncoder-based U-Nets of models/unet.py:434-524: (fixed base_channels or None, ResBlocks?)
_UNET_ENCODERS = {'U22': (None, False), 'SlimU22': (32, False), 'WideU22': (128, False), 'ResUNet': (None, True)}
for _k in ('SlimU22', 'WideU22', 'ResUNet'):
    BACKBONES[_k] = ('unet', _k)


def _readout(P, x, cmid, cout, prefix, act, act_scale, out_index, k=7, fuse=True, up0=False, stride=1, deferred=False,
             bilinear_phases=False, bl_source=None, hidden='relu'):
    """ReadOut (commons.py:461-511): conv kxk(bias) -> BN -> ``hidden`` activation (default ReLU) -> Dropout2d(eval: identity) ->
    conv 1x1(bias).
    ``bilinear_phases`` (fused heads over a bilinear source, k = 3 (mod 4)): the op additionally carries its decomposition for
    the exact x2 case (include/cpn_hip.h CPN_SUBPIXEL_BL_*): four k2 x k2 phase convs on the low-resolution map + the same
    conv restricted to the image frame.  ``bl_source`` (fp8 plans, whose resize is an op of its own): the tensor in front of
    that resize -- the phase convs read it, head and frame conv read ``x``, the materialised resized map."""
    if stride > 2:
        # the conv kernel's k x k strides are 1 and 2.  A pointwise conv commutes with subsampling, so ReadOut at stride s = the k x k
        # conv at stride 2 (+ BN + activation) followed by the 1x1 conv AT STRIDE s / 2: the same taps and sums per output pixel
        # (floor(floor(a / 2) / (s / 2)) = floor(a / s): the same output size), 4 / s^2 of the hidden pixels are computed in vain
        assert not deferred and up0 != 'bilinear'
        t = P.conv(x, cmid, k, w=prefix + 'block.0.', bn=prefix + 'block.1.', bias=True, act='relu' if hidden == 'relu' else 'none',
                   up0=up0, stride=2)
        if hidden not in ('relu', 'none'):
            t = P.act(t, hidden)
        P.conv(t, cout, 1, w=prefix + 'block.4.', bias=True, act=act, act_scale=act_scale, out_index=out_index, stride=stride // 2,
               pad=0)
        return
    if hidden != 'relu':
        # any other hidden activation: conv k x k + BN (no activation, NHWC) -> CPN_OP_ACT -> conv 1x1 + final activation.  The
        # heads' fused forms (ReadOut tail in the conv kernel, score gate, bilinear phases) are ReLU-only: the libm code of the
        # other activations lives in three small elementwise kernels instead of every conv epilogue
        assert not deferred, 'only fused ReLU ReadOut heads can be deferred'
        t = P.conv(x, cmid, k, w=prefix + 'block.0.', bn=prefix + 'block.1.', bias=True, act='none', up0=up0, stride=stride)
        if hidden != 'none':
            t = P.act(t, hidden)
        P.conv(t, cout, 1, w=prefix + 'block.4.', bias=True, act=act, act_scale=act_scale, out_index=out_index)
        return
    if fuse and FUSE_READOUT and _pad32(cmid) in (32, 64, 128, 256) and cout <= 32:
        # one kernel: conv kxk + BN + ReLU -> (bf16, LDS) -> 1x1 conv + final activation -> fp32 NCHW head map
        kw = dict(w=prefix + 'block.0.', bn=prefix + 'block.1.', bias=True, act='relu', out_index=out_index,
                  fuse=dict(w=prefix + 'block.4.', cout=cout, act=act, act_scale=act_scale))
        triple = bool(bilinear_phases) and (up0 == 'bilinear' or bl_source is not None) and k % 4 == 3 and stride == 1 and \
            not deferred
        P.conv(x, cmid, k, up0=up0, stride=stride, deferred=deferred, sub='blhead' if triple else None, **kw)
        if triple:
            P.conv(x if bl_source is None else bl_source, cmid, (k + 3) // 2, sub=('blphase', k), **kw)
            P.conv(x, cmid, k, up0=up0, sub=('blframe', k), **kw)
        return
    assert not deferred, 'only fused ReadOut heads can be deferred'
    t = P.conv(x, cmid, k, w=prefix + 'block.0.', bn=prefix + 'block.1.', bias=True, act='relu', up0=up0, stride=stride)
    P.conv(t, cout, 1, w=prefix + 'block.4.', bias=True, act=act, act_scale=act_scale, out_index=out_index)


def build_plan(backbone: str, in_channels: int, order: int = 5, score_channels: int = 1, refinement: bool = True,
               refinement_margin: float = 3., refinement_buckets: int = 1, order_weights: bool = True,
               backbone_kwargs: dict = None, fuse_readout: bool = True, uncertainty_head: bool = False,
               contour_head_channels: int = None, refinement_head_channels: int = None,
               kernel_sizes: dict = None, fuse_bilinear: bool = True, contour_head_stride: int = 1,
               refinement_head_stride: int = 1, features: dict = None, sparse_heads: bool = False,
               subpixel: bool = False, stem_fast: bool = False, fuse_blocks: bool = False, hoist_heads: bool = True,
               bilinear_phases: bool = False, head_activations: dict = None, refinement_full_res: bool = True,
               refinement_interpolation: str = 'bilinear', fuse_kwargs: dict = None) -> Plan:
    """Plan of ``Cpn<backbone>`` (celldetection/models/cpn.py:287-439,771-2061; heads: CPNCore.__init__
    cpn.py:125-236).  ``kernel_sizes``: optional {'score'|'location'|'fourier'|'uncertainty'|'refinement': k}
    (the reference's ``kernel_size_<head>`` kwargs, default 7).  ``contour_head_stride`` / ``refinement_head_stride``: stride
    (1 or 2) of the k x k conv of the ReadOut heads (commons.py:494).  ``features``: optional {'score'|'location'|
    'contour'|'uncertainty'|'refinement': key or [key, key]} = the reference's ``<head>_features`` kwargs (cpn.py:135-139):
    decoder level '0', '1', ... or 'encoder.<k>' (UNet family); two or three keys are fused like ``Fuse2d``
    (commons.py:640-674: the other features nearest-resized to the first one's size, concat, 1x1 conv + BN + ReLU).
    ``sparse_heads``: score-gated location / Fourier heads -- CPN.forward reads their maps at the proposal pixels only
    (cpn.py:613-637), so the two convs are packed but not executed by the graph (``deferred`` ops) and evaluated at the
    proposals by ``ops.sparse_heads``; needs both heads fused, on the same plain feature, stride 1, same kernel size
    and a hidden width of 128 or 256 (``Plan.meta['sparse_heads']`` = None when the plan does not qualify).
    ``subpixel`` (bf16 plans; ``'triples'`` in fp8 plans): the first conv of every UNet decoder level (models/unet.py:213-224)
    additionally carries its sub-pixel decomposition; the executor picks it wherever the top-down map is upsampled by exactly 2.
    In an fp8 plan the partial sums between the phase convs and the lateral conv travel as bf16.
    ``stem_fast`` (bf16 plans, ResNet-family encoders with <= 4 input channels): the 7x7 stride-2 stem additionally carries
    its dedicated kernel on a padded 4-channel input layout (``Plan.stem_fast_path``).
    ``fuse_blocks`` (bf16 plans, ResNeXt encoders): every bottleneck block additionally carries conv1 -> grouped conv2 as
    one fused op (``Plan.conv_pair``).
    ``hoist_heads``: the score / location / Fourier / uncertainty head ops are placed right behind the op that completes their
    input feature (``Plan.hoist``) instead of behind the whole backbone (CPNCore.forward, cpn.py:238-283, runs the backbone
    first): same ops, same results, the heads of the UNet models no longer wait for decoder level 0.
    ``bilinear_phases`` (bf16 / fp8 plans): the refinement head over the bilinear-resized feature map (FPN models) additionally
    carries its sub-pixel decomposition (``_readout``).  ``head_activations``: optional {'score'|'location'|'fourier'|
    'uncertainty'|'refinement': plan activation name} = the reference's ``head_activation`` / ``head_activation_<head>`` kwargs
    (cpn.py:183-233; default 'relu'; see ``head_activation_name``).  ``refinement_full_res=False``: the refinement head reads its
    feature at the feature's resolution (cpn.py:276-279).  ``refinement_interpolation``: 'bilinear' | 'bicubic' -- the mode of that resize
    (the non-interpolating modes raise in the reference as soon as a resize is needed, see cpn.CPN._check_refinement_interpolation).
    ``fuse_kwargs`` (cpn.py:173 -> Fuse2d, commons.py:640-674): ``activation`` (a torch.nn name as for the heads, or None),
    ``norm_layer`` (None | 'batchnorm2d'), ``bias``, and -- for heads fused over TWO features -- ``kernel_size`` k with ``padding``
    k // 2 (a k x k conv over the virtual concat; over more features only the 1x1 conv commutes with the nearest resize)."""
    fkw = dict(fuse_kwargs or {})
    f_act = fkw.pop('activation', 'relu')
    f_act = 'none' if f_act is None else head_activation_name(f_act)
    f_norm = fkw.pop('norm_layer', 'batchnorm2d')
    if f_norm is not None and str(getattr(f_norm, '__name__', f_norm)).lower() != 'batchnorm2d':
        raise NotImplementedError(f'fuse_kwargs: norm_layer {f_norm!r} is not supported by the HIP engine (None | batchnorm2d)')
    f_k, f_bias = int(fkw.pop('kernel_size', 1)), bool(fkw.pop('bias', True))
    f_pad = int(fkw.pop('padding', 0))
    if f_k % 2 == 0 or f_pad != f_k // 2:
        raise NotImplementedError('fuse_kwargs: kernel_size must be odd with padding = kernel_size // 2 on the HIP path')
    if fkw:
        raise NotImplementedError(f'fuse_kwargs: options {sorted(fkw)} are not supported by the HIP engine')
    ha = dict(score='relu', location='relu', fourier='relu', uncertainty='relu', refinement='relu')
    ha.update(head_activations or {})
    if any(v not in _ACT or v == 'tanh_scaled' for v in ha.values()):
        raise ValueError(f'unknown head activation in {ha}')
    if contour_head_stride not in (1, 2, 4, 8) or refinement_head_stride not in (1, 2, 4, 8):
        raise NotImplementedError('head strides other than 1, 2, 4 and 8 are not supported by the HIP engine')
    feats_cfg = dict(score='1', location='1', contour='1', uncertainty='1', refinement='0')
    feats_cfg.update(features or {})
    if backbone not in BACKBONES:
        raise ValueError(f'Unsupported backbone {backbone!r}; supported: {sorted(BACKBONES)}')
    if score_channels < 1 or refinement_buckets < 1:
        raise ValueError('score_channels and refinement_buckets must be >= 1')
    ks = dict(kernel_sizes or {})
    family, enc = BACKBONES[backbone]
    bkw = dict(backbone_kwargs or {})
    _check_kwargs(bkw, ('backbone_kwargs',) + (('fpn_channels',) if family == 'fpn' else ()), backbone)
    P = Plan()
    if order_weights:
        P.entries.append(('order_weights', (order, 1), 'buffer'))
    x = P.input(in_channels)
    bp = 'core.backbone.'
    res_blocks = False
    if enc in _UNET_ENCODERS:
        ekw = dict(bkw.get('backbone_kwargs') or {})
        base, res_blocks = _UNET_ENCODERS[enc]
        if base is not None:
            if 'base_channels' in ekw:  # SlimU22 / WideU22 pass base_channels themselves (unet.py:490,520)
                raise TypeError(f"{enc}: got multiple values for keyword argument 'base_channels'")
            ekw['base_channels'] = base
        feats, channels, strides = _unet_encoder(P, x, in_channels, bp + 'body.', res_blocks=res_blocks, **ekw)
    else:
        ekw = dict(bkw.get('backbone_kwargs') or {})
        feats, channels, strides = _resnet(P, x, in_channels, bp + 'body.', enc, stem_fast=bool(stem_fast),
                                           fuse_blocks=bool(fuse_blocks), **ekw)
    def _keys(v):
        return [str(k) for k in v] if isinstance(v, (list, tuple)) else [str(v)]

    wanted = {k for name, v in feats_cfg.items() for k in _keys(v)
              if (name != 'uncertainty' or uncertainty_head) and (name != 'refinement' or refinement)}
    if family == 'unet':
        results, out_list = _generalized_unet(P, feats, channels, strides, bp + 'unet.', subpixel=subpixel, res_blocks=res_blocks)
        level = {str(i): (results[i], out_list[i]) for i in results}
        level.update({f'encoder.{i}': (feats[i], channels[i]) for i in range(len(feats))})
    else:
        fc = bkw.get('fpn_channels', 256)
        live = sorted({int(k) for k in wanted if k.isdigit()} | {0, 1})
        if any(i >= len(feats) for i in live):
            raise ValueError(f'FPN feature keys must be < {len(feats)}: {sorted(wanted)}')
        outs = _fpn(P, feats, channels, bp + 'fpn.', fc, live=live)
        level = {str(i): (outs[i], fc) for i in outs}
    for k in wanted:
        if k not in level:
            raise NotImplementedError(f'feature key {k!r} is not available on the HIP path for {backbone} '
                                      f'(available: {sorted(level)})')

    def _head_input(name, fuse_prefix):
        """-> (tensor id, channels) of a head's input; two keys: Fuse2d = 1x1 conv + BN + ReLU over the virtual concat
        [first | second nearest-resized to the first's size] (channels_out = channels of the first key, cpn.py:88-100)."""
        keys = _keys(feats_cfg[name])
        (t0, ch0) = level[keys[0]]
        if len(keys) == 1:
            return t0, ch0
        ts = [level[k] for k in keys]
        (t1, ch1) = ts[1]
        w_, bn_ = fuse_prefix + 'block.0.', (fuse_prefix + 'block.1.' if f_norm is not None else None)
        conv_act = f_act if f_act in ('relu', 'none') else 'none'  # (other activations: an elementwise op behind the conv)

        def finish(t_):
            return t_ if conv_act == f_act else P.act(t_, f_act)

        if len(keys) == 2:
            t = P.conv(t0, ch0, f_k, w=w_, bn=bn_, bias=f_bias, act=conv_act, src1=t1, up1=True)
            return finish(t), ch0
        if f_k != 1:
            raise NotImplementedError('fuse_kwargs: kernel_size > 1 over three or more features (the split into per-feature '
                                      'convs needs a conv that commutes with the nearest resize: 1x1 only)')
        # three features: conv1x1(cat(f0, f1^, f2^)) = conv1x1 over [f0 | f1^] + (conv1x1 over f2)^ -- a 1x1 conv commutes with
        # the nearest resize, so the third feature's share runs at ITS resolution (no bias, BN scale folded) and joins as a
        # nearest-resized residual in front of bias + ReLU (the conv kernel reads two concat sources and one residual).
        # More features: the sum continues in steps of two -- [running sum | f_i^] (identity block over the running sum, the
        # share of f_i) + (conv1x1 over f_i+1)^ -- every feature is resized from ITS size to the first one's in one step, as
        # Fuse2d does (a chain of resizes would pick other pixels at odd sizes); bias + ReLU close the last step
        off = [0]
        for (_, c) in ts:
            off.append(off[-1] + c)
        P.conv_keys(w_, ch0, off[-1], 1, f_bias)
        if bn_ is not None:
            P.bn_keys(bn_, ch0)
        n = len(ts)

        def part(i):
            return P.conv(ts[i][0], ch0, 1, w=w_, bn=bn_, bias=f_bias, share=(off[i], off[i + 1], False))

        res = part(2)
        t = P.conv(t0, ch0, 1, w=w_, bn=bn_, bias=f_bias, act=conv_act if n == 3 else 'none', src1=t1, up1=True, res=res,
                   res_up=True, share=(0, off[2], n == 3))
        for i in range(3, n, 2):
            res = part(i + 1) if i + 1 < n else None
            last = i + 2 >= n
            t = P.conv(t, ch0, 1, w=w_, bn=bn_, bias=f_bias, act=conv_act if last else 'none', src1=ts[i][0], up1=True, res=res,
                       res_up=res is not None, share=(off[i], off[i + 1], last, ch0))
        return finish(t), ch0

    heads_begin = len(P.ops)
    f1s, c1 = _head_input('score', 'core.score_fuse.')
    scale = P.tensors[f1s]['down'] * contour_head_stride
    cm1 = None if contour_head_channels is None else int(contour_head_channels)
    cm0 = None if refinement_head_channels is None else int(refinement_head_channels)
    # binary: sigmoid fused into the head; multi-class: raw logits, softmax/argmax in cpn_class_scores (cpn.py:583-585)
    hs = contour_head_stride
    _readout(P, f1s, cm1 or c1, score_channels, 'core.score_head.', 'sigmoid' if score_channels == 1 else 'none', 0.,
             _lib.OUT_SCORES, k=ks.get('score', 7), fuse=fuse_readout, stride=hs, hidden=ha['score'])
    fl, cl = _head_input('location', 'core.location_fuse.')
    # score-gated heads: both heads on the same single feature key (then `_head_input` adds no op and the two share fl)
    same_src = _keys(feats_cfg['contour']) == _keys(feats_cfg['location']) and len(_keys(feats_cfg['location'])) == 1
    gate = bool(sparse_heads) and fuse_readout and FUSE_READOUT and same_src and hs == 1 and order * 4 <= 32 and \
        _pad32(cm1 or cl) in (128, 256) and ks.get('location', 7) == ks.get('fourier', 7) and ks.get('location', 7) > 1 and \
        ha['location'] == ha['fourier'] == 'relu'  # (the gathered kernel's hidden activation is ReLU)
    first_head_op = len(P.ops)
    _readout(P, fl, cm1 or cl, 2, 'core.location_head.', 'none', 0., _lib.OUT_LOCATIONS, k=ks.get('location', 7),
             fuse=fuse_readout, stride=hs, deferred=gate, hidden=ha['location'])
    ff, cf = _head_input('contour', 'core.fourier_fuse.')
    _readout(P, ff, cm1 or cf, order * 4, 'core.fourier_head.', 'none', 0., _lib.OUT_FOURIER, k=ks.get('fourier', 7),
             fuse=fuse_readout, stride=hs, deferred=gate, hidden=ha['fourier'])
    assert not gate or (ff == fl and cf == cl)
    sparse_meta = dict(ops=(first_head_op, first_head_op + 1), src=fl) if gate else None
    if uncertainty_head:  # cpn.py:209-221: 4 channels, sigmoid
        fu, cu = _head_input('uncertainty', 'core.uncertainty_fuse.')
        _readout(P, fu, cm1 or cu, 4, 'core.uncertainty_head.', 'sigmoid', 0., _lib.OUT_UNCERTAINTY,
                 k=ks.get('uncertainty', 7), fuse=fuse_readout, stride=hs, hidden=ha['uncertainty'])
    heads_end = len(P.ops)
    if refinement:
        r, c0 = _head_input('refinement', 'core.refinement_fuse.')
        cm0 = cm0 or c0
        # cpn.py:277-278: bilinear resize of the features to the input size.  FPN: always (f0 lives at stride 2);
        # ResNet + UNet: the bridge level is 2 * ceil(H / 2) pixels high, i.e. H + 1 for odd H (no-op alias otherwise);
        # U22: level 0 has the input size by construction (3x3 convs, padding 1)
        # bf16 / fp8 plans: the resize is fused into the head conv's halo loader (up0 = 'bilinear': the full-resolution
        # 256-channel map of the FPN models is never written); the fp32 verification plan keeps the separate op
        # refinement_full_res=False (cpn.py:276-279): the head runs at the feature's own resolution, its 2 * buckets output
        # maps are resized to the input size instead (the engine's fp32 bilinear kernel, cpn._Engine.run)
        resize = (family == 'fpn' or enc not in _UNET_ENCODERS or _keys(feats_cfg['refinement']) != ['0']) and refinement_full_res
        kr = ks.get('refinement', 7)
        bicubic = refinement_interpolation == 'bicubic'  # (its own resize op in every plan: the fused loader / phases are bilinear)
        fused_resize = resize and fuse_bilinear and kr > 1 and refinement_head_stride == 1 and not bicubic
        bilinear_phases = bilinear_phases and not bicubic
        r_low = None
        if resize and not fused_resize:
            r_low = r
            r = P.bilinear_to_input(r, 'bicubic' if bicubic else 'bilinear')
            resize_op = P.ops[-1]
        _readout(P, r, cm0, 2 * refinement_buckets, 'core.refinement_head.', 'tanh_scaled', float(refinement_margin),
                 _lib.OUT_REFINEMENT, k=kr, fuse=fuse_readout, up0='bilinear' if fused_resize else False,
                 stride=refinement_head_stride, bilinear_phases=bilinear_phases and (fused_resize or r_low is not None),
                 bl_source=r_low if (bilinear_phases and kr > 1 and refinement_head_stride == 1) else None,
                 hidden=ha['refinement'])
        if r_low is not None and any(op.get('sub') == 'blhead' for op in P.ops[-3:]):
            resize_op['ring_for_bl'] = True  # (the executor writes only the frame's neighbourhood of that map when the phases run)
    # the score / location / Fourier (/ uncertainty) heads run as soon as their features exist (CPN_HOIST=0: the reference's order)
    if hoist_heads and os.environ.get('CPN_HOIST', '1') != '0':
        shift = P.hoist(heads_begin, heads_end)
        if sparse_meta is not None:
            sparse_meta['ops'] = tuple(i - shift for i in sparse_meta['ops'])
    P.meta = dict(backbone=backbone, order=order, head_down=scale, refinement=refinement, in_channels=in_channels,
                  score_channels=score_channels, refinement_buckets=refinement_buckets,
                  uncertainty_head=bool(uncertainty_head), sparse_heads=sparse_meta,
                  refinement_interpolation=refinement_interpolation)
    return P


# ---------------------------------------------------------------------------------------------------------------------
# packing
# ---------------------------------------------------------------------------------------------------------------------
_ACT = {'none': _lib.ACT_NONE, 'relu': _lib.ACT_RELU, 'sigmoid': _lib.ACT_SIGMOID, 'tanh_scaled': _lib.ACT_TANH_SCALED,
        'leaky_relu': _lib.ACT_LEAKY_RELU, 'silu': _lib.ACT_SILU, 'gelu': _lib.ACT_GELU, 'elu': _lib.ACT_ELU, 'tanh': _lib.ACT_TANH,
        'hardswish': _lib.ACT_HARDSWISH, 'mish': _lib.ACT_MISH, 'selu': _lib.ACT_SELU, 'softplus': _lib.ACT_SOFTPLUS}
# hidden activations of the ReadOut heads (``head_activation*`` of models/cpn.py:183-233 -> ``lookup_nn(name)``: the torch.nn module
# of that name, case-insensitive, default arguments) -> activation names of the plan
HEAD_ACTIVATIONS = {'relu': 'relu', 'leakyrelu': 'leaky_relu', 'silu': 'silu', 'gelu': 'gelu', 'elu': 'elu', 'tanh': 'tanh',
                    'sigmoid': 'sigmoid', 'hardswish': 'hardswish', 'mish': 'mish', 'selu': 'selu', 'softplus': 'softplus',
                    'identity': 'none'}


def head_activation_name(value) -> str:
    """Plan activation for a ``head_activation*`` value of the reference (a torch.nn class name as ``lookup_nn`` resolves it)."""
    key = value.__name__ if isinstance(value, type) else str(value)
    key = key.lower().replace('_', '')
    if key not in HEAD_ACTIVATIONS:
        raise NotImplementedError(f'head activation {value!r} is not supported by the HIP engine '
                                  f'(supported: {sorted(HEAD_ACTIVATIONS)})')
    return HEAD_ACTIVATIONS[key]


def _fold(sd, op):
    """Conv weight/bias with eval-mode BatchNorm (eps 1e-5) folded in, float64 math."""
    w = sd[op['w'] + 'weight'].detach().double().cpu()
    cout = w.shape[0]
    b = sd[op['w'] + 'bias'].detach().double().cpu() if op['bias'] else torch.zeros(cout, dtype=torch.float64)
    if op['bn'] is not None:
        g = sd[op['bn'] + 'weight'].detach().double().cpu()
        beta = sd[op['bn'] + 'bias'].detach().double().cpu()
        mu = sd[op['bn'] + 'running_mean'].detach().double().cpu()
        var = sd[op['bn'] + 'running_var'].detach().double().cpu()
        s = g / torch.sqrt(var + 1e-5)
        w = w * s[:, None, None, None]
        b = (b - mu) * s + beta
    return w, b


def _bundle_geometry(cin, cout, groups, kc=32):
    """(bundles, cin_b, cout_b, groups_per_bundle) or None when the grouped conv must be densified
    (kc = channels per weight record: 32 bf16 | 64 fp8)."""
    if groups == 1:
        return None
    cig, cog = cin // groups, cout // groups
    if cig != cog:
        return None
    bw = cig if cig % kc == 0 else (kc if kc % cig == 0 else None)
    if bw is None or cin % bw:
        return None
    return cin // bw, bw, bw, bw // cig


def _pad64(c):
    return (c + 63) // 64 * 64


def pack(plan: Plan, state_dict, device, precision: str = 'bf16', act_scales=None, effective_weights: list = None):
    """-> (tensor_descs, op_descs, weight_blob[bf16 | f32, device], bias_blob[f32, device]).

    bf16: weights [bundle][cin_b/32][k*k (+1 zero slab if the item count is odd)][cout_b][32]; fp32 (verification path): [bundle][k*k][cin_b][cout_b].
    fp8 (e4m3, groundwork for BASELINE configs[4]): ``act_scales[tensor id]`` = value per activation code; channels
    are padded to 64; weights [bundle][cin_b/64][k*k (+1 zero slab if odd)][cout_b][64] as e4m3 codes of
    ``w * input_scale / weight_scale[cout]``; returns additionally (mult_blob[f32] = weight_scale per output channel,
    [(res_scale, out_inv_scale)] per op); the weight blob is a byte tensor.  ``effective_weights`` (tests): a list that
    receives, per conv op, the dequantised weights the fp8 kernel effectively applies to real-valued inputs
    (``[cout, cin/groups, k, k]`` float64) and the bias."""
    f32 = precision == 'fp32'
    fp8 = precision == 'fp8'
    if fp8 and act_scales is None:
        raise ValueError('fp8 packing needs the per-tensor activation scales')
    _pad = _pad64 if fp8 else _pad32
    KC = 64 if fp8 else 32  # channels per weight record
    wdt, wsz = (torch.float32, 4) if f32 else (torch.bfloat16, 2)
    if fp8:
        wsz = 1
    mparts, op_scales = [], []
    tens = (_lib.TensorDesc * len(plan.tensors))()
    for i, t in enumerate(plan.tensors):
        tens[i].channels, tens[i].down = _pad(t['c']) * t.get('phases', 1), t['down']
        # fp8 plans: the partial-sum tensor of a sub-pixel triple ([phase][c], ``phases`` == 4) is stored as bf16 -- flagged by a
        # negative scale (include/cpn_hip.h cpn_tensor_desc)
        tens[i].scale = (-1. if t.get('phases', 1) == 4 else float(act_scales[i])) if fp8 else 0.
    ops = (_lib.OpDesc * len(plan.ops))()
    wparts, bparts = [], []
    woff = boff = 0
    for i, op in enumerate(plan.ops):
        d = ops[i]
        d.src0 = d.src1 = d.res = d.dst = -1
        d.bias_offset = d.mult_offset = -1
        op_scales.append((0., 0.))
        d.alt = int(op.get('alt', 0))
        if d.alt and f32:
            raise ValueError('the stem fast path is a bf16 / fp8-plan feature')
        if op['op'] in ('input', 'input_stem'):
            d.op, d.dst, d.in_channels = (_lib.OP_INPUT if op['op'] == 'input' else _lib.OP_INPUT_STEM), op['dst'], op['in_channels']
            continue
        if op['op'] == 'stem7':
            # weights [7][cout_b][32] bf16: filter row ky, output channel, (kx 0..7, c 0..3) -- the 7 taps of a filter row
            # over a 4-channel NHWC input are 28 contiguous values; kx = 7 and c >= in_channels meet zeros
            # (fp8 plans: the stem computes in bf16 on the bf16 input, too -- only its OUTPUT is e4m3 codes of the dst
            # tensor's scale; no weight quantisation, the multiplier slots that keep bias / mult indices aligned are ones)
            w, b = _fold(state_dict, op)
            cout, cin = op['cout'], op['cin']
            coutp = _pad(cout)
            wk = torch.zeros(7, coutp, 8, 4, dtype=torch.float64)
            wk[:, :cout, :7, :cin] = w.permute(2, 0, 3, 1)  # [cout, cin, ky, kx] -> [ky, cout, kx, cin]
            bias = torch.zeros(coutp, dtype=torch.float64)
            bias[:cout] = b
            wq = wk.reshape(-1).to(torch.bfloat16)
            wparts.append(wq.view(torch.uint8) if fp8 else wq)
            bparts.append(bias.to(torch.float32))
            if fp8:
                mparts.append(torch.ones(coutp, dtype=torch.float32))
                op_scales[-1] = (0., 1. / float(act_scales[op['dst']]))
            d.op, d.src0, d.dst = _lib.OP_STEM7, op['src0'], op['dst']
            d.kh = d.kw = 7
            d.stride, d.pad, d.bundles, d.cin_b, d.cout_b = 2, 3, 1, 32, coutp
            d.weight_offset, d.bias_offset = woff, boff
            d.act, d.cout_real, d.out_index = _lib.ACT_RELU, cout, -1
            d.fuse_weight_offset = d.fuse_bias_offset = -1
            woff += wq.numel() * 2
            boff += bparts[-1].numel()
            continue
        if op['op'] == 'conv_pair':  # shares the packed weights / biases of the two convs in front of it
            if f32 or fp8:
                raise ValueError('fused bottleneck heads are a bf16-plan feature')
            c1, c2 = ops[op['first']], ops[op['first'] + 1]
            assert op['first'] == i - 2 and c2.bundles * c2.cout_b == c1.cout_b and c2.cin_b == c2.cout_b
            assert c1.dst == c2.src0 and op['w'].startswith(plan.ops[op['first']]['w']), 'conv_pair must directly follow its two convs'
            d.op, d.src0, d.dst = _lib.OP_CONV_PAIR, op['src0'], op['dst']
            d.kh = d.kw = 3
            d.stride, d.pad = c2.stride, 1
            d.bundles, d.cin_b, d.cout_b, d.c0_used = c2.bundles, c1.cin_b, c1.cout_b, c1.cin_b
            d.weight_offset, d.bias_offset = c1.weight_offset, c1.bias_offset
            d.fuse_weight_offset, d.fuse_bias_offset, d.fuse_cout = c2.weight_offset, c2.bias_offset, c2.cout_b
            d.act, d.fuse_act, d.out_index, d.cout_real = _lib.ACT_RELU, _lib.ACT_RELU, -1, c2.cout_real
            continue
        if op['op'] == 'conv_bridge':  # shares the packed weights / biases of the scatter conv and the 3x3 conv in front of it
            if f32 or fp8:
                raise ValueError('the fused bridge level is a bf16-plan feature')
            c1, c2 = ops[op['first']], ops[op['first'] + 1]
            assert op['first'] == i - 2 and c1.cout_b == c2.cin_b == c2.cout_b == 64
            assert c1.subpixel == _lib.SUBPIXEL_SCATTER and c1.dst == c2.src0 and \
                op['w'].startswith(plan.ops[op['first']]['w']), 'conv_bridge must directly follow its scatter conv + 3x3 conv'
            d.op, d.src0, d.dst, d.res = _lib.OP_CONV_BRIDGE, op['src0'], op['dst'], c2.res
            d.kh = d.kw = 3
            d.stride, d.pad, d.bundles, d.cin_b, d.cout_b, d.c0_used = 1, 1, 1, c1.cin_b, 64, c1.cin_b
            d.res_up, d.act, d.act_scale, d.out_index, d.cout_real = c2.res_up, c2.act, c2.act_scale, -1, c2.cout_real
            d.weight_offset, d.bias_offset = c1.weight_offset, c1.bias_offset
            d.fuse_weight_offset, d.fuse_bias_offset, d.fuse_cout = c2.weight_offset, c2.bias_offset, 0
            continue
        if op['op'] == 'act':
            d.op, d.src0, d.dst, d.act = _lib.OP_ACT, op['src0'], op['dst'], _ACT[op['act']]
            continue
        if op['op'] == 'maxpool':
            d.op, d.src0, d.dst = _lib.OP_MAXPOOL, op['src0'], op['dst']
            d.kh = d.kw = op['k']
            d.stride, d.pad = op['stride'], op['pad']
            continue
        if op['op'] == 'bilinear':
            d.op, d.src0, d.dst = _lib.OP_BILINEAR, op['src0'], op['dst']
            d.act = 1 if op.get('mode') == 'bicubic' else 0  # (include/cpn_hip.h: a resize op's act selects the mode)
            if d.act and fp8:
                raise NotImplementedError("refinement_interpolation='bicubic' is a bf16 / fp32-plan feature (bicubic weights are "
                                          'negative in places: the result leaves the e4m3 range of its source)')
            # feeds a bilinear sub-pixel triple: only the frame's neighbourhood of the map is needed when the phase convs run
            d.subpixel = _lib.SUBPIXEL_BL_FRAME if op.get('ring_for_bl') else _lib.SUBPIXEL_NONE
            continue
        # conv
        w, b = _fold(state_dict, op)
        k, groups, cin, cout = op['k'], op['groups'], op['cin'], op['cout']
        sub = op.get('sub')
        bl = sub == 'blhead' or (isinstance(sub, tuple) and sub[0] in ('blphase', 'blframe'))
        if sub is not None and (f32 or (fp8 and isinstance(sub, tuple) and sub[0] == 'scatter')):
            raise ValueError('sub-pixel conv triples are a bf16 / fp8-plan feature (the scattered bridge form: bf16)')
        if isinstance(sub, tuple) and sub[0] == 'lateral':  # the lateral's share of the head conv's weights (+ its bias)
            w = w[:, :sub[1]]
        if op.get('share') is not None:  # a channel range of the stated conv; its (BN-folded) bias travels with ONE of the parts
            lo, hi, with_bias = op['share'][:3]
            w = w[:, lo:hi]
            if len(op['share']) > 3:  # [running sum | feature]: identity block in front (Fuse2d over more than three features)
                eye = torch.eye(cout, op['share'][3], dtype=w.dtype)[:, :, None, None]
                w = torch.cat((eye, w), 1)
            if not with_bias:
                b = torch.zeros_like(b)
        phase = isinstance(sub, tuple) and sub[0] in ('phase', 'scatter', 'blphase')
        if isinstance(sub, tuple) and sub[0] == 'blphase':  # four k2 x k2 kernels on the low-resolution map (float64 sums)
            from .subpixel import collapse_bilinear_taps
            w = collapse_bilinear_taps(w).reshape(4, cout, cin, k, k)
        elif phase:  # four 2 x 2 kernels on the low-resolution map (tap sums in float64, rounded to bf16 once)
            from .subpixel import collapse_upsampled_taps
            w = collapse_upsampled_taps(w[:, sub[1]:]).reshape(4, cout, cin, 2, 2)
        c0 = plan.tensors[op['src0']]['c']
        c0p = _pad(c0)
        c1 = plan.tensors[op['src1']]['c'] if op['src1'] is not None else 0
        cinp = c0p + (_pad(c1) if op['src1'] is not None else 0)
        coutp = _pad(cout) if op['dst'] is not None else _pad32(cout)
        if fp8:  # fold the input scales into the weights: the MFMA then accumulates real-valued units / weight scale
            w = w.clone()
            if groups == 1 and c1:
                w[:, :c0] *= act_scales[op['src0']]
                w[:, c0:] *= act_scales[op['src1']]
            else:
                w *= act_scales[op['src0']]
        geo = _bundle_geometry(cin, cout, groups, KC)
        if phase:
            bundles, cin_b, cout_b = 4, cinp, coutp
            kk = w.shape[-1]  # 2 (nearest phases) | k2 (bilinear phases)
            dense = torch.zeros(4, coutp, cinp, kk, kk, dtype=torch.float64)
            dense[:, :cout, :cin] = w
            packed = dense.reshape(4, coutp, cinp // KC, KC, kk * kk).permute(0, 2, 4, 1, 3)
            bias = None  # (partial sums; the lateral op of the triple adds the bias)
            if fp8 and sub[0] == 'phase':  # (an all-zero bias keeps the bias / multiplier indices of the e4m3 kernel aligned)
                bias = torch.zeros(4 * coutp, dtype=torch.float64)
            if sub[0] in ('scatter', 'blphase'):  # one bias shared by the four phases
                bias = torch.zeros(coutp, dtype=torch.float64)
                bias[:cout] = b
        elif geo is None:
            dense = torch.zeros(coutp, cinp, k, k, dtype=torch.float64)
            if groups == 1:
                dense[:cout, :c0] = w[:, :c0]
                if c1:
                    dense[:cout, c0p:c0p + c1] = w[:, c0:]
            else:  # densified grouped conv (block diagonal)
                cig, cog = cin // groups, cout // groups
                for g in range(groups):
                    dense[g * cog:(g + 1) * cog, g * cig:(g + 1) * cig] = w[g * cog:(g + 1) * cog]
            bundles, cin_b, cout_b = 1, cinp, coutp
            packed = dense.reshape(1, coutp, cinp, k * k).permute(0, 3, 2, 1) if f32 else \
                dense.reshape(1, coutp, cinp // KC, KC, k * k).permute(0, 2, 4, 1, 3)
            bias = torch.zeros(coutp, dtype=torch.float64)
            bias[:cout] = b
        else:
            bundles, cin_b, cout_b, gpb = geo
            cig = cin // groups
            dense = torch.zeros(bundles, cout_b, cin_b, k, k, dtype=torch.float64)
            wg = w.reshape(bundles, gpb, cig, cig, k, k)  # [bundle, group-in-bundle, cout_g, cin_g, k, k]
            for g in range(gpb):
                dense[:, g * cig:(g + 1) * cig, g * cig:(g + 1) * cig] = wg[:, g]
            packed = dense.reshape(bundles, cout_b, cin_b, k * k).permute(0, 3, 2, 1) if f32 else \
                dense.reshape(bundles, cout_b, cin_b // KC, KC, k * k).permute(0, 2, 4, 1, 3)
            bias = b.clone()
        if fp8:
            packed = packed.contiguous().reshape(bundles, -1, cout_b, KC)  # [bundle][item][cout][64]
            wscale = (packed.abs().amax((1, 3)) / 448.).clamp_min(1e-30)      # [bundle][cout]
            if isinstance(sub, tuple) and sub[0] == 'blphase':  # the four phases share ONE bias and ONE multiplier per channel
                wscale = wscale.amax(0, keepdim=True).expand(bundles, -1)
            codes = (packed / wscale[:, None, :, None]).to(torch.float32).to(torch.float8_e4m3fn).view(torch.uint8)
            if codes.shape[1] % 2:  # the kernel's pipeline step holds two items
                codes = torch.cat((codes, torch.zeros_like(codes[:, :1])), 1)
            wparts.append(codes.contiguous().reshape(-1))
            mparts.append((wscale[:1] if (isinstance(sub, tuple) and sub[0] == 'blphase') else wscale).reshape(-1).to(torch.float32))
            wide = lambda t_: plan.tensors[t_].get('phases', 1) == 4  # bf16 partial sums: values, no code scale
            op_scales[-1] = ((1. if wide(op['res']) else float(act_scales[op['res']])) if op['res'] is not None else 0.,
                             (1. if wide(op['dst']) else 1. / float(act_scales[op['dst']])) if op['dst'] is not None else 0.)
            if effective_weights is not None and (phase or (isinstance(sub, tuple) and sub[0] == 'lateral')):
                effective_weights.append(dict(w=None, b=None))  # (a member op: the simulator follows the head op it restates)
            elif effective_weights is not None:
                dq = codes[:, :packed.shape[1]].view(torch.float8_e4m3fn).to(torch.float64) * wscale[:, None, :, None]
                dq = dq.reshape(bundles, cin_b // KC, k * k, cout_b, KC).permute(0, 3, 1, 4, 2)  # [B][cout][chunk][64][tap]
                dq = dq.reshape(bundles, cout_b, cin_b, k, k)
                if geo is None:
                    weff = torch.zeros_like(w)
                    if groups == 1:
                        weff[:, :c0] = dq[0, :cout, :c0] / act_scales[op['src0']]
                        if c1:
                            weff[:, c0:] = dq[0, :cout, c0p:c0p + c1] / act_scales[op['src1']]
                    else:
                        cig_, cog_ = cin // groups, cout // groups
                        for g_ in range(groups):
                            weff[g_ * cog_:(g_ + 1) * cog_] = dq[0, g_ * cog_:(g_ + 1) * cog_, g_ * cig_:(g_ + 1) * cig_]
                        weff /= act_scales[op['src0']]
                else:
                    cig_ = cin // groups
                    weff = torch.stack([dq[:, g_ * cig_:(g_ + 1) * cig_, g_ * cig_:(g_ + 1) * cig_] for g_ in range(gpb)], 1)
                    weff = weff.reshape(cout, cig_, k, k) / act_scales[op['src0']]
                effective_weights.append(dict(w=weff, b=b.clone()))
        elif f32:
            wparts.append(packed.contiguous().reshape(-1).to(wdt))
        else:  # bf16: [bundle][item][cout][32]; the kernel's pipeline step holds two items -> pad an odd item count
            packed = packed.contiguous().reshape(bundles, -1, cout_b, KC)
            if packed.shape[1] % 2:
                packed = torch.cat((packed, torch.zeros_like(packed[:, :1])), 1)
            wparts.append(packed.contiguous().reshape(-1).to(wdt))
        if bias is not None:
            bparts.append(bias.to(torch.float32))
        d.op = _lib.OP_CONV_DEFERRED if op.get('deferred') else _lib.OP_CONV
        d.subpixel = {None: _lib.SUBPIXEL_NONE, 'head': _lib.SUBPIXEL_HEAD, 'phase': _lib.SUBPIXEL_PHASE,
                      'lateral': _lib.SUBPIXEL_LATERAL, 'scatter': _lib.SUBPIXEL_SCATTER, 'blhead': _lib.SUBPIXEL_BL_HEAD,
                      'blphase': _lib.SUBPIXEL_BL_PHASE, 'blframe': _lib.SUBPIXEL_BL_FRAME}[sub[0] if isinstance(sub, tuple) else sub]
        d.src0 = op['src0']
        d.src1 = -1 if op['src1'] is None else op['src1']
        d.res = -1 if op['res'] is None else op['res']
        d.dst = -1 if op['dst'] is None else op['dst']
        d.up0, d.up1 = (2 if op['up0'] == 'bilinear' else int(op['up0'])), int(op['up1'])
        d.res_up = 2 if op['res_up'] == 'shuffle' else int(op['res_up'])
        d.c0_used = c0p if op['src1'] is not None else cinp
        d.kh = d.kw = k
        d.stride, d.pad = op['stride'], op['pad']
        d.bundles, d.cin_b, d.cout_b = bundles, cin_b, cout_b
        d.weight_offset, d.bias_offset = woff, (boff if bias is not None else -1)
        d.act, d.act_scale = _ACT[op['act']], float(op['act_scale'])
        d.out_index = -1 if op['out_index'] is None else op['out_index']
        d.cout_real = cout
        d.fuse_weight_offset = d.fuse_bias_offset = -1
        d.fuse_cout = 0
        woff += wparts[-1].numel() * wsz
        if bias is not None:
            boff += bparts[-1].numel()
        # keep blob offsets 16-byte aligned
        padw = (-wparts[-1].numel()) % (16 if fp8 else 8)
        if padw:
            wparts.append(torch.zeros(padw, dtype=torch.uint8 if fp8 else wdt))
            woff += padw * wsz
        if op.get('fuse'):
            assert not f32, 'fused heads are a bf16-only feature'
            fz = op['fuse']
            w2 = state_dict[fz['w'] + 'weight'].detach().double().cpu().reshape(fz['cout'], cout)
            b2 = state_dict[fz['w'] + 'bias'].detach().double().cpu()
            W2 = torch.zeros(32, coutp, dtype=torch.float64)
            W2[:fz['cout'], :cout] = w2
            B2 = torch.zeros(32, dtype=torch.float64)
            B2[:fz['cout']] = b2
            wparts.append(W2.reshape(-1).to(torch.bfloat16).view(torch.uint8) if fp8 else W2.reshape(-1).to(torch.bfloat16))
            bparts.append(B2.to(torch.float32))
            if fp8:
                mparts.append(torch.ones(32, dtype=torch.float32))  # keeps mult/bias indices aligned
            d.fuse_weight_offset, d.fuse_bias_offset = woff, boff
            d.fuse_cout, d.fuse_act, d.fuse_act_scale = fz['cout'], _ACT[fz['act']], float(fz['act_scale'])
            d.cout_real = fz['cout']
            woff += wparts[-1].numel() * (1 if fp8 else 2)
            boff += bparts[-1].numel()
    wblob = torch.cat(wparts).to(device)
    bblob = torch.cat(bparts).to(device)
    if fp8:  # multipliers live behind the biases in ONE float blob (cpn_op_desc.mult_offset); bias and multiplier
        nb = bblob.numel()  # entries share their relative offsets
        for d in ops:
            if d.op == _lib.OP_CONV and d.bias_offset >= 0:
                d.mult_offset = nb + d.bias_offset
        allb = torch.cat((bblob, torch.cat(mparts).to(device)))
        return tens, ops, wblob, allb, allb[nb:], op_scales
    return tens, ops, wblob, bblob


def reference_flops(plan: Plan, H, W, only=None):
    """2*MAC FLOPs of the reference graph per input (convs only, batch 1).  ``only``: optional
    predicate on the op dict (e.g. the backbone stack: ``lambda op: 'backbone' in op['w']``)."""
    total = 0.
    for op in plan.ops:
        sub = op.get('sub')
        if only is not None and op['op'] == 'conv' and not only(op):
            continue
        if op['op'] != 'conv' or (isinstance(sub, tuple) and sub[0] != 'scatter'):  # (phase / lateral / bl ops restate their head)
            continue
        t0 = plan.tensors[op['src0']]
        scatter = isinstance(sub, tuple)  # stands for the reference's 3x3 conv over the x2-upsampled source
        down_in = 1 if op['up0'] == 'bilinear' else t0['down'] // (2 if (op['up0'] or scatter) else 1)
        ho, wo = H // (down_in * op['stride']), W // (down_in * op['stride'])
        f = 2. * ho * wo * op['cout'] * (op['cin'] // op['groups']) * (3 if scatter else op['k']) ** 2
        # the reference runs the UNet inner 1x1 after the upsample (4x the pixels), unet.py:213-218
        if 'inner_blocks' in op['w'] and '.unet.' in op['w']:
            f *= 4
        total += f
        if op.get('fuse'):
            total += 2. * ho * wo * op['fuse']['cout'] * op['cout']
    return total
