"""Golden-vector generator (BUILD CONTAINER ONLY; needs /root/reference, which never travels).

Imports the read-only Python reference (celldetection 0.4.9) through ``oracle/ref_shim.py`` and writes small
``.npz`` fixtures (inputs + expected outputs, nothing of the reference's source) next to this file:

* ``ops.npz``      G1-G5: fouriers2contours / local_refinement / rel_location2abs_location / scale_* /
                   NMS (+chunked ``batched_box_nmsi``) / remove_border_contours / stitching rule
* ``tiling.npz``   G6: ``get_tiling_slices`` tables
* ``model_<name>.npz`` G7: tiny-model end-to-end: synthetic weights are re-creatable from (seed, key names)
                   via ``celldetection_amd.synth``; the file holds the 8 calibrated head tensors, the input,
                   the five ``CPNCore`` maps and the full ``CPN.forward`` outputs (nms on/off, offsets, bounds)
* ``stitch.npz``   G8: multi-tile stitch (TileLoader offsets/overlaps -> border removal -> global NMS)
* ``labels.npz``   G10: ``celldetection.data.cpn.contours2labels`` (the reference's own loop, imported; the one cv2 call
                   inside it goes through ``ref_shim.cv2_drawContours`` = the restated fill, third-party unpinned):
                   overlapping / nested contours, gap 0 / 3, initial_depth 1 / 2, ioa_thresh None / .3 / .8 with
                   return_indices, sort_by ascending / descending, unrounded and unclipped inputs, ragged lists
* ``preprocess.npz`` G11: ``cd.data.normalize_percentile`` (data/misc.py:156-161) and the script's ``preprocess``
                   (cpn_inference.py:196-222), both IMPORTED; skimage / cv2 / albumentations calls inside them go through the
                   stand-ins of ``ref_shim`` (third-party arithmetic: unpinned); ``to_uint8=False`` involves numpy only
* ``forward_tiled.npz`` G12: ``LitCpn.forward_tiled`` (models/lightning_cpn.py:88-177), the reference's own method on the imported
                   class (``ref_shim.LightningModule`` stands in for pytorch_lightning's base class): per-tile ``CPN.forward``
                   outputs as the method saw them + its final per-image results, for masks / extra keys / other parameters
* ``stitch_dups.npz`` G8b: the stitching rule on synthetic per-tile detections WITH cross-tile duplicates (the global
                   NMS removes > 10 % of what survives the border rule)

Run:  python tests/golden/make_golden.py
"""
import os
import sys
import warnings

from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')

import ref_shim  # noqa: E402

cd = ref_shim.import_reference()
from celldetection.models.cpn import local_refinement  # noqa: E402
from celldetection.ops import cpn as rops  # noqa: E402
from celldetection_amd.synth import synth_state_dict, calibrate_heads  # noqa: E402

torch.set_num_threads(4)

MODEL_SPECS = {
    # name: (class, kwargs, input shape)
    'CpnU22': ('CpnU22', dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}),
               (2, 3, 64, 96)),
    'CpnResNeXt101UNet': ('CpnResNeXt101UNet',
                          dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}),
                          (2, 3, 64, 96)),
    'CpnResNet18FPN': ('CpnResNet18FPN', dict(in_channels=3, backbone_kwargs={
        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}), (2, 3, 64, 96)),
    'CpnResNet50FPN': ('CpnResNet50FPN', dict(in_channels=3, backbone_kwargs={
        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}), (1, 3, 96, 64)),
    'CpnResNet50UNet': ('CpnResNet50UNet',
                        dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}),
                        (1, 3, 64, 64)),
    # wider variant whose channel counts are MFMA-friendly (multiples of 32), samples/order changed
    'CpnU22_wide': ('CpnU22', dict(in_channels=3, order=7, samples=48, score_thresh=.8, nms_thresh=.3,
                                   backbone_kwargs={'backbone_kwargs': {'base_channels': 32}}),
                    (1, 3, 96, 128)),
    # CPN.forward variants: bucketed refinement, uncertainty head (+ certainty filter and
    # uncertainty-weighted NMS), multi-class scores, narrower head channels / other head kernel sizes
    'CpnU22_buckets': ('CpnU22', dict(in_channels=3, refinement_buckets=6,
                                      backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}), (2, 3, 64, 96)),
    'CpnU22_uncertainty': ('CpnU22', dict(in_channels=3, uncertainty_head=True, uncertainty_nms=True,
                                          certainty_thresh=.65,
                                          backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}), (2, 3, 64, 96)),
    'CpnU22_classes4': ('CpnU22', dict(in_channels=3, classes=4,
                                       backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}), (2, 3, 64, 96)),
    'CpnResNet18FPN_heads': ('CpnResNet18FPN', dict(in_channels=3, contour_head_channels=24,
                                                    refinement_head_channels=8, kernel_size_score=3,
                                                    kernel_size_refinement=5, refinement_buckets=3, backbone_kwargs={
                                                        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}),
                             (1, 3, 64, 64)),
    # arbitrary input sizes (not multiples of 32; odd): conv / pool floor rules, top-down maps nearest-resized to the
    # lateral's size (unet.py:213-217, torchvision FPN), features bilinear-resized to the input size (cpn.py:277-279)
    'CpnResNeXt101UNet_odd': ('CpnResNeXt101UNet',
                              dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}),
                              (1, 3, 75, 101)),
    'CpnResNeXt101UNet_100x140': ('CpnResNeXt101UNet',
                                  dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}),
                                  (2, 3, 100, 140)),
    'CpnResNet18FPN_odd': ('CpnResNet18FPN', dict(in_channels=3, backbone_kwargs={
        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}), (1, 3, 75, 101)),
    'CpnU22_odd': ('CpnU22', dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}),
                   (1, 3, 75, 101)),
    'CpnU22_300': ('CpnU22', dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}),
                   (1, 3, 300, 300)),
    # head options of CPNCore (cpn.py:125-236): strided ReadOut heads, heads reading other / fused (Fuse2d) features
    'CpnU22_strided': ('CpnU22', dict(in_channels=3, contour_head_stride=2, refinement_head_stride=2,
                                      backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}), (2, 3, 64, 96)),
    'CpnResNet18FPN_fuse': ('CpnResNet18FPN', dict(in_channels=3, score_features=['1', '2'], contour_features=['1', '2'],
                                                   location_features=['1', '2'], backbone_kwargs={
                                                       'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}),
                            (1, 3, 64, 96)),
    # Fuse2d over THREE features (the HIP plan splits the 1x1 conv: two concat sources + one nearest-resized residual)
    'CpnResNet18FPN_fuse3': ('CpnResNet18FPN', dict(in_channels=3, score_features=['1', '2', '3'],
                                                    contour_features=['1', '2', '3'], location_features=['1', '3', '2'],
                                                    refinement_features=['0', '1', '2'], backbone_kwargs={
                                                        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}),
                             (1, 3, 64, 96)),
    # Fuse2d over FOUR / FIVE features, finer and coarser than the first one (the HIP plan continues the sum in steps of two)
    'CpnResNet18FPN_fuse5': ('CpnResNet18FPN', dict(in_channels=3, score_features=['1', '2', '3', '0'],
                                                    contour_features=['1', '0', '2', '3', '2'],
                                                    location_features=['1', '3', '0', '2'],
                                                    refinement_features=['0', '1', '2', '3'], backbone_kwargs={
                                                        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}),
                             (1, 3, 75, 101)),
    # head strides 4 / 8 (the HIP plan: k x k conv at stride 2 + the 1x1 conv at stride s / 2)
    'CpnU22_stride4': ('CpnU22', dict(in_channels=3, contour_head_stride=4, refinement_head_stride=8,
                                      backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}), (2, 3, 128, 160)),
    # hidden activations of the ReadOut heads other than ReLU (head_activation / head_activation_<head>, cpn.py:183-233)
    'CpnU22_headact': ('CpnU22', dict(in_channels=3, head_activation='silu', head_activation_score='gelu',
                                      head_activation_refinement='LeakyReLU',
                                      backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}), (2, 3, 64, 96)),
    'CpnResNet18FPN_headact': ('CpnResNet18FPN', dict(in_channels=3, head_activation='elu', head_activation_fourier='tanh',
                                                      head_activation_location='mish', head_activation_refinement='hardswish',
                                                      backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}),
                               (1, 3, 64, 96)),
    # round 6 (keep in sync with tests/model_specs.py)
    'CpnResNet18UNet': ('CpnResNet18UNet', dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}), (2, 3, 64, 96)),
    'CpnResNet34UNet': ('CpnResNet34UNet', dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}), (1, 3, 96, 64)),
    'CpnResUNet': ('CpnResUNet', dict(in_channels=3, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}), (2, 3, 64, 96)),
    'CpnSlimU22': ('CpnSlimU22', dict(in_channels=3), (1, 3, 64, 96)),
    'CpnWideU22': ('CpnWideU22', dict(in_channels=1, order=3), (1, 1, 48, 64)),
    'CpnResNet18FPN_lowres': ('CpnResNet18FPN', dict(in_channels=3, refinement_full_res=False, backbone_kwargs={
        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}), (2, 3, 64, 96)),
    'CpnResNet18FPN_bicubic': ('CpnResNet18FPN', dict(in_channels=3, refinement_interpolation='bicubic', backbone_kwargs={
        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}), (1, 3, 75, 101)),
    'CpnResNet18FPN_bicubic_lowres': ('CpnResNet18FPN', dict(in_channels=3, refinement_interpolation='bicubic',
                                                             refinement_full_res=False, backbone_kwargs={
        'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}), (2, 3, 64, 96)),
    'CpnResNet18FPN_fusekw3': ('CpnResNet18FPN', dict(in_channels=3, score_features=['1', '2'], contour_features=['1', '0'],
                                                      location_features=['1', '2'],
                                                      fuse_kwargs=dict(kernel_size=3, padding=1, activation='LeakyReLU'),
                                                      backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}),
                               (1, 3, 64, 96)),
    'CpnResNet18FPN_fusekw': ('CpnResNet18FPN', dict(in_channels=3, score_features=['1', '2', '3'], contour_features=['1', '2'],
                                                     refinement_features=['0', '1', '2'],
                                                     fuse_kwargs=dict(norm_layer=None, activation=None, bias=False),
                                                     backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}}),
                              (1, 3, 75, 101)),
    'CpnResNet50UNet_feats': ('CpnResNet50UNet', dict(in_channels=3, score_features='2', contour_features='2',
                                                      location_features='2', refinement_features=['0', 'encoder.0'],
                                                      backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}),
                              (1, 3, 64, 64)),
}


def npy(t):
    if isinstance(t, (list, tuple)):
        return [npy(i) for i in t]
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays')


# ------------------------------------------------------------------------------------------------------------
def gen_ops():
    g = torch.Generator().manual_seed(1234)
    out = {}
    # G1 fouriers2contours
    for tag, (p, o, s) in {'a': (37, 5, 32), 'b': (5, 8, 128), 'c': (3, 1, 7), 'd': (11, 12, 64)}.items():
        f = torch.randn(p, o, 4, generator=g) * 3
        loc = torch.rand(p, 2, generator=g) * 200
        con, sampling = rops.fouriers2contours(f, loc, samples=s)
        out[f'f2c_{tag}_fourier'], out[f'f2c_{tag}_loc'], out[f'f2c_{tag}_out'] = npy(f), npy(loc), npy(con)
    # G2 local_refinement (buckets=1): includes .5 coordinates, out-of-range, negative
    n, h, w = 2, 24, 40
    refinement = (torch.rand(n, 2, h, w, generator=g) * 2 - 1) * 3
    con = torch.rand(9, 16, 2, generator=g) * torch.tensor([w + 8., h + 8.]) - 4
    con[0, :8] = torch.floor(con[0, :8]) + .5  # exact halves -> round-half-even
    con[1, :4] = torch.tensor([[-.5, .5], [1.5, 2.5], [w - .5, h - .5], [w + 3., -7.]])
    b = torch.tensor([0, 1, 1, 0, 0, 1, 0, 1, 1])
    for iters in (1, 4):
        res, all_res = local_refinement(con.clone(), refinement, num_loops=iters, num_buckets=1,
                                        original_size=(h, w), sampling=None, b=b)
        out[f'refine_out_{iters}'] = npy(res)
        if iters == 4:
            out['refine_all'] = np.stack(npy(all_res))
    out['refine_in'], out['refine_map'], out['refine_b'] = npy(con), npy(refinement), npy(b)
    # G2b bucketed refinement (cpn.py:72-82) with the default sampling: bucket tables + refined contours
    for nb, smp in ((6, 16), (3, 16)):
        sampling = torch.linspace(0, 1.0, smp)
        tab = rops.resolve_refinement_buckets(sampling, nb)
        out[f'bucket_idx_{nb}'] = np.stack([npy(i) for i, _ in tab])
        out[f'bucket_w_{nb}'] = np.stack([npy(w_) for _, w_ in tab])
        ref_b = (torch.rand(n, 2 * nb, h, w, generator=torch.Generator().manual_seed(4321 + nb)) * 2 - 1) * 3
        res, _ = local_refinement(con.clone(), ref_b, num_loops=3, num_buckets=nb, original_size=(h, w),
                                  sampling=sampling, b=b)
        out[f'refine_bucket_map_{nb}'], out[f'refine_bucket_out_{nb}'] = npy(ref_b), npy(res)
    # G3 locations / scaling
    loc = torch.randn(2, 2, 5, 7, generator=g)
    out['rel2abs_in'], out['rel2abs_out'] = npy(loc), npy(rops.rel_location2abs_location(loc))
    c = torch.randn(6, 9, 2, generator=g) * 10
    out['scale_con_in'] = npy(c)
    out['scale_con_out'] = npy(rops.scale_contours((24, 40), (96, 120), c))
    f, l = torch.randn(6, 5, 4, generator=g), torch.randn(6, 2, generator=g)
    out['scale_f_in'], out['scale_l_in'] = npy(f), npy(l)
    fo, lo = rops.scale_fourier((24, 40), (96, 120), f.clone(), l.clone())
    out['scale_f_out'], out['scale_l_out'] = npy(fo), npy(lo)
    # G4 NMS (third-party torchvision semantics as restated in oracle/ref_shim.py -> "unpinned")
    m = 700
    xy = torch.rand(m, 2, generator=g) * 100
    wh = torch.rand(m, 2, generator=g) * 30
    boxes = torch.cat((xy, xy + wh), 1)
    boxes[5] = boxes[4]  # duplicates
    boxes[10, 2:] = boxes[10, :2]  # zero area
    boxes[11] = boxes[10]  # identical zero-area boxes -> NaN IoU
    scores = torch.rand(m, generator=g)
    scores[20:40] = scores[20]  # ties
    scores[4] = scores[5]
    out['nms_boxes'], out['nms_scores'] = npy(boxes), npy(scores)
    for thr in (.2, .5, 0.):
        out[f'nms_keep_{thr}'] = npy(torch.ops.torchvision.nms(boxes, scores, thr))
    out['nmsi_chunked_keep'] = npy(rops.batched_box_nmsi([boxes], [scores], .2, batch_size=128)[0])
    out['nmsi_plain_keep'] = npy(rops.batched_box_nmsi([boxes, boxes[:50]], [scores, scores[:50]], .2)[1])
    # G5 border removal / stitching rule
    con = torch.rand(40, 12, 2, generator=g) * torch.tensor([64., 48.])
    offsets = torch.tensor([-3., 5.])
    for i, flags in enumerate([(True, True, True, True), (False, True, False, True), (True, False, True, False)]):
        top, right, bottom, left = flags
        out[f'border_keep_{i}'] = npy(rops.remove_border_contours(con, (48, 64), 4, top=top, right=right,
                                                                  bottom=bottom, left=left, offsets=offsets))
    out['border_in'], out['border_offsets'] = npy(con), npy(offsets)
    overlaps = torch.tensor([[8, 12], [0, 16]])
    out['stitch_overlaps'] = npy(overlaps)
    out['stitch_keep'] = npy(rops.filter_contours_by_stitching_rule(con, (48, 64), overlaps, rule='ex_br',
                                                                    offsets=offsets))
    # G1b fouriers2contours with a caller-supplied sampling vector (appended in round 6 with its own generator: the arrays above
    # keep their values)
    g2 = torch.Generator().manual_seed(4242)
    f = torch.randn(9, 6, 4, generator=g2) * 2
    loc = torch.rand(9, 2, generator=g2) * 100
    # (32 samples: with 2 * S a multiple of the SIMD block torch's CPU sum over `order` runs in ascending order -- for other S its
    # remainder columns go through a 4-accumulator cascade, a machine-dependent order that only the 1e-4 tolerance tests cover)
    samp = torch.sort(torch.rand(32, generator=g2)).values
    con, samp_out = rops.fouriers2contours(f, loc, samples=5, sampling=samp)  # (`samples` is ignored then)
    assert torch.equal(samp_out, samp) and con.shape == (9, 32, 2)
    out['f2c_s_fourier'], out['f2c_s_loc'], out['f2c_s_sampling'], out['f2c_s_out'] = npy(f), npy(loc), npy(samp), npy(con)
    save('ops.npz', **out)


def gen_tiling():
    out = {}
    cases = {'a': ((16384, 16384), 512, 384), 'b': ((16384, 16384), 512, 512), 'c': ((16384, 16384), 1024, 768),
             'd': ((1000, 700), 512, 384), 'e': ((300, 300), 512, 384), 'f': ((1025, 513), (512, 256), (500, 200))}
    for tag, (size, crop, stride) in cases.items():
        slices, overlaps, shape = cd.get_tiling_slices(size, crop, stride, return_overlaps=True)
        sl = np.array([[[s.start, s.stop] for s in item] for item in slices], dtype=np.int64)
        ov = np.array([[list(o) for o in item] for item in overlaps], dtype=np.int64)
        out[f'{tag}_size'] = np.array(size)
        out[f'{tag}_crop'] = np.array(crop if isinstance(crop, tuple) else (crop, crop))
        out[f'{tag}_stride'] = np.array(stride if isinstance(stride, tuple) else (stride, stride))
        out[f'{tag}_slices'], out[f'{tag}_overlaps'], out[f'{tag}_shape'] = sl, ov, np.array(shape)
    save('tiling.npz', **out)


def build_ref_model(name, seed=0, **cal_kw):
    cls, kwargs, shape = MODEL_SPECS[name]
    model = getattr(cd.models, cls)(**kwargs).eval()
    sd = synth_state_dict(model.state_dict(), seed=seed)
    x_cal = torch.rand(*shape, generator=torch.Generator().manual_seed(99))

    def core_fn(sd_):
        model.load_state_dict(sd_)
        with torch.no_grad():
            s, l, r, f, _ = model.core(x_cal)
        return s, l, r, f

    sd, overrides = calibrate_heads(sd, core_fn, **cal_kw)
    model.load_state_dict(sd)
    return model, overrides, shape


def flat_outputs(prefix, y, out):
    for k, v in y.items():
        if v is None:
            continue
        for i, t in enumerate(v):
            out[f'{prefix}.{k}.{i}'] = npy(t)


# per-spec calibrate_heads arguments.  The head-option / odd-size models have coarse head grids (stride 4: 16 x 24 pixels):
# denser scores and smaller contours so that >= 30 detections per image survive the NMS (with 1..8 kept
# detections the keep-index part of those fixtures pinned almost nothing)
# stride-4 head grid (16 x 24 pixels per 64 x 96 image): dense scores, small contours and SMALL refinement steps -- with the
# default refinement_raw_std = 1 (up to 3 px per iteration) the pixel snapping of local_refinement turns bf16-sized differences
# into pixel-sized box changes, which an IoU > 0.5 match of small boxes does not survive (measured: bf16 0.88, fp8 0.55)
_FPN_DENSE = dict(score_shift=1.5, fourier_std=.4, location_std=.4, refinement_raw_std=.3)
_U22_SMALL = dict(score_shift=-.5, fourier_std=.3, location_std=.4)                   # full-resolution head grid: smaller contours
MODEL_CALIBRATION = {'CpnU22_classes4': dict(score_shift=-3.5),
                     # every model fixture keeps >= 30 detections per image after the NMS (the FPN families of
                     # BASELINE configs[1] / [4] and the only pin of uncertainty_nms kept 2..5)
                     'CpnResNet18FPN': _FPN_DENSE, 'CpnResNet18FPN_odd': _FPN_DENSE, 'CpnResNet18FPN_heads': dict(_FPN_DENSE, score_shift=2.2, fourier_std=.35),
                     # (deepest tiny model: its fp8 run needs larger boxes for an IoU match; a lower logit spread keeps the dense scores
                     # out of the sigmoid's saturation, where fp32 summation-order noise re-orders near-equal scores)
                     'CpnResNet50FPN': dict(_FPN_DENSE, score_shift=2.4, score_gain=1.5, fourier_std=.6, refinement_raw_std=.2), 'CpnU22': _U22_SMALL, 'CpnU22_buckets': dict(_U22_SMALL, score_shift=-.2, refinement_raw_std=.3),
                     'CpnU22_uncertainty': dict(score_shift=1., fourier_std=.3, location_std=.4),
                     # (CpnResNeXt101UNet keeps its round-1 fixture, 21 kept per image: a denser one put two proposals within
                     # fp32 summation-order noise of each other in the score sort -- a tie the fp32 parity test cannot order)
                     'CpnResNeXt101UNet_odd': dict(fourier_std=1.),
                     'CpnResNet50UNet': dict(fourier_std=.6),
                     'CpnU22_strided': dict(score_shift=0., fourier_std=.12, location_std=.3),
                     'CpnU22_odd': dict(score_shift=-1.5, fourier_std=.25, location_std=.4),
                     'CpnResNet50UNet_feats': dict(score_shift=-.3, fourier_std=.25, location_std=.4),
                     'CpnU22_stride4': dict(score_shift=1.2, fourier_std=.07, location_std=.3, refinement_raw_std=.3),
                     'CpnU22_headact': _U22_SMALL, 'CpnResNet18FPN_headact': _FPN_DENSE,
                     'CpnResNet18UNet': _U22_SMALL, 'CpnResNet34UNet': _U22_SMALL, 'CpnResUNet': _U22_SMALL, 'CpnSlimU22': _U22_SMALL,
                     'CpnWideU22': _U22_SMALL, 'CpnResNet18FPN_lowres': _FPN_DENSE, 'CpnResNet18FPN_bicubic': _FPN_DENSE,
                     'CpnResNet18FPN_bicubic_lowres': _FPN_DENSE,
                     'CpnResNet18FPN_fusekw3': dict(score_shift=1., fourier_std=.35, location_std=.4, refinement_raw_std=.3),
                     'CpnResNet18FPN_fusekw': dict(score_shift=.5, fourier_std=.4, location_std=.4, refinement_raw_std=.3),
                     'CpnResNet18FPN_fuse': dict(score_shift=-.5, fourier_std=.4, location_std=.4),
                     'CpnResNet18FPN_fuse3': dict(score_shift=.5, fourier_std=.4, location_std=.4, refinement_raw_std=.3),
                     'CpnResNet18FPN_fuse5': dict(score_shift=.5, fourier_std=.4, location_std=.4, refinement_raw_std=.3)}


def gen_model(name, seed=0):
    model, overrides, shape = build_ref_model(name, seed, **MODEL_CALIBRATION.get(name, {}))
    out = {f'override.{k}': npy(v) for k, v in overrides.items()}
    out['seed'] = np.array(seed)
    tmpl = model.state_dict()
    out['sd_keys'] = np.array(list(tmpl.keys()))
    out['sd_shapes'] = np.array([','.join(str(int(d)) for d in v.shape) for v in tmpl.values()])
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(7))
    out['x'] = npy(x)
    with torch.no_grad():
        s, l, r, f, u = model.core(x)
        out['core.scores'], out['core.locations'], out['core.refinement'], out['core.fourier'] = \
            npy(s), npy(l), npy(r), npy(f)
        if u is not None:
            out['core.uncertainty'] = npy(u)
        flat_outputs('nms', model(x), out)
        flat_outputs('nonms', model(x, nms=False), out)
        offsets = torch.tensor([[100, 200], [-5, 7]][:shape[0]])
        out['offsets'] = npy(offsets)
        flat_outputs('offs', model(x, offsets=offsets.clone()), out)
        g = torch.Generator().manual_seed(3)
        ub = (torch.rand(shape[0], 1, max(shape[2] // 4, 1), max(shape[3] // 4, 1), generator=g) > .3).float()
        lb = (torch.rand(shape[0], 1, max(shape[2] // 8, 1), max(shape[3] // 8, 1), generator=g) > .97).float()
        out['scores_upper_bound'], out['scores_lower_bound'] = npy(ub), npy(lb)
        flat_outputs('bounds', model(x, scores_upper_bound=ub, scores_lower_bound=lb), out)
        if name == 'CpnU22':  # no refinement: contours IS contour_proposals -> both `+= offsets` hit one tensor
            model.refinement_iterations = 0
            flat_outputs('noref_offs', model(x, offsets=offsets.clone()), out)
            model.refinement_iterations = 4
        if name == 'CpnU22':  # run-time attribute changes (config/flags)
            model.samples, model.refinement_iterations, model.score_thresh, model.nms_thresh = 17, 2, .7, .5
            flat_outputs('attr', model(x), out)
            model.order = 3
            flat_outputs('attr_order3', model(x), out)
    n_det = [len(t) for t in model(x)['scores']] if name != 'CpnU22' else None
    # smallest gap between two different proposal scores of an image: the fp32 verification path reproduces the NMS order only
    # if that gap exceeds the conv stack's fp32 summation-order noise (~1e-6 relative) -- exact ties are ordered by index
    gaps = []
    for i in range(shape[0]):
        d = np.diff(np.sort(out[f'nonms.scores.{i}'].astype(np.float64)))
        gaps.append(float(d[d > 0].min()) if (d > 0).any() else float('inf'))
    print(name, 'detections/img (last config):', n_det, 'proposals/img (nonms):',
          [len(out[f'nonms.scores.{i}']) for i in range(shape[0])], 'min score gap', ['%.1e' % v for v in gaps])
    save(f'model_{name}.npz', **out)


def gen_stitch():
    """G8: TileLoader/apply_model semantics on a 2x3-tile image with the tiny CpnU22 (reference loop restated with
    the reference's own functions: get_tiling_slices, CPN.forward(offsets), remove_border_contours, nms)."""
    model, overrides, _ = build_ref_model('CpnU22', 0, fourier_std=.6, location_std=.5, score_shift=-3.)
    H, W, crop, stride = 160, 224, (96, 96), (64, 64)
    img = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(11))
    slices, overlaps, shape = cd.get_tiling_slices((H, W), crop, stride, return_overlaps=True)
    slices, overlaps = list(slices), list(overlaps)
    h_tiles, w_tiles = shape
    coll = {}
    with torch.no_grad():
        for idx, (sl, ov) in enumerate(zip(slices, overlaps)):
            tile = img[(...,) + sl]
            offs = torch.as_tensor([[sl[1].start, sl[0].start]])
            y = model(tile, offsets=offs.clone())
            h_i, w_i = np.unravel_index(idx, shape)
            keep = rops.remove_border_contours(y['contours'][0], crop, 4, top=h_i > 0, right=w_i < w_tiles - 1,
                                               bottom=h_i < h_tiles - 1, left=w_i > 0, offsets=-offs[0])
            for k in ('contours', 'boxes', 'scores', 'classes', 'locations', 'fourier', 'contour_proposals'):
                v = y[k][0][keep]
                coll[k] = torch.cat((coll[k], v)) if k in coll else v
    keep = torch.ops.torchvision.nms(coll['boxes'], coll['scores'], model.nms_thresh)
    out = {f'override.{k}': npy(v) for k, v in overrides.items()}
    out.update(img=npy(img), crop=np.array(crop), stride=np.array(stride), border=np.array(4))
    tmpl = model.state_dict()
    out['sd_keys'] = np.array(list(tmpl.keys()))
    out['sd_shapes'] = np.array([','.join(str(int(d)) for d in v.shape) for v in tmpl.values()])
    out['pre_nms_count'] = np.array(len(coll['scores']))
    for k, v in coll.items():
        out[f'final.{k}'] = npy(v[keep])
    print('stitch: tiles', shape, 'pre-nms', len(coll['scores']), 'final', len(keep))
    save('stitch.npz', **out)


def gen_stitch_dups():
    """G8b: the stitching rule itself (border removal per tile -> concat -> ONE global NMS, cpn_inference.py:370-408)
    with the reference's own functions on synthetic per-tile detection lists that DO contain cross-tile duplicates:
    objects of a global canvas are 'detected' by every tile that contains their centre (tile-specific sub-pixel jitter
    and score noise), so the global NMS has real work (the model-level stitch.npz happens to suppress nothing)."""
    rng = np.random.default_rng(2024)
    H, W, crop, stride, border, S, O = 400, 520, (128, 128), (96, 96), 4, 16, 3
    slices, overlaps, shape = cd.get_tiling_slices((H, W), crop, stride, return_overlaps=True)
    slices, overlaps = list(slices), list(overlaps)
    h_tiles, w_tiles = shape
    n_obj = 260
    ctr = rng.uniform([2, 2], [W - 2, H - 2], (n_obj, 2))
    rad = rng.uniform(3, 9, n_obj)
    base_score = rng.uniform(.5, 1., n_obj)
    ang = np.linspace(0, 2 * np.pi, S, endpoint=False)
    out = dict(size=np.array((H, W)), crop=np.array(crop), stride=np.array(stride), border=np.array(border),
               n_tiles=np.array(len(slices)), nms_thresh=np.array(.2, np.float32))
    coll = {}
    pre = 0
    for idx, sl in enumerate(slices):
        h0, w0 = sl[0].start, sl[1].start
        inside = (ctr[:, 0] >= w0) & (ctr[:, 0] < w0 + crop[1]) & (ctr[:, 1] >= h0) & (ctr[:, 1] < h0 + crop[0])
        ids = np.nonzero(inside)[0]
        k = len(ids)
        jit = rng.uniform(-.3, .3, (k, 1, 2))
        con = ctr[ids, None] + rad[ids, None, None] * np.stack((np.cos(ang), np.sin(ang)), -1)[None] * \
            rng.uniform(.85, 1.15, (k, S, 1)) + jit
        con = np.clip(con, [w0, h0], [w0 + crop[1] - 1, h0 + crop[0] - 1]).astype(np.float32)  # tile-local clamp + offset
        sco = (base_score[ids] + rng.uniform(-.02, .02, k)).astype(np.float32)
        d = dict(contours=con, boxes=np.concatenate((con.min(1), con.max(1)), 1).astype(np.float32), scores=sco,
                 classes=np.ones(k, np.int64), locations=(ctr[ids] + jit[:, 0]).astype(np.float32),
                 fourier=rng.standard_normal((k, O, 4)).astype(np.float32), contour_proposals=(con + .25).astype(np.float32))
        for kk, v in d.items():
            out[f'tile{idx}.{kk}'] = v
        h_i, w_i = np.unravel_index(idx, shape)
        offs = torch.tensor([w0, h0])
        keep = rops.remove_border_contours(torch.as_tensor(con), crop, border, top=h_i > 0, right=w_i < w_tiles - 1,
                                           bottom=h_i < h_tiles - 1, left=w_i > 0, offsets=-offs)
        keep_br = keep & rops.filter_contours_by_stitching_rule(torch.as_tensor(con), crop, torch.as_tensor(overlaps[idx]),
                                                                rule='ex_br', offsets=-offs)
        out[f'tile{idx}.keep_border'] = npy(keep)
        out[f'tile{idx}.keep_border_exbr'] = npy(keep_br)
        pre += int(keep.sum())
        for kk, v in d.items():
            vv = torch.as_tensor(v)[keep]
            coll[kk] = torch.cat((coll[kk], vv)) if kk in coll else vv
    keep = torch.ops.torchvision.nms(coll['boxes'], coll['scores'], .2)
    out['pre_nms_count'] = np.array(pre)
    for kk, v in coll.items():
        out[f'final.{kk}'] = npy(v[keep])
    print('stitch_dups: tiles', shape, 'pre-nms', pre, 'final', len(keep))
    assert len(keep) <= 0.9 * pre
    save('stitch_dups.npz', **out)


def label_contours(seed, k, size, s=24, inside=False, nested=0):
    """Star-shaped float contours [k, s, 2] (xy) on an image of ``size`` (h, w): random centres (some outside the image
    unless ``inside``), radii 4..18, fractional coordinates incl. exact .5 values; the last ``nested`` ones sit inside the
    first ones."""
    rng = np.random.default_rng(seed)
    h, w = size
    t = np.linspace(0, 2 * np.pi, s, endpoint=False)
    out = []
    for i in range(k):
        r = rng.uniform(4, 18)
        lo = r * 1.4 + 1 if inside else -6
        c = np.array([rng.uniform(lo, w - 1 - lo), rng.uniform(lo, h - 1 - lo)])
        if i >= k - nested:
            j = i - (k - nested)
            c = out[j].mean(0) + rng.uniform(-1, 1, 2)
            r = rng.uniform(2, 4)
        rad = r * (1 + .3 * np.sin(rng.integers(2, 5) * t + rng.uniform(0, 6)))
        con = np.stack((c[0] + rad * np.cos(t), c[1] + rad * np.sin(t)), 1)
        con = np.round(con * 4) / 4  # quarters: exact .5 coordinates occur (np.round half-to-even)
        out.append(con.astype(np.float32))
    return np.stack(out)


LABEL_CASES = [
    # (name, contour spec, size, kwargs of contours2labels)
    ('default', dict(seed=1, k=60, nested=8), (120, 160), dict()),
    ('gap0', dict(seed=2, k=60, nested=8), (120, 160), dict(gap=0)),
    ('depth2', dict(seed=3, k=50, nested=5), (100, 140), dict(initial_depth=2)),
    ('ioa03', dict(seed=4, k=70, nested=12), (120, 160), dict(ioa_thresh=.3, return_indices=True)),
    ('ioa08', dict(seed=4, k=70, nested=12), (120, 160), dict(ioa_thresh=.8, return_indices=True, gap=0)),
    ('ioa_noidx', dict(seed=5, k=40, nested=6), (90, 90), dict(ioa_thresh=.5)),
    ('idx_noioa', dict(seed=5, k=40, nested=6), (90, 90), dict(return_indices=True)),
    ('sort_desc', dict(seed=6, k=60, nested=10), (120, 160), dict(sort_by='scores', ioa_thresh=.3, return_indices=True)),
    ('sort_asc', dict(seed=6, k=60, nested=10), (120, 160), dict(sort_by='scores', sort_descending=False)),
    ('sort_only', dict(seed=7, k=30, nested=4), (80, 100), dict(sort_by='scores', return_indices=True, ioa_thresh=.8)),
    ('unrounded', dict(seed=8, k=50, nested=6), (120, 160), dict(rounded=False)),
    ('unclipped', dict(seed=9, k=40, nested=6, inside=True), (120, 160), dict(clip=False)),
    ('raw', dict(seed=10, k=40, nested=6, inside=True), (120, 160), dict(clip=False, rounded=False, gap=1)),
    ('dense', dict(seed=11, k=150, nested=30), (96, 96), dict(ioa_thresh=.6, return_indices=True)),
    ('ragged', dict(seed=12, k=30, nested=4), (100, 100), dict()),
    ('empty', dict(seed=0, k=0), (40, 50), dict(initial_depth=2, return_indices=True)),
]


def gen_labels():
    """G10: outputs of the reference's own ``contours2labels`` loop (data/cpn.py:292-358)."""
    import json
    from celldetection.data.cpn import contours2labels
    out, meta = {}, []
    for name, spec, size, kw in LABEL_CASES:
        spec = dict(spec)
        k = spec.pop('k')
        con = label_contours(k=k, size=size, **spec) if k else np.zeros((0, 24, 2), np.float32)
        kw = dict(kw)
        if kw.get('sort_by') == 'scores':
            scores = np.random.default_rng(100 + spec['seed']).permutation(k).astype(np.float32) / k  # distinct values
            kw['sort_by'] = scores
            out[f'{name}.sort_by'] = scores
        if name == 'ragged':  # List[Array[num_points, 2]] of different lengths
            arg = [c[:24 - (i % 5) * 3].copy() for i, c in enumerate(con)]
            out[f'{name}.lengths'] = np.array([len(a) for a in arg])
        else:
            arg = con.copy()  # (clip without rounding works in place on the caller's array)
        res = contours2labels(arg, size, **kw)
        labels, keep = res if kw.get('return_indices') else (res, None)
        assert labels.dtype == np.int32 and labels.shape[:2] == tuple(size)
        out[f'{name}.contours'] = con
        out[f'{name}.labels'] = labels
        if keep is not None:
            out[f'{name}.keep'] = np.asarray(keep, np.int64)
        meta.append(dict(name=name, size=list(size), kwargs={a: (b if not isinstance(b, np.ndarray) else 'sort_by')
                                                            for a, b in kw.items()}))
        print(f'labels/{name}: {k} contours -> {labels.shape[2]} channels, max label {int(labels.max()) if labels.size else 0}'
              + (f', kept {len(keep)}' if keep is not None else ''))
    out['cases'] = np.array(json.dumps(meta))
    save('labels.npz', **out)


def gen_behaviours():
    """Error behaviour of options of the built rows, RECORDED from the imported reference: what
    ``CPN.forward`` does with ``functional=True`` and with a non-interpolating ``refinement_interpolation``."""
    import json
    rec = {}

    def run(fn):
        try:
            fn()
            return None
        except Exception as e:
            return dict(type=type(e).__name__, message=str(e))

    tiny = dict(backbone_kwargs={'backbone_kwargs': {'base_channels': 8}})
    m = cd.models.CpnU22(3, **tiny).eval()
    m.functional = True
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        # default init: no pixel above the threshold -> the empty advanced index does not fail; with proposals it does
        y = m(x)
        rec['functional_true_no_proposals'] = dict(error=None, fourier_shape=list(y['fourier'][0].shape),
                                                   contours_shape=list(y['contours'][0].shape))
        m.core.score_head.block[4].bias += 6.
        rec['functional_true_with_proposals'] = run(lambda: m(x))
        fpn = dict(backbone_kwargs={'fpn_channels': 16, 'backbone_kwargs': {'base_channel': 8}})
        for mode in ('nearest', 'area', 'nearest-exact'):
            mm = cd.models.CpnResNet18FPN(3, refinement_interpolation=mode, **fpn).eval()
            rec[f'refinement_interpolation_{mode}_fpn_forward'] = run(lambda: mm(x))
        # U22: level 0 has the input size -> no resize -> the mode is never used
        mu = cd.models.CpnU22(3, refinement_interpolation='nearest', **tiny).eval()
        rec['refinement_interpolation_nearest_u22_forward'] = run(lambda: mu(x))
        rec['slimu22_base_channels_kwarg'] = run(lambda: cd.models.CpnSlimU22(3, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}}))
    path = os.path.join(HERE, 'reference_behaviours.json')
    with open(path, 'w') as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1))


FORWARD_TILED_CASES = [
    # name, crop, stride, kwargs of forward_tiled (inputs_mask: built below)
    ('default', 96, 64, dict()),
    ('mask', 96, 64, dict(inputs_mask=True)),
    ('extra', 96, 64, dict(extra_keys=('classes', 'locations', 'fourier', 'contour_proposals'), extra_nms=dict(fourier=False))),
    ('params', 128, 80, dict(border_removal=2, min_box_size=9., nms_thresh=.05)),  # (tuple crops trip the method's own assert)
]


def gen_forward_tiled():
    """G12: the in-model tile loop pinned to the imported ``LitCpn.forward_tiled``."""
    import json
    from celldetection.models.lightning_cpn import LitCpn
    model, overrides, _ = build_ref_model('CpnU22', 0, fourier_std=.6, location_std=.5, score_shift=-3.)
    lit = LitCpn(model).eval()
    H, W = 160, 224
    # uint8 images: LitBase.forward -> prepare_inputs divides by 255 (lightning_base.py:774-780)
    img = torch.randint(0, 256, (2, 3, H, W), generator=torch.Generator().manual_seed(21), dtype=torch.uint8)
    mask = torch.zeros(2, 1, H, W)
    mask[0, :, :, :100] = 1      # image 0: left part only
    mask[1, :, 20:40, 30:50] = 1  # image 1: one blob in the first tile -> tiles empty in BOTH images are skipped, the others run
    out = {f'override.{k}': npy(v) for k, v in overrides.items()}
    tmpl = model.state_dict()
    out['sd_keys'] = np.array(list(tmpl.keys()))
    out['sd_shapes'] = np.array([','.join(str(int(d)) for d in v.shape) for v in tmpl.values()])
    out['img'], out['mask'] = npy(img), npy(mask)
    meta = []
    calls = []
    orig_forward = model.forward

    def recording_forward(inputs, *a, **k):
        y = orig_forward(inputs, *a, **k)
        calls.append((inputs.clone(), OrderedDict((kk, [t.clone() for t in v]) for kk, v in y.items() if v is not None)))
        return y

    model.forward = recording_forward
    try:
        for name, crop, stride, kw in FORWARD_TILED_CASES:
            kw = dict(kw)
            if kw.get('inputs_mask') is True:
                kw['inputs_mask'] = mask
            calls.clear()
            with torch.no_grad():
                y = lit.forward_tiled(img, crop_size=crop, stride=stride, **kw)
            slices, shape = cd.get_tiling_slices((H, W), crop, stride)
            slices = list(slices)
            called = []
            for ci, (x_in, y_tile) in enumerate(calls):  # which tile was forwarded: match the crop
                idx = next(i for i, sl in enumerate(slices)
                           if i not in called and torch.equal(img[(...,) + tuple(sl)].float() / 255, x_in))
                called.append(idx)
                for kk, v in y_tile.items():
                    for j, t in enumerate(v):
                        key = f't{crop}_{stride}.tile{idx}.{kk}.{j}'  # (per-tile outputs depend on the tiling only: stored once)
                        assert key not in out or np.array_equal(out[key], npy(t))
                        out[key] = npy(t)
            for kk, v in y.items():
                for j, t in enumerate(v):
                    out[f'{name}.final.{kk}.{j}'] = npy(t)
            meta.append(dict(name=name, crop=crop, stride=stride, tiles_called=called, n_tiles=len(slices), tiles=f't{crop}_{stride}',
                             kwargs={a: (list(b) if isinstance(b, tuple) else b) for a, b in kw.items() if a != 'inputs_mask'},
                             use_mask='inputs_mask' in kw, keys=list(y.keys())))
            print(f'forward_tiled/{name}: tiles forwarded {called} of {len(slices)}; final per image', [len(t) for t in y['scores']])
    finally:
        model.forward = orig_forward
    out['cases'] = np.array(json.dumps(meta))
    save('forward_tiled.npz', **out)


PREPROCESS_KW = [dict(grayscale=True), dict(gamma=.7), dict(gamma=2.2, grayscale=True), dict(contrast=1.3),
                 dict(contrast=.8, brightness=.2), dict(contrast=1., brightness=.5), dict(percentile=99.),
                 dict(percentile=(2., 98.), gamma=1.5), dict(percentile=99., gamma=1.5, contrast=1.2, brightness=-.1, grayscale=True)]


def preprocess_images():
    """name -> channels-last numpy image (the script's layout); seeded, re-creatable, also stored in the fixture."""
    rng = np.random.default_rng(2025)
    h, w = 29, 37
    imgs = {f'u8c{c}': rng.integers(0, 256, (h, w, c) if c else (h, w)).astype(np.uint8) for c in (0, 1, 3, 4)}
    imgs['u8low'] = rng.integers(0, 40, (h, w, 3)).astype(np.uint8)          # many ties at the percentile positions
    imgs['u16c3'] = rng.gamma(2., 900., (h, w, 3)).clip(0, 65535).astype(np.uint16)
    imgs['u16c0'] = rng.gamma(2., 300., (h, w)).clip(0, 65535).astype(np.uint16)
    imgs['f32c3'] = rng.standard_normal((h, w, 3)).astype(np.float32)
    imgs['f32c1'] = (rng.gamma(1.5, 1., (h, w, 1)) * 1e-3).astype(np.float32)
    return imgs


def gen_preprocess():
    """G11: pins SURVEY row f2.  normalize_percentile: every image x percentile {99.9, 99, (1, 99), (0, 100),
    (2.5, 60)} x to_uint8 {False (numpy only: fully pinned), True (through the img_as_ubyte stand-in)}; preprocess: the script's
    control flow (implicit normalisation of non-uint8 inputs, grayscale by channel count, GRAY2RGB, gamma, contrast / brightness
    only when contrast != 1) incl. the inputs it rejects."""
    import importlib
    import json
    importlib.import_module('celldetection_scripts.cpn_inference')
    script = sys.modules['celldetection_scripts.cpn_inference']
    imgs = preprocess_images()
    out = {f'img.{k}': v for k, v in imgs.items()}
    cases = []
    for name, img in imgs.items():
        for pi, pct in enumerate((99.9, 99, (1, 99), (0, 100), (2.5, 60))):
            for u8 in (False, True):
                key = f'np.{name}.{pi}.{int(u8)}'
                out[key] = cd.data.normalize_percentile(img.copy(), pct, to_uint8=u8)
                assert out[key].dtype == (np.uint8 if u8 else np.float64)
                cases.append(dict(fn='normalize_percentile', img=name, key=key, percentile=pct, to_uint8=u8))
    for name, img in imgs.items():
        for ki, kw in enumerate(PREPROCESS_KW + [dict()]):
            key = f'pp.{name}.{ki}'
            with warnings.catch_warnings(record=True) as wl:
                warnings.simplefilter('always')
                try:
                    res = script.preprocess(img.copy(), **kw)
                    err = None
                except Exception as e:  # (inputs the script rejects: recorded as such)
                    res, err = None, type(e).__name__
            if res is not None:
                out[key] = res
                assert res.dtype == np.uint8 and res.ndim == 3 and res.shape[-1] in (1, 3, 4)
            cases.append(dict(fn='preprocess', img=name, key=key, kwargs=kw, error=err,
                              warned=any('implicit percentile' in str(w_.message) for w_ in wl)))
    # the 2-channel grayscale branch: mean over the channels -> float64 -> cv2.cvtColor rejects the depth
    two = np.random.default_rng(1).integers(0, 256, (8, 9, 2)).astype(np.uint8)
    try:
        script.preprocess(two, grayscale=True)
        err = None
    except Exception as e:
        err = type(e).__name__
    cases.append(dict(fn='preprocess', img='two_channels', key=None, kwargs=dict(grayscale=True), error=err, warned=False))
    out['img.two_channels'] = two
    out['cases'] = np.array(json.dumps(cases))
    print('preprocess:', len(cases), 'cases;', sum(1 for c in cases if c.get('error')), 'rejected inputs')
    save('preprocess.npz', **out)


def gen_checkpoint():
    """G9: a model file written by the reference's OWN ``save_fetchable_model`` (util/util.py:545-560) -- tiny CpnU22 with one
    attribute changed after construction (-> ``updated_kwargs``) -- plus the reference's outputs for one input.  The file
    holds only builtins and tensors ({'cd.__version__', 'cd.models': {model, kwargs, updated_kwargs}, 'state_dict'})."""
    model, _, shape = build_ref_model('CpnU22', 0, score_shift=-1.5, fourier_std=.4, location_std=.4)
    model.score_thresh = .85
    model.samples = 24
    path = os.path.join(HERE, 'ref_checkpoint_CpnU22.pt')
    cd.save_fetchable_model(model, path, append_hash=False)
    x = torch.rand(1, *shape[1:], generator=torch.Generator().manual_seed(11))
    out = {'x': npy(x)}
    with torch.no_grad():
        flat_outputs('nms', model(x), out)
    print('checkpoint:', os.path.getsize(path) // 1024, 'KiB; detections', [len(t) for t in model(x)['scores']])
    save('ref_checkpoint_CpnU22_outputs.npz', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['ops', 'tiling', 'models', 'stitch', 'checkpoint', 'labels', 'preprocess', 'forward_tiled', 'behaviours']
    if 'forward_tiled' in which:
        gen_forward_tiled()
    if 'behaviours' in which:
        gen_behaviours()
    if 'preprocess' in which:
        gen_preprocess()
    if 'labels' in which:
        gen_labels()
    if 'checkpoint' in which:
        gen_checkpoint()
    if 'ops' in which:
        gen_ops()
    if 'tiling' in which:
        gen_tiling()
    if 'models' in which:
        for name_ in MODEL_SPECS:
            gen_model(name_)
    for name_ in which:  # single models by name
        if name_ in MODEL_SPECS:
            gen_model(name_)
    if 'stitch' in which:
        gen_stitch()
    if 'stitch' in which or 'stitch_dups' in which:
        gen_stitch_dups()
