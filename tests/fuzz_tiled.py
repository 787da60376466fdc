"""Seeded random slides x tilings through the product slide loop (inference.tiled_inference: on-device crops, batched forward with
offsets, batched border rule, gather, global NMS) against the same per-tile detections stitched by the ORACLE's tiling table, border
rule and NMS (celldetection_scripts/cpn_inference.py:311-429) -- exact, like tests/test_gpu_slide.py, on geometries nobody wrote down:
slides smaller than a crop, strides equal to / smaller than the crop, ragged last tiles and batches, masks that skip tiles.

    python tests/fuzz_tiled.py [cases] [seed]
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import celldetection_amd as cda  # noqa: E402
import cpn_oracle as orc  # noqa: E402
from celldetection_amd import inference  # noqa: E402
from celldetection_amd.synth import calibrate_heads, synth_state_dict  # noqa: E402


def run(cases=30, seed=0):
    rng = random.Random(seed)
    dev = torch.device('cuda:0')
    model = cda.models.CpnU22(3, score_thresh=.6, backbone_kwargs={'backbone_kwargs': {'base_channels': 8}})
    sd = synth_state_dict(model.state_dict(), seed=3)
    xc = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(0))
    sd, _ = calibrate_heads(sd, lambda s_: orc.core_forward(s_, xc))
    model.load_state_dict(sd)
    model = model.to(dev)
    failed = 0
    for i in range(cases):
        H, W = rng.randrange(60, 420), rng.randrange(60, 420)
        crop = (rng.choice([64, 80, 96, 128, 160]), rng.choice([64, 96, 128, 160, 192]))
        stride = tuple(max(16, int(c * rng.choice([1., .75, .5, .8125]))) for c in crop)
        bs, border = rng.choice([1, 3, 4, 8]), rng.choice([0, 2, 4, 6])
        u8 = rng.random() < .5
        g = torch.Generator().manual_seed(i)
        slide = torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g) if u8 else torch.rand(3, H, W, generator=g)
        slide = slide.to(dev)
        use_mask = rng.random() < .3
        mask = None
        if use_mask:
            mask = torch.zeros(H, W, device=dev)
            y0, x0 = rng.randrange(0, H // 2), rng.randrange(0, W // 2)
            mask[y0:y0 + rng.randrange(8, H // 2), x0:x0 + rng.randrange(8, W // 2)] = 1.
        tag = f'[{i}] slide {H}x{W} {"u8" if u8 else "f32"} crop {crop} stride {stride} batch {bs} border {border} mask {use_mask}'
        try:
            model.sparse_heads = rng.choice(['auto', False])
            t = {}
            res = inference.tiled_inference(model, slide, crop_size=crop, strides=stride, batch_size=bs, border_removal=border,
                                            mask=mask, timings=t)
            # the oracle's loop over the same per-tile detections
            model.sparse_heads = False
            eff = (min(crop[0], H), min(crop[1], W))
            slices, overlaps, shape = orc.get_tiling_slices((H, W), eff, stride)
            coll = {}
            for idx, ((h0, h1), (w0, w1)) in enumerate(slices):
                kw = {}
                if mask is not None:
                    mc = mask[h0:h1, w0:w1]
                    if not bool(mc.any()):
                        continue
                    kw['scores_upper_bound'] = mc[None, None]
                offs = torch.tensor([[w0, h0]])
                y = model(slide[None, :, h0:h1, w0:w1], offsets=offs, **kw)
                h_i, w_i = np.unravel_index(idx, shape)
                con = y['contours'][0].cpu().numpy()
                keep = orc.remove_border_contours(con, (h1 - h0, w1 - w0), border, top=h_i > 0, right=w_i < shape[1] - 1,
                                                  bottom=h_i < shape[0] - 1, left=w_i > 0, offsets=-offs[0].numpy().astype(np.float32))
                for k in inference.KEYS:
                    v = y[k][0].cpu().numpy()[keep]
                    coll[k] = np.concatenate((coll[k], v)) if k in coll else v
            msgs = []
            if not coll:
                if len(res['scores']):
                    msgs.append(f'{len(res["scores"])} detections, the reference loop has none')
            else:
                keep = orc.nms(coll['boxes'], coll['scores'], model.nms_thresh)
                for k in inference.KEYS:
                    a, b = res[k].cpu().numpy(), coll[k][keep]
                    if a.shape != b.shape:
                        msgs.append(f'{k}: shape {a.shape} vs {b.shape}')
                    elif a.size and not np.array_equal(a, b):
                        msgs.append(f'{k}: max abs diff {float(np.abs(a.astype(np.float64) - b).max()):.3e}')
            if msgs:
                failed += 1
                print(tag, 'FAILED', '; '.join(msgs[:4]), flush=True)
            else:
                print(tag, 'ok', f'({len(slices)} tiles, {len(res["scores"])} detections)', flush=True)
        except Exception as e:
            failed += 1
            print(tag, f'ERROR {type(e).__name__}: {str(e)[:300]}', flush=True)
    print('fuzz_tiled:', cases, 'cases,', failed, 'failed')
    return failed


if __name__ == '__main__':
    sys.exit(1 if run(*(int(a) for a in sys.argv[1:3])) else 0)
