"""Where a small slide's time goes: host-side hand-over times of every batch of the tile loop (no extra synchronisation).

    python tools/slide_profile.py [S=4096] [gated=0|1]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402
from celldetection_amd import inference  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device('cuda:0')
model, _ = build_model('CpnResNeXt101UNet', dev)
model.sparse_heads = 'auto' if (len(sys.argv) > 2 and sys.argv[2] == '1') else False
slide = torch.randint(0, 256, (3, S, S), dtype=torch.uint8, device=dev, generator=torch.Generator(dev).manual_seed(3))
kw = dict(crop_size=(512, 512), strides=(384, 384), batch_size=16)
for _ in range(2):
    inference.tiled_inference(model, slide[:, :2048, :2048], **kw)
torch.cuda.synchronize()
for rep in range(3):
    t = {}
    t0 = time.perf_counter()
    res = inference.tiled_inference(model, slide, timings=t, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    b = t.pop('batch_done_s')
    gaps = [b[0]] + [b[i] - b[i - 1] for i in range(1, len(b))]
    print(f'rep {rep}: {dt * 1e3:.1f} ms total; tiles {t["tiles"] * 1e3:.1f} gather {t["gather"] * 1e3:.2f} nms {t["nms"] * 1e3:.2f} ms; '
          f'batch hand-over gaps (ms): ' + ' '.join(f'{g * 1e3:.1f}' for g in gaps) +
          f'; after the last batch: {(t["tiles"] - b[-1]) * 1e3:.1f} ms', flush=True)
