"""The slide-level steps around the tile loop on ONE GPU (the rows SURVEY section 8f calls "next"): percentile normalisation of
a 3 x S x S uint16 slide, tiled inference, label rasterisation of the result, HDF5 export.   python tools/slide_pipeline_bench.py [S]"""
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402
import celldetection_amd as cda  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device('cuda:0')
model, _ = build_model('CpnResNeXt101UNet', dev)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


raw = (torch.rand(3, S, S, generator=torch.Generator().manual_seed(3)) * 4095).to(torch.int32).to(torch.uint16).to(dev)
cda.preprocess.normalize_percentile(raw[:, :512, :512])  # warm-up
slide, t_pre = timed(lambda: cda.preprocess.normalize_percentile(raw, 99.9))
del raw
cda.inference.tiled_inference(model, slide[:, :1024, :1024], (512, 512), (384, 384), batch_size=16)
res, t_inf = timed(lambda: cda.inference.tiled_inference(model, slide, (512, 512), (384, 384), batch_size=16))
cda.contours2labels(res['contours'][:100], (S, S))
(labels, st), t_lab = timed(lambda: cda.contours2labels(res['contours'], (S, S), return_stats=True))
out = {k: v.cpu().numpy() for k, v in res.items()}
t0 = time.perf_counter()
path = os.path.join(tempfile.gettempdir(), 'slide_result.h5')
ok = cda.h5.hdf5_available()
if ok:
    cda.to_h5(path, **out, attributes=dict(contours=dict(args=json.dumps(dict(slide=S)))))
t_h5 = time.perf_counter() - t0
print(json.dumps(dict(slide=S, detections=int(res['scores'].numel()), normalize_percentile_s=t_pre, tiled_inference_s=t_inf,
                      contours2labels_s=t_lab, label_channels=st['channels'], label_rounds=st['rounds'],
                      to_h5_s=t_h5 if ok else None, h5_bytes=os.path.getsize(path) if ok else None)))
