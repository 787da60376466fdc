"""Times the post-processing (compaction + decode + refinement + NMS + gathers) of the bench workload on fixed head maps."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402

dev = torch.device('cuda:0')
model, _ = build_model('CpnResNeXt101UNet', dev)
x = torch.rand(16, 3, 512, 512, generator=torch.Generator().manual_seed(100)).to(dev)
maps = model.core_forward(x)
for _ in range(3):
    y = model.postprocess(*maps, (512, 512))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    y = model.postprocess(*maps, (512, 512))
torch.cuda.synchronize()
print('postprocess %.3f ms per batch of 16 (%d detections)' % ((time.perf_counter() - t0) / 20 * 1e3,
                                                              sum(len(s) for s in y['scores'])))
