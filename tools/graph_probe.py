"""Probe: conv-graph execution eager (122 back-to-back launches on the stream) vs hipGraph replay of the same plan."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celldetection_amd as cda  # noqa: E402
from celldetection_amd.synth import synth_state_dict  # noqa: E402

dev = torch.device('cuda:0')
model = cda.models.CpnResNeXt101UNet(3)
model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
model = model.to(dev)
x = torch.rand(16, 3, 512, 512, generator=torch.Generator().manual_seed(1)).to(dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print('eager  %.3f ms' % timeit(lambda: model.core_forward(x)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    model.core_forward(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = model.core_forward(x)
torch.cuda.synchronize()
print('graph  %.3f ms' % timeit(g.replay))
