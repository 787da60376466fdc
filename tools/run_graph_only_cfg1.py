"""Runs ONLY the conv graph of CpnResNet18FPN (BASELINE configs[1]: batch 8 x 3x512x512, synthetic weights) K times --
used under rocprofv3 (kernel trace / FETCH_SIZE / WRITE_SIZE passes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celldetection_amd as cda  # noqa: E402
from celldetection_amd.synth import synth_state_dict  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda:0')
model = cda.models.CpnResNet18FPN(3)
model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
model = model.to(dev)
x = torch.rand(8, 3, 512, 512, generator=torch.Generator().manual_seed(1)).to(dev)
for _ in range(K):
    model.core_forward(x)
torch.cuda.synchronize()
print('graph executions:', K)
