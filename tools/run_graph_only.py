"""Runs ONLY the conv graph of a CPN model K times (synthetic weights; no calibration, no post-processing) -- used under
rocprofv3 to attribute kernel time / PMC counters per graph execution.  Launches are eager (CPN_HIP_GRAPH=0) so that every
execution is the same sequence of dispatches.

    python tools/run_graph_only.py [K] [model] [batch] [tile] [precision]      default: 5 CpnResNeXt101UNet 16 512 bf16
"""
import os
import sys

os.environ['CPN_HIP_GRAPH'] = '0'
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celldetection_amd as cda  # noqa: E402
from celldetection_amd.synth import synth_state_dict  # noqa: E402

a = sys.argv[1:] + ['5', 'CpnResNeXt101UNet', '16', '512', 'bf16'][len(sys.argv) - 1:]
K, name, batch, tile, precision = int(a[0]), a[1], int(a[2]), int(a[3]), a[4]
dev = torch.device('cuda:0')
model = getattr(cda.models, name)(3)
model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
model = model.to(dev)
x = torch.rand(batch, 3, tile, tile, generator=torch.Generator().manual_seed(1)).to(dev)
if precision == 'fp8':
    model.precision = 'fp8'
    model.calibrate_fp8(x[:2])  # (one bf16 run on two tiles ahead of the K counted executions: ~1/8 of a graph's traffic)
for _ in range(K):
    model.core_forward(x)
torch.cuda.synchronize()
print('graph executions:', K)
