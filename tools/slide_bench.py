"""BASELINE.json configs[3] on ONE GPU: CpnResNeXt101UNet tiled inference over a synthetic 1x3xSxS uint8 slide
(tiles 512 / stride 384, batch 16): tiles/s of the whole slide loop incl. cropping, u8->bf16 conversion, border
removal, global NMS.   python tools/slide_bench.py [S=16384]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402
from celldetection_amd import inference, util  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device('cuda:0')
model, _ = build_model('CpnResNeXt101UNet', dev)
slide = torch.randint(0, 256, (3, S, S), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).to(dev)
ntiles = len(list(util.get_tiling_slices((S, S), (512, 512), (384, 384))[0]))
inference.tiled_inference(model, slide[:, :1024, :1024], (512, 512), (384, 384), batch_size=16)  # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
res = inference.tiled_inference(model, slide, (512, 512), (384, 384), batch_size=16)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'slide {S}x{S}: {ntiles} tiles in {dt:.2f} s = {ntiles / dt:.1f} tiles/s, {res["scores"].numel()} detections '
      f'after the global NMS')
