import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, ROOT)
import cpn_oracle as orc
from bench import build_model, _cpu_quota
print('quota', _cpu_quota(), 'cpu_count', os.cpu_count())
model, sd = build_model('CpnResNeXt101UNet', torch.device('cuda:0'))
sd = {k: v.detach().cpu() for k, v in sd.items()}
x = torch.rand(2, 3, 512, 512, generator=torch.Generator().manual_seed(2))
for th in (8, 12, 16, 24, 32, 48):
    torch.set_num_threads(th)
    orc.cpn_forward(sd, x[:1])
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); orc.cpn_forward(sd, x); best = min(best, time.perf_counter() - t0)
    print(f'threads {th}: {2 / best:.3f} tiles/s', flush=True)
