"""Seeded random models x input sizes through the whole conv graph against the fp32 CPU oracle (test infrastructure: this is a test
tool, the oracle is the checker): reduced-width CPN models of every backbone family, batch 1 - 3, heights and widths that are
multiples of nothing.  The fp32 verification path must reproduce the oracle's head maps (shape and values), the bf16 product path
must stay within bf16 distance, forward() must return the oracle's post-processing of the HIP maps.

    python tests/fuzz_model.py [cases] [seed]
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
from celldetection_amd.synth import calibrate_heads, synth_state_dict  # noqa: E402

FAMILIES = {
    'CpnU22': lambda r: dict(backbone_kwargs={'backbone_kwargs': {'base_channels': r.choice([8, 16])}}),
    'CpnSlimU22': lambda r: {},
    'CpnResUNet': lambda r: dict(backbone_kwargs={'backbone_kwargs': {'base_channels': r.choice([8, 16])}}),
    'CpnResNet18FPN': lambda r: dict(backbone_kwargs={'fpn_channels': r.choice([16, 32]), 'backbone_kwargs': {'base_channel': 8}}),
    'CpnResNet50FPN': lambda r: dict(backbone_kwargs={'fpn_channels': 32, 'backbone_kwargs': {'base_channel': 8}}),
    'CpnResNet18UNet': lambda r: dict(backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}),
    'CpnResNet34UNet': lambda r: dict(backbone_kwargs={'backbone_kwargs': {'base_channel': 8}}),
    'CpnResNeXt101UNet': lambda r: dict(backbone_kwargs={'backbone_kwargs': {'base_channel': 32}}),
}
NAMES = ('scores', 'locations', 'refinement', 'fourier')


def run(cases=40, seed=0):
    rng = random.Random(seed)
    dev = torch.device('cuda:0')
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    failed = 0
    worst = {}
    for i in range(cases):
        fam = rng.choice(list(FAMILIES))
        kw = FAMILIES[fam](rng)
        n, h, w = rng.choice([1, 1, 2, 3]), rng.randrange(33, 150), rng.randrange(33, 190)
        if rng.random() < .3:
            h, w = h // 16 * 16 + 16, w // 32 * 32 + 32
        tag = f'[{i}] {fam} {kw} x {n}x3x{h}x{w}'
        try:
            model = getattr(cda.models, fam)(3, score_thresh=rng.choice([.5, .6]), **kw)
            sd = synth_state_dict(model.state_dict(), seed=rng.randrange(1 << 30))
            x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(i))
            # heads rescaled to sensible ranges on the oracle (like the golden fixtures and the smoke test): with raw random heads the
            # score logits have a std of tens, and bf16's 1 - 2 % of that is an O(1) logit error -- a property of the weights
            sd, _ = calibrate_heads(sd, lambda s_: orc.core_forward(s_, x))
            model.load_state_dict(sd)
            model = model.to(dev)
            ref = orc.core_forward(sd, x)
            ref = (torch.sigmoid(ref[0]),) + tuple(ref[1:4])
            msgs = []
            # fp32 path: the exactness check.  bf16: a sanity bound -- the error of a reduced-width deep model with synthetic weights has
            # a wide distribution over seeds (0.4 - 6 %, rare draws 8 - 13 % on the 8-channel ResNet-UNets) that does not depend on the
            # engine's decompositions (same numbers with subpixel = False, CPN_BRIDGE=0, CPN_HOIST=0: tests/bf16_error_switches.py)
            for prec, bound in (('fp32', 2e-4), ('bf16', 2e-1)):
                model.precision = prec
                got = model.core_forward(x.to(dev))
                for name, g_, e_ in zip(NAMES, got, ref):
                    if g_ is None or e_ is None:
                        continue
                    if g_.shape != e_.shape:
                        msgs.append(f'{prec} {name}: shape {tuple(g_.shape)} vs {tuple(e_.shape)}')
                        continue
                    rel = ((g_.cpu().float() - e_).norm() / (e_.norm() + 1e-12)).item()
                    worst[(prec, name)] = max(worst.get((prec, name), 0.), rel)
                    if not np.isfinite(rel) or rel > bound:
                        msgs.append(f'{prec} {name}: rel L2 {rel:.3e} > {bound}')
            # forward() == the oracle's post-processing of the HIP maps (fp32 path: decode / refinement / NMS are exact restatements)
            model.precision = 'fp32'
            maps = model.core_forward(x.to(dev))
            y = model.postprocess(*maps, (h, w))
            exp = orc.cpn_postprocess(*[m.cpu() for m in maps], input_size=(h, w), scores_are_probabilities=True,
                                      score_thresh=model.score_thresh)
            for k in ('contours', 'boxes', 'scores'):
                for j, (a, b) in enumerate(zip(y[k], exp[k])):
                    if tuple(a.shape) != tuple(np.asarray(b).shape):
                        msgs.append(f'post {k}[{j}]: shape {tuple(a.shape)} vs {np.asarray(b).shape}')
                    elif a.numel() and float(np.abs(a.cpu().numpy() - np.asarray(b)).max()) > 1e-4:
                        msgs.append(f'post {k}[{j}]: max abs diff {float(np.abs(a.cpu().numpy() - np.asarray(b)).max()):.3e}')
            dets = sum(len(s) for s in y['scores'])
            if msgs:
                failed += 1
                print(tag, 'FAILED', '; '.join(msgs[:6]), flush=True)
            else:
                print(tag, 'ok', f'({dets} detections)', flush=True)
        except Exception as e:
            failed += 1
            print(tag, f'ERROR {type(e).__name__}: {str(e)[:300]}', flush=True)
    print('fuzz_model:', cases, 'cases,', failed, 'failed; worst rel L2', {f'{k[0]}.{k[1]}': f'{v:.2e}' for k, v in sorted(worst.items())})
    return failed


if __name__ == '__main__':
    sys.exit(1 if run(*(int(a) for a in sys.argv[1:3])) else 0)
