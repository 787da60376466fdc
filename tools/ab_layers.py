"""In-process A/B of conv-kernel variants that are selected per launch through environment switches (CPN_PWR, CPN_RW, or a
switch added for an experiment): ONE model build, then the per-op profile of the conv graph under every setting (min over repeats), printed
per layer class and for the layers whose time changes.

    python tools/ab_layers.py [--model CpnResNeXt101UNet] [--batch 16] [--tile 512] "CPN_PWR=1" "CPN_RW=1" "CPN_PWR=1 CPN_RW=1"
    python tools/ab_layers.py --fresh "CPN_PAIR=0"        (switches read when a shape is planned need a fresh engine)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import celldetection_amd as cda  # noqa: E402
from celldetection_amd.synth import synth_state_dict  # noqa: E402


def klass(p):
    n = p['name']
    if p['op'] == 'conv_pair':
        return 'enc pair (conv1+conv2)'
    if p['op'] == 'conv_bridge':
        return 'dec 64ch'
    if p['op'] != 'conv':
        return 'helpers'
    if 'backbone.body' in n:
        return 'enc 1x1' if p['k'] == 1 else ('enc grouped' if p['groups'] > 1 else 'enc stem')
    if 'backbone' in n:
        if (p['cout'] or 0) <= 64:
            return 'dec 64ch'
        return 'dec 1x1' if p['k'] == 1 else ('dec .0 (cat/up)' if n.endswith('.0.') else 'dec .3')
    return 'head ref' if 'refinement' in n else 'heads 7x7'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='CpnResNeXt101UNet')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--tile', type=int, default=512)
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--fresh', action='store_true', help='re-create the engine per setting (switches that are read when a '
                                                         'shape is planned, e.g. CPN_PAIR)')
    ap.add_argument('settings', nargs='*')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    model = getattr(cda.models, args.model)(3)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev)
    x = torch.rand(args.batch, 3, args.tile, args.tile, generator=torch.Generator().manual_seed(1)).to(dev)
    eng = model.engine(dev)
    settings = [''] + list(args.settings)
    res = {}
    keys = sorted({kv.split('=')[0] for s in settings for kv in s.split() if kv})
    for rep in range(args.reps):  # interleaved: box drift hits every setting alike
        for s in settings:
            for k in keys:
                os.environ.pop(k, None)
            for kv in s.split():
                k, v = kv.split('=')
                os.environ[k] = v
            if args.fresh:
                model.repack()
                eng = model.engine(dev)
                eng.profile(x, model.core.order, True)  # (a fresh engine's first run pays one-time set-up)
            prof = eng.profile(x, model.core.order, True)
            cur = res.setdefault(s, prof)
            for a, b in zip(cur, prof):
                a['ms'] = min(a['ms'], b['ms'])
    base = res['']
    for s in settings:
        tot = {}
        for p in res[s]:
            tot[klass(p)] = tot.get(klass(p), 0.) + p['ms']
        print(f'### {s or "baseline"}: total {sum(tot.values()):.3f} ms   ' +
              '  '.join(f'{k} {v:.3f}' for k, v in sorted(tot.items())))
        if s:
            for a, b in zip(base, res[s]):
                if a['op'] == 'conv' and abs(a['ms'] - b['ms']) > .04 * a['ms'] and a['ms'] > .02:
                    print(f"    {a['index']:3d} k{a['k']} s{a['stride']} g{a['groups']:<2d} {a['cin']:5d}->{a['cout']:<5d} "
                          f"{a['ms']:7.3f} -> {b['ms']:7.3f} ms  {a['name']}")


if __name__ == '__main__':
    main()
