"""Seeded random shapes through the single-conv entry point (cpn_conv2d) against the fp32 conv of the same bf16-rounded operands --
the checker and tolerances of tests/test_gpu_kernels.py::test_conv, on shapes nobody wrote down: ragged tiles, odd channel counts,
two sources with / without the x2 nearest upsample, residuals, strides, groups, fused ReadOut tails, bilinear sources, narrow maps.

    python tests/fuzz_conv.py [cases] [seed]        prints one line per failure and a summary; exit code 1 if anything failed
"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))  # (this file lives there: run_conv is the kernel tests' helper)
from test_gpu_kernels import run_conv  # noqa: E402


def sample(rng):
    k = rng.choice([1, 1, 3, 3, 3, 5, 7, 7])
    cfg = dict(k=k, n=rng.choice([1, 1, 2, 3, 5]), seed=rng.randrange(1 << 30))
    kind = rng.choice(['plain', 'plain', 'concat', 'res', 'stride2', 'grouped', 'fused', 'bilinear', 'narrow', 'f32out'])
    cfg['h'] = rng.choice([8, 12, 16, 24, 32, 40, 44, 64, 72, 96])
    cfg['w'] = rng.choice([16, 24, 32, 40, 48, 64, 72, 96, 136])
    cfg['cin'] = rng.choice([8, 16, 24, 32, 40, 64, 96, 128, 160, 256])
    cfg['cout'] = rng.choice([8, 16, 32, 48, 64, 96, 128, 192, 256, 320, 512])
    cfg['act'] = rng.choice(['relu', 'relu', 'none'])
    cfg['bias'] = rng.random() < .7
    cfg['bn'] = rng.random() < .7
    if kind == 'concat':
        cfg['cin1'] = rng.choice([16, 32, 64, 96])
        cfg['up1'] = rng.random() < .6
        cfg['up0'] = (not cfg['up1']) and rng.random() < .2
        cfg['cin'] = rng.choice([32, 64, 96])  # (concat sources are 32-channel aligned in the plans)
    elif kind == 'res':
        cfg['res'] = True
        cfg['res_up'] = k > 1 and rng.random() < .3
    elif kind == 'stride2':
        cfg['stride'] = 2
        cfg['k'] = rng.choice([1, 3, 7])
    elif kind == 'grouped':
        cfg['k'] = 3
        g = rng.choice([2, 4, 8, 32])
        cfg['cin'] = cfg['cout'] = g * rng.choice([8, 16, 32])
        cfg['groups'] = g
    elif kind == 'fused':
        cfg['k'] = rng.choice([3, 5, 7])
        cfg['cout'] = rng.choice([8, 64, 128, 256])
        cfg['fuse_cout'] = rng.choice([1, 2, 20])
        cfg['fuse_act'] = rng.choice(['none', 'sigmoid', 'tanh_scaled'])
        cfg['act'] = 'relu'
    elif kind == 'bilinear':
        cfg['k'] = rng.choice([3, 5, 7])
        cfg['bilinear'] = True
        cfg['h'] = rng.choice([16, 32, 48, 64])
        cfg['w'] = rng.choice([32, 64, 96])
    elif kind == 'narrow':
        cfg['w'] = 16
        cfg['h'] = rng.choice([8, 16, 24, 32])
        cfg['k'] = rng.choice([3, 5, 7])
    elif kind == 'f32out':
        cfg['k'] = rng.choice([1, 3, 5])
        cfg['cout'] = rng.choice([1, 2, 3, 20])
        cfg['bn'] = False
        cfg['act'] = rng.choice(['none', 'sigmoid', 'tanh_scaled'])
        cfg['out_f32'] = True
    if cfg.get('up0') or cfg.get('up1') or cfg.get('res_up') or cfg.get('bilinear'):
        cfg['h'] += cfg['h'] % 2
        cfg['w'] += cfg['w'] % 2
    return kind, cfg


def check(name, cfg):
    got, ref, f32 = run_conv(torch.device('cuda:0'), **cfg)
    if got.shape != ref.shape or not torch.isfinite(got).all():
        return f'shape {tuple(got.shape)} vs {tuple(ref.shape)}, non-finite {(~torch.isfinite(got)).sum().item()}'
    err = (got - ref).abs()
    scale = ref.abs().max().item() + 1e-6
    tol = (2e-3 if f32 else 1e-2) * max(scale, 1.)
    bad = (err > tol + (0 if f32 else 8e-3) * ref.abs()).sum().item()
    # fused tails round the hidden activation to bf16: a value on a rounding boundary may round the other way than in the reference,
    # and ONE flipped hidden unit moves all fuse_cout outputs of its pixel (tanh_scaled x 3 on top)
    allowed = max(int(cfg['fuse_cout']), int(2e-4 * err.numel())) if cfg.get('fuse_cout') else 0
    if bad > allowed or (allowed and err.max().item() >= 5e-2 * max(scale, 1.)):
        return f'{bad} / {err.numel()} elements off; max abs err {err.max().item():.4e}, ref max {scale:.3e}'
    return None


def sample_fp8(rng):
    """Shapes for the e4m3 kernel (cpn_conv2d_fp8): the checker is tests/test_gpu_kernels.py::test_conv_fp8_vs_dequantised_reference
    (fp32 conv on the same e4m3-quantised operands)."""
    k = rng.choice([1, 3, 3, 5, 7])
    cfg = dict(k=k, n=rng.choice([1, 2, 3]), h=rng.choice([16, 24, 32, 40, 64]), w=rng.choice([32, 48, 64, 96]),
               cin=rng.choice([3, 24, 64, 128, 192, 256]), cout=rng.choice([40, 64, 128, 256]))
    kind = rng.choice(['plain', 'plain', 'res', 'stride2', 'concat', 'grouped', 'f32out', 'fused'])
    if kind == 'res':
        cfg['res'] = True
    elif kind == 'stride2':
        cfg['stride'] = 2
        cfg['k'] = rng.choice([1, 3, 7])
    elif kind == 'concat':
        cfg.update(cin=rng.choice([64, 128]), cin1=rng.choice([64, 128]), up1=rng.random() < .6, k=3)
        cfg['h'] += cfg['h'] % 2
    elif kind == 'grouped':
        g_ = rng.choice([4, 32])
        cfg.update(k=3, groups=g_, cin=g_ * 8, cout=g_ * 8)
    elif kind == 'f32out':
        cfg.update(k=1, cout=rng.choice([1, 2, 20]), bn=False, act='none', out_f32=True)
    elif kind == 'fused':
        cfg.update(k=rng.choice([3, 7]), cout=rng.choice([64, 256]), fuse_cout=rng.choice([1, 2, 20]), fuse_act='none')
    return kind, cfg


def run_fp8(cases=100, seed=0):
    import test_gpu_kernels as tk
    rng = random.Random(seed)
    dev = torch.device('cuda:0')
    failed, kinds = 0, {}
    for i in range(cases):
        kind, cfg = sample_fp8(rng)
        kinds[kind] = kinds.get(kind, 0) + 1
        tk.FP8_CASES['_fuzz'] = cfg
        try:
            tk.test_conv_fp8_vs_dequantised_reference(dev, '_fuzz')
        except Exception as e:
            failed += 1
            print(f'[{i}] fp8 {kind} FAILED {type(e).__name__}: {str(e)[:200]}  {cfg}', flush=True)
    tk.FP8_CASES.pop('_fuzz', None)
    print(f'fuzz_conv (e4m3): {cases} cases {kinds}, {failed} failed')
    return failed


def run(cases=200, seed=0):
    rng = random.Random(seed)
    failed, errors, kinds = 0, 0, {}
    for i in range(cases):
        kind, cfg = sample(rng)
        kinds[kind] = kinds.get(kind, 0) + 1
        try:
            msg = check(kind, cfg)
        except Exception as e:  # a shape the plan / launcher rejects is reported, not counted as a numerical failure
            errors += 1
            print(f'[{i}] {kind} REJECTED {type(e).__name__}: {str(e)[:160]}  {cfg}', flush=True)
            continue
        if msg:
            failed += 1
            print(f'[{i}] {kind} FAILED {msg}  {cfg}', flush=True)
    print(f'fuzz_conv: {cases} cases {kinds}, {failed} failed, {errors} rejected')
    return failed


if __name__ == '__main__':
    a_ = [int(a) for a in sys.argv[1:3]]
    sys.exit(1 if run(*a_) + run_fp8(max(20, (a_[0] if a_ else 200) // 3), a_[1] if len(a_) > 1 else 0) else 0)
