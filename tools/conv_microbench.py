"""Single-layer micro-benchmark of the implicit-GEMM conv kernel (for rocprofv3 --pmc runs and A/B tuning).

    python tools/conv_microbench.py [case ...]      cases: head7 dec3 dec3b pw1024 ref7 grp
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celldetection_amd import _lib, graph  # noqa: E402

CASES = {
    'head7': dict(n=16, h=256, w=256, cin=256, cout=256, k=7),
    'dec3': dict(n=16, h=64, w=64, cin=1024, cout=1024, k=3),
    'dec3b': dict(n=16, h=256, w=256, cin=256, cout=256, k=3),
    'dec3cat': dict(n=16, h=128, w=128, cin=256, cin1=512, cout=512, k=3),
    'pw1024': dict(n=16, h=32, w=32, cin=1024, cout=1024, k=1),
    'pw256': dict(n=16, h=128, w=128, cin=256, cout=256, k=1),
    'pw512': dict(n=16, h=64, w=64, cin=512, cout=512, k=1),
    'pw2048': dict(n=16, h=16, w=16, cin=2048, cout=2048, k=1),
    'ref7': dict(n=16, h=512, w=512, cin=64, cout=64, k=7),
    'c64': dict(n=16, h=512, w=512, cin=64, cout=64, k=3),
    'grp': dict(n=16, h=32, w=32, cin=1024, cout=1024, k=3, groups=32),
    'head7x3': dict(n=16, h=256, w=256, cin=256, cout=768, k=7),  # the three contour heads as ONE 256 -> 768 launch
    'bl7': dict(n=8, h=512, w=512, cin=256, cout=256, k=7, bilinear=True),  # FPN refinement head: bilinear-resized source
    'k3': dict(n=8, h=256, w=256, cin=256, cout=256, k=3),  # tap count at fixed shape (bilinear phase convs are k = 5)
    'k5': dict(n=8, h=256, w=256, cin=256, cout=256, k=5),
    'k7': dict(n=8, h=256, w=256, cin=256, cout=256, k=7),
    'k9': dict(n=8, h=256, w=256, cin=256, cout=256, k=9),
    'bl7s': dict(n=2, h=256, w=256, cin=256, cout=256, k=7, bilinear=True),
}


ZERO = bool(int(os.environ.get('CPN_MB_ZERO', '0')))  # all-zero operands: DVFS ceiling probe (power vs issue bound)


FP8 = bool(int(os.environ.get('CPN_MB_FP8', '0')))   # e4m3 kernel (cpn_conv2d_fp8) instead of bf16


def run_fp8(name, P, sd, cfg, reps):
    dev = torch.device('cuda:0')
    n, h, w, cin, cout, k = (cfg[x] for x in ('n', 'h', 'w', 'cin', 'cout', 'k'))
    cin1, groups = cfg.get('cin1', 0), cfg.get('groups', 1)
    p64 = lambda c: (c + 63) // 64 * 64
    scales = {i: 1. / 64 for i in range(len(P.tensors))}
    tens, ops, wblob, bblob, mblob, op_scales = graph.pack(P, sd, dev, precision='fp8', act_scales=scales)

    def codes(*shape):
        return torch.randn(*shape, device=dev).mul_(64).clamp_(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)

    bl = cfg.get('bilinear', False)
    x0 = codes(n, h // 2 if bl else h, w // 2 if bl else w, p64(cin))
    x1 = codes(n, h // 2, w // 2, p64(cin1)) if cin1 else None
    dst = torch.empty(n, h, w, p64(cout), dtype=torch.uint8, device=dev)
    lib = _lib.load()

    def launch():
        _lib.check(lib.cpn_conv2d_fp8(ops[0], _lib.ptr(x0), x0.shape[-1], _lib.ptr(x1), 0 if x1 is None else x1.shape[-1],
                                      _lib.ptr(None), 0, _lib.ptr(dst), dst.shape[-1], n, h, w, _lib.ptr(wblob),
                                      _lib.ptr(bblob), _lib.ptr(mblob), 0., 64., _lib.stream_ptr()))

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = 2. * n * h * w * cout * ((cin + cin1) // groups) * k * k / 1e9
    print(f'{name:8s} {ms:8.3f} ms  {gf / ms:8.1f} TF/s  ({gf:.1f} GF)  [fp8]', flush=True)


def run(name, reps=20):
    cfg = dict(CASES[name])
    dev = torch.device('cuda:0')
    n, h, w, cin, cout, k = (cfg[x] for x in ('n', 'h', 'w', 'cin', 'cout', 'k'))
    cin1, groups = cfg.get('cin1', 0), cfg.get('groups', 1)
    P = graph.Plan()
    s0 = P.tensor(cin, 1)
    s1 = P.tensor(cin1, 2) if cin1 else None
    bl = cfg.get('bilinear', False)
    if bl:
        P.tensors[s0]['down'] = 2
    P.conv(s0, cout, k, w='c.', bn=None, bias=True, act='relu', groups=groups, src1=s1, up1=bool(cin1),
           up0='bilinear' if bl else False)
    sd = {'c.weight': torch.randn(cout, (cin + cin1) // groups, k, k) * .05, 'c.bias': torch.randn(cout) * .1}
    if ZERO:
        sd = {k_: torch.zeros_like(v_) for k_, v_ in sd.items()}
    if FP8:
        return run_fp8(name, P, sd, cfg, reps)
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    p32 = lambda c: (c + 31) // 32 * 32
    x0 = torch.randn(n, h // 2 if bl else h, w // 2 if bl else w, p32(cin), device=dev).to(torch.bfloat16)
    x1 = torch.randn(n, h // 2, w // 2, p32(cin1), device=dev).to(torch.bfloat16) if cin1 else None
    if ZERO:
        x0.zero_()
        if x1 is not None:
            x1.zero_()
    dst = torch.empty(n, h, w, p32(cout), dtype=torch.bfloat16, device=dev)
    lib = _lib.load()

    def launch():
        _lib.check(lib.cpn_conv2d(ops[0], _lib.ptr(x0), x0.shape[-1], _lib.ptr(x1), 0 if x1 is None else x1.shape[-1],
                                  _lib.ptr(None), 0, _lib.ptr(dst), dst.shape[-1], n, h, w, _lib.ptr(wblob),
                                  _lib.ptr(bblob), _lib.stream_ptr()))

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gf = 2. * n * h * w * cout * ((cin + cin1) // groups) * k * k / 1e9
    print(f'{name:8s} {ms:8.3f} ms  {gf / ms:8.1f} TF/s  ({gf:.1f} GF)', flush=True)


if __name__ == '__main__':
    for c in (sys.argv[1:] or list(CASES)):
        run(c)
