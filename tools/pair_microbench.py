"""Micro-benchmark of the fused bottleneck head (csrc/conv_pair.hip) against the two conv launches it replaces.

    python tools/pair_microbench.py [case ...]      cases: l3 l2 l4 l2_256 l3_256
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from celldetection_amd import _lib, graph  # noqa: E402
from test_gpu_conv_pair import _pair_plan, _pad32  # noqa: E402

CASES = {  # ResNeXt101 32x8d stages at 16 x 512^2 (l2 / l3 / l4) and 16 x 256^2 tiles
    'l3': dict(n=16, h=32, w=32, cin=1024, cmid=1024, groups=32),
    'l2': dict(n=16, h=64, w=64, cin=512, cmid=512, groups=32),
    'l4': dict(n=16, h=16, w=16, cin=2048, cmid=2048, groups=32),
    'l1': dict(n=16, h=128, w=128, cin=256, cmid=256, groups=32),  # generic 16 x 32 tiles
    'l2s': dict(n=16, h=128, w=128, cin=256, cmid=512, groups=32, stride=2),   # first blocks of stages 2 / 3 / 4: stride-2 conv2
    'l3s': dict(n=16, h=64, w=64, cin=512, cmid=1024, groups=32, stride=2),
    'l4s': dict(n=16, h=32, w=32, cin=1024, cmid=2048, groups=32, stride=2),
    'l1_256': dict(n=16, h=64, w=64, cin=256, cmid=256, groups=32),
    'l2_256': dict(n=16, h=32, w=32, cin=512, cmid=512, groups=32),
    'l3_256': dict(n=16, h=16, w=16, cin=1024, cmid=1024, groups=32),
}


def main():
    dev = torch.device('cuda:0')
    lib = _lib.load()
    reps = int(os.environ.get('CPN_MB_REPS', '50'))
    for name in (sys.argv[1:] or list(CASES)):
        c = CASES[name]
        n, h, w, cin, cmid = c['n'], c['h'], c['w'], c['cin'], c['cmid']
        st = c.get('stride', 1)
        P, sd = _pair_plan(cin, cmid, c['groups'], 0, st)
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        tens, ops, wblob, bblob = graph.pack(P, sd, dev)
        x = torch.randn(n, h, w, _pad32(cin), device=dev).to(torch.bfloat16)
        mid = torch.empty(n, h, w, cmid, dtype=torch.bfloat16, device=dev)
        out = torch.empty(n, ho, wo, cmid, dtype=torch.bfloat16, device=dev)
        out2 = torch.empty_like(out)

        def fused():
            _lib.check(lib.cpn_conv_pair(ops[2], _lib.ptr(x), x.shape[-1], _lib.ptr(out), cmid, n, h, w, _lib.ptr(wblob),
                                         _lib.ptr(bblob), _lib.stream_ptr()))

        def conv(i, src, dst):
            _lib.check(lib.cpn_conv2d(ops[i], _lib.ptr(src), src.shape[-1], None, 0, None, 0, _lib.ptr(dst), cmid, n, h, w,
                                      _lib.ptr(wblob), _lib.ptr(bblob), _lib.stream_ptr()))

        def timed(fn):
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3

        t_f = timed(fused)
        t_1 = timed(lambda: conv(0, x, mid))
        t_2 = timed(lambda: conv(1, mid, out2))
        t_12 = timed(lambda: (conv(0, x, mid), conv(1, mid, out2)))
        same = torch.equal(out, out2)
        gf = 2. * n * cmid * (h * w * cin + ho * wo * cmid // c['groups'] * 9) / 1e9
        print(f'{name:8s} fused {t_f:7.1f} us   conv1 {t_1:6.1f} + conv2 {t_2:6.1f} = {t_1 + t_2:6.1f} (back to back {t_12:6.1f}) us   '
              f'x{t_12 / t_f:.2f}   {gf / t_f * 1e3:7.1f} TF/s algorithmic   identical={same}', flush=True)


if __name__ == '__main__':
    main()
