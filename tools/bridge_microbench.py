"""Micro-benchmark of the fused bridge level (csrc/conv_igemm.hip MODE_BR) against the two conv launches it replaces, at the
flagship's shape (64 -> 64 channels, 16 x 256^2 -> 512^2).

    python tools/bridge_microbench.py          (CPN_HIP_LIB=<variant .so> for tuning ablations, tools/build_variant.sh)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from celldetection_amd import _lib, graph  # noqa: E402
from test_gpu_conv_bridge import _bridge_plan  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    lib = _lib.load()
    reps = int(os.environ.get('CPN_MB_REPS', '50'))
    n, h, w, cin = 16, 256, 256, 64
    P, sd = _bridge_plan(cin, 0)
    tens, ops, wblob, bblob = graph.pack(P, sd, dev)
    x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
    mid = torch.empty(n, 2 * h, 2 * w, 64, dtype=torch.bfloat16, device=dev)
    out, out2 = torch.empty_like(mid), torch.empty_like(mid)

    def fused():
        _lib.check(lib.cpn_conv_bridge(ops[2], _lib.ptr(x), cin, None, 0, _lib.ptr(out), 64, n, h, w, _lib.ptr(wblob),
                                       _lib.ptr(bblob), _lib.stream_ptr()))

    def conv(i, src, dst, hh, ww):
        _lib.check(lib.cpn_conv2d(ops[i], _lib.ptr(src), 64, None, 0, None, 0, _lib.ptr(dst), 64, n, hh, ww, _lib.ptr(wblob),
                                  _lib.ptr(bblob), _lib.stream_ptr()))

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
    t_1 = timed(lambda: conv(0, x, mid, h, w))
    t_2 = timed(lambda: conv(1, mid, out2, 2 * h, 2 * w))
    t_12 = timed(lambda: (conv(0, x, mid, h, w), conv(1, mid, out2, 2 * h, 2 * w)))
    same = torch.equal(out, out2)
    gb = (x.numel() + out.numel()) * 2 / 1e9
    print(f'bridge 16 x 256^2 -> 512^2: fused {t_f:7.1f} us ({gb / t_f * 1e6 / 1e3:.2f} TB/s of its {gb:.2f} GB)   scatter {t_1:6.1f} + '
          f'3x3 {t_2:6.1f} = {t_1 + t_2:6.1f} (back to back {t_12:6.1f}) us   x{t_12 / t_f:.2f}   identical={same}', flush=True)


if __name__ == '__main__':
    main()
