"""Score-gated heads micro-benchmark: cpn_sparse_heads at several proposal counts vs the two dense fused head convs on the
same feature tensor (sets ops.SPARSE_HEADS_MAX_DENSITY).

    python tools/sparse_microbench.py [n h w cin hid k]      default: 16 256 256 256 256 7  (BASELINE configs[2] head grid)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celldetection_amd import _lib, graph, ops  # noqa: E402


def main():
    n, h, w, cin, hid, k = (int(v) for v in (sys.argv[1:7] + ['16', '256', '256', '256', '256', '7'][len(sys.argv) - 1:]))
    dev = torch.device('cuda:0')
    P = graph.Plan()
    x = P.tensor(cin, 1)
    for prefix, cout, oi in (('a.', 2, _lib.OUT_LOCATIONS), ('b.', 20, _lib.OUT_FOURIER)):
        P.conv(x, hid, k, w=prefix + 'block.0.', bn=prefix + 'block.1.', bias=True, act='relu', out_index=oi,
               fuse=dict(w=prefix + 'block.4.', cout=cout, act='none', act_scale=0.))
    g = torch.Generator().manual_seed(0)
    sd = {}
    for key, shape, kind in P.entries:
        if key.endswith('running_var'):
            sd[key] = torch.rand(shape, generator=g) + .5
        elif kind == 'long':
            sd[key] = torch.zeros(shape, dtype=torch.long)
        elif len(shape) == 4:
            sd[key] = torch.randn(shape, generator=g) / np.sqrt(np.prod(shape[1:]))
        else:
            sd[key] = torch.randn(shape, generator=g) * .5 + (1. if key.endswith('1.weight') else 0.)
    tens, opd, wblob, bblob = graph.pack(P, sd, dev)
    cs = tens[0].channels
    feat = torch.randn(n, h, w, cs, device=dev).relu_().to(torch.bfloat16)
    src = (feat.data_ptr(), cs, (n, h, w))

    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    total = n * h * w
    gf_px = 2. * 2 * hid * cin * k * k / 1e9  # both heads, first conv, per pixel
    td = timed(lambda: (ops.dense_head(opd[0], *src, wblob, bblob), ops.dense_head(opd[1], *src, wblob, bblob)))
    print(f'dense   {total:9d} px   {td:8.3f} ms   {gf_px * total / td:8.1f} TF/s')
    for frac in (.001, .01, .05, .1, .25, .5, 1.):
        p = max(1, int(total * frac))
        idx = torch.randperm(total, generator=g)[:p].sort().values.to(torch.int32).to(dev)
        ts = timed(lambda: ops.sparse_heads(opd[0], opd[1], *src, idx, wblob, bblob))
        print(f'sparse  {p:9d} prop {ts:8.3f} ms   {gf_px * p / ts:8.1f} TF/s   density {frac:5.3f}   vs dense x{td / ts:6.2f}')


if __name__ == '__main__':
    main()
