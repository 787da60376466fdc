"""Shader clock of the bf16 conv kernels while the conv graph of a CPN model runs, measured INSIDE the kernels: the probe library
libcpn_hip_clock.so (csrc/conv_igemm.hip with -DCPN_EXP_CLOCK=2, include/cpn_hip.h cpn_debug_clock_probe) lets one workgroup of every
conv launch time its main loop with s_memtime (shader clock) and s_memrealtime (100 MHz).  Prints one JSON line; bench.py runs this
as a child process and puts the line into roofline.shader_clock.

    python tools/clock_probe.py [K] [model] [batch] [tile]        default: 20 CpnResNeXt101UNet 16 512   (bf16, hipGraph replay)
"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('CPN_HIP_LIB', os.path.join(ROOT, 'celldetection_amd', 'libcpn_hip_clock.so'))
import torch  # noqa: E402

sys.path.insert(0, ROOT)
import celldetection_amd as cda  # noqa: E402
from celldetection_amd import _lib  # noqa: E402
from celldetection_amd.synth import synth_state_dict  # noqa: E402

NOMINAL_MHZ = 2400.        # MI355X peak engine clock (the 2.5 PFLOP/s bf16 figure is quoted at it)


def main():
    a = sys.argv[1:] + ['20', 'CpnResNeXt101UNet', '16', '512'][len(sys.argv) - 1:]
    K, name, batch, tile = int(a[0]), a[1], int(a[2]), int(a[3])
    dev = torch.device('cuda:0')
    model = getattr(cda.models, name)(3)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev)
    x = torch.rand(batch, 3, tile, tile, generator=torch.Generator().manual_seed(1)).to(dev)
    lib = _lib.load()
    buf = (ctypes.c_uint64 * 15)()
    for _ in range(10):  # warm-up: plan, hipGraph capture, the chip's power state
        model.core_forward(x)
    torch.cuda.synchronize()
    _lib.check(lib.cpn_debug_clock_probe(buf, 1))
    for _ in range(K):
        model.core_forward(x)
    torch.cuda.synchronize()
    _lib.check(lib.cpn_debug_clock_probe(buf, 0))
    v = [int(t) for t in buf]
    out = {'nominal_mhz': NOMINAL_MHZ, 'graph_executions': K, 'model': name, 'batch': batch, 'tile': tile,
           'method': 's_memtime (shader clock) / s_memrealtime (100 MHz) ticks over the main loop of one workgroup per conv launch, '
                     'summed per tap count (libcpn_hip_clock.so, cpn_debug_clock_probe)'}
    tot = [0, 0, 0, 0, 0]
    for i, key in enumerate(('conv7x7', 'conv3x3', 'other_taps')):
        core, ref, steps, launches, pipe4 = v[5 * i:5 * i + 5]
        tot = [t + u for t, u in zip(tot, (core, ref, steps, launches, pipe4))]
        if launches:
            out[key] = {'mhz': round(100. * core / ref, 1), 'frac_of_nominal': round(100. * core / ref / NOMINAL_MHZ, 4),
                        'cycles_per_step': round(core / steps, 1), 'matrix_pipe_duty': round(pipe4 / 4. / core, 4),
                        'launches': launches}
    if tot[3]:
        out['all_convs_mhz'] = round(100. * tot[0] / tot[1], 1)
        out['note'] = ('matrix_pipe_duty = cycles the MFMAs of the loop occupy on a SIMD (32 per 32x32x16 bf16 MFMA, all waves of the CU) / '
                       'shader cycles of the loop; duty x frac_of_nominal = the fraction of the nominal peak the main loop sustains')
    print(json.dumps(out))


if __name__ == '__main__':
    main()
