"""Builds libcpn_hip.so (hand-written HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m celldetection_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the repo snapshot to the
GPU box.  Only gfx950 (MI355X / CDNA4) is targeted -- no multi-arch, no compatibility paths.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libcpn_hip.so')
OUT_CLOCK = os.path.join(HERE, 'libcpn_hip_clock.so')  # measurement variant: shader-clock probe in the bf16 conv kernels
ARCH = 'gfx950'

SOURCES = {
    # file: extra flags
    'conv_igemm.hip': [],
    'conv_fp8.hip': [],  # the same source with CPN_FP8 = 1 (e4m3 operands)
    'conv_pair.hip': [],
    'misc_kernels.hip': [],
    'misc_fp8.hip': [],
    'conv_f32.hip': [],
    # decode/NMS must reproduce the reference's fp32 operation order bit-for-bit: no FMA contraction
    'decode_nms.hip': ['-ffp-contract=off'],
    'nms_binned.hip': ['-ffp-contract=off'],
    'labels.hip': [],
    'sparse_heads.hip': [],
    'stem.hip': [],
    'cpn_abi.hip': [],
}
HEADERS = ['cpn_kernels.h', 'cpn_error.h', 'lds_dma.h', 'conv_igemm.hip', os.path.join('..', '..', 'include', 'cpn_hip.h')]


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isfile(c) or c == 'hipcc'):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=True):
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, procs = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [_hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-c', s, '-o', o] + extra + \
                  os.environ.get('CPN_HIPCC_FLAGS', '').split()
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    if force or procs or _stale(OUT, objs):
        cmd = [_hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC'] + objs + ['-o', OUT]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    # libcpn_hip_clock.so: the same library with the shader-clock probe compiled into the bf16 conv kernels
    # (include/cpn_hip.h cpn_debug_clock_probe; bench.py runs it in a child process next to the roofline numbers)
    s = os.path.join(CSRC, 'conv_igemm.hip')
    o = os.path.join(objdir, 'conv_igemm_clock.o')
    if force or _stale(o, [s] + hdrs):
        cmd = [_hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-c', s, '-o', o, '-DCPN_EXP_CLOCK=2'] + \
              os.environ.get('CPN_HIPCC_FLAGS', '').split()
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    cobjs = [o if x.endswith(os.sep + 'conv_igemm.o') else x for x in objs]
    if force or _stale(OUT_CLOCK, cobjs):
        cmd = [_hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC'] + cobjs + ['-o', OUT_CLOCK]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
