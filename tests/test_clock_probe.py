"""The shader-clock probe of the bf16 conv kernels (include/cpn_hip.h cpn_debug_clock_probe, tools/clock_probe.py): the measurement
library runs the same graph, reports a plausible clock and a matrix-pipe duty in (0, 1], and leaves the outputs of the convs alone."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_clock_probe_reports_clock_and_duty_of_the_conv_kernels():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ)
    env.pop('CPN_HIP_LIB', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'clock_probe.py'), '3', 'CpnResNet18FPN', '2', '256'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['graph_executions'] == 3 and out['nominal_mhz'] == 2400.
    seen = [k for k in ('conv7x7', 'conv3x3', 'other_taps') if k in out]
    assert 'conv7x7' in seen and 'conv3x3' in seen, out  # the 7x7 heads and the 3x3 convs of the ResNet / FPN
    for k in seen:
        assert 500. < out[k]['mhz'] <= 2600., out[k]        # a shader clock, not the 100-MHz reference
        assert 0. < out[k]['matrix_pipe_duty'] <= 1., out[k]
        assert out[k]['launches'] % 3 == 0 and out[k]['launches'] > 0, out[k]
    assert 500. < out['all_convs_mhz'] <= 2600.


def test_probe_library_computes_the_same_maps():
    """Same plan through both libraries (child process for the probe library): bit-identical head maps."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "import celldetection_amd as cda\n"
        "from celldetection_amd.synth import synth_state_dict\n"
        "m = cda.models.CpnResNet18FPN(3); m.load_state_dict(synth_state_dict(m.state_dict(), seed=0)); m = m.to('cuda:0')\n"
        "x = torch.rand(1, 3, 128, 160, generator=torch.Generator().manual_seed(2)).to('cuda:0')\n"
        "maps = m.core_forward(x)\n"
        "torch.save([t.cpu() for t in maps if t is not None], sys.argv[1])\n" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for tag, lib in (('product', None), ('clock', os.path.join(ROOT, 'celldetection_amd', 'libcpn_hip_clock.so'))):
            env = dict(os.environ)
            env.pop('CPN_HIP_LIB', None)
            if lib:
                env['CPN_HIP_LIB'] = lib
            path = os.path.join(d, tag + '.pt')
            r = subprocess.run([sys.executable, '-c', code, path], capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(torch.load(path))
        assert len(outs[0]) == len(outs[1]) >= 4
        for a, b in zip(*outs):
            assert torch.equal(a, b)
