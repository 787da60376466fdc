"""RCCL on hardware (one GPU is all a test box has): the 'nccl' process group bench.py / tiled_inference use is created
under torch.distributed.run with one rank, and the collectives of the slide path (count all-gather, padded payload
all-gather, MAX all-reduce, barrier) run on it.  The multi-rank logic itself is covered by the gloo tests."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_one_rank_collectives_and_slide_loop():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'rccl_probe.py')]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and 'RCCL_PROBE_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
