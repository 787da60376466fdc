"""bench.py launch contract (CPU, gloo): `--gpus N` outside torchrun starts N ranks itself, a launcher/flag mismatch
fails loudly, and the sharding + packed all-gather plumbing of the multi-GPU path runs end to end (``--dry-run``: no
GPU, no kernels, not a measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=240):
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True,
                          env=e, timeout=timeout, cwd=ROOT)


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


import pytest


@pytest.mark.parametrize('n', [2, 4])
def test_self_launch_n_ranks_gloo(n):
    r = _run(['--gpus', str(n), '--backend', 'gloo', '--dry-run'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d['dry_run'] is True and d['n_gpus'] == n and d['world_size_seen'] == n
    assert d['gathered_detections'] == 3 * d['tiles'] + n * (n - 1) // 2  # 3 per tile + rank: ragged per-rank counts
    assert r.stdout.strip().count('\n') == 0  # ONE line on stdout, whatever the libraries print


def test_single_rank_dry_run():
    d = _json_line(_run(['--gpus', '1', '--dry-run']).stdout)
    assert d['n_gpus'] == 1


def test_world_size_mismatch_fails():
    r = _run(['--gpus', '2', '--dry-run'], env=dict(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'))
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


@pytest.mark.parametrize('slide,tiles', [(900, [3, 2, 2, 2]), (600, [1, 1, 1, 1]), (512, [1, 0, 0, 0])])
def test_four_rank_slide_dry_run_ragged_counts(slide, tiles):
    """the slide workload's loop (tiling, sharding, batching, packed variable-length all-gather,
    redundant global NMS) on 4 gloo ranks with ragged per-rank counts -- tiles without detections, ranks without
    detections, ranks without a single tile -- must give the identical result on every rank."""
    r = _run(['--gpus', '4', '--backend', 'gloo', '--dry-run', '--workload', 'slide', '--slide', str(slide), '--batch', '2'])
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d['identical_on_all_ranks'] and d['n_gpus'] == 4 and d['tiles_per_rank'] == tiles
    assert d['gathered_detections'] == sum(d['detections_per_rank']) and d['final_detections'] <= d['gathered_detections']
    if slide == 900:
        assert 0 in d['detections_per_rank'] or min(d['detections_per_rank']) < max(d['detections_per_rank'])


def test_eight_rank_slide_dry_run_at_configs3_geometry():
    """BASELINE configs[3]'s own geometry -- 16384^2, tiles 512 / 384 = 1849 tiles, batch 16 -- on 8 gloo ranks:
    231 / 232 tiles per rank (ragged last batches of 7 / 8 tiles), one packed all-gather, the global NMS on every rank, identical
    results everywhere.  (CPU stand-ins for the kernels; the slide is a lazily mapped zero tensor.)"""
    r = _run(['--gpus', '8', '--backend', 'gloo', '--dry-run', '--workload', 'slide', '--slide', '16384', '--tile', '512',
              '--stride', '384', '--batch', '16'], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d['identical_on_all_ranks'] and d['n_gpus'] == 8 and d['world_size_seen'] == 8
    assert sum(d['tiles_per_rank']) == 1849 and d['tiles_per_rank'] == [232] + [231] * 7
    assert d['gathered_detections'] == sum(d['detections_per_rank']) > 1849
    assert 0 < d['final_detections'] <= d['gathered_detections']


def test_host_threads_follow_the_cpu_quota(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, '_cpu_quota', lambda: 16.)
    monkeypatch.setattr(bench.os, 'cpu_count', lambda: 256)
    monkeypatch.setattr(bench.os, 'sched_getaffinity', lambda pid: set(range(256)), raising=False)
    assert [bench.host_threads(w) for w in (1, 2, 4, 8, 32)] == [16, 8, 4, 2, 1]
    monkeypatch.setattr(bench, '_cpu_quota', lambda: None)
    assert bench.host_threads(8) == 32
