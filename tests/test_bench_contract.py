"""The JSON line contract of bench.py, checked on the lines this round committed under profiles/ (produced on the MI355X by
tools/gpu_round_pass.sh): every field the driver reads is present and self-consistent, the roofline numbers recompute from
their own parts, and the score-gated line prices EXECUTED FLOPs."""
import glob
import json
import os

import pytest

P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
LINES = sorted(glob.glob(os.path.join(P, 'r03_bench_*.json')) + glob.glob(os.path.join(P, 'r04_bench_*.json')) +
               glob.glob(os.path.join(P, 'r05_bench_*.json')) + glob.glob(os.path.join(P, 'r06_bench_*.json')))


def _load(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_round_three_lines_are_committed():
    names = {os.path.basename(p) for p in LINES}
    for want in ('r03_bench_n1.json', 'r03_bench_sparse_heads_n1.json', 'r03_bench_slide_n1.json',
                 'r03_bench_configs1_resnet18fpn.json', 'r03_bench_configs4_resnet50fpn_bf16.json',
                 'r03_bench_configs4_resnet50fpn_fp8.json', 'r03_bench_fp8_n1.json'):
        assert want in names, want


@pytest.mark.parametrize('path', LINES, ids=[os.path.basename(p) for p in LINES])
def test_line_contract(path):
    d = _load(path)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['unit'] == 'tiles/s' and d['higher_is_better'] is True and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert d['metric'].startswith('tiles/sec (3x') and d['dtype'] in ('bf16', 'fp8') and d['n_gpus'] == 1
    assert d['scaling'] in ('weak', 'strong') and isinstance(d['config'].get('workload'), str) and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == (5000. if d['dtype'] == 'fp8' else 2500.)
    # `frac` prices the REFERENCE graph's FLOPs (the contract's algorithmic work) over the measured time: where a plan executes
    # fewer (sub-pixel / bilinear phase decompositions) it may pass what the hardware can do -- `executed_frac` may not
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.2 < r['frac'] < 0.9
    if 'executed_frac' in r:
        assert 0.2 < r['executed_frac'] < 0.75 and r['executed_frac'] <= r['frac'] * 1.01  # (padded channels: up to 1e-4 above)
    assert 'traffic' in r
    if d['scaling'] == 'weak':  # tile workload: value = tiles of all steps / time; the conv graph is bracketed by HIP events
        tiles = d['config']['tiles_per_gpu_per_step']
        assert abs(d['value'] - tiles / (d['ms_per_step'] / 1e3)) / d['value'] < 1e-6
        assert r['launch_ms'] <= d['ms_per_step'] * 1.01
        assert abs(r['executed_frac'] - r['executed_gflop_per_launch'] / r['launch_ms'] / r['peak']) < 1e-9
        assert r['executed_gflop_per_launch'] > 0
        bb, dom = r['backbone_stack'], r['dominant_kernel']
        assert bb is not None and dom is not None and 0 < bb['frac'] < 0.75 and 0.3 < dom['frac'] < 0.75
        assert abs(bb['frac'] - bb['algorithmic_gflop'] / bb['ms'] / r['peak']) < 1e-9


def test_default_line_is_the_dense_reference_graph_with_cpu_baseline():
    d = _load(os.path.join(P, 'r03_bench_n1.json'))
    assert 'configs[2]' in d['config']['workload'] and d['config']['heads'] == 'dense (reference graph)'
    assert d['config']['world_size_seen_by_rccl'] is None  # one rank: no process group, RCCL never initialised
    r = d['roofline']
    assert abs(r['achieved'] - r['algorithmic_gflop_per_launch'] / r['launch_ms']) < 1e-6  # ALGORITHMIC FLOPs of the reference graph
    assert abs(r['algorithmic_gflop_per_launch'] - 16 * 2392.83) < 1. and r['traffic'] > 2e10
    assert r['executed_gflop_per_launch'] < r['algorithmic_gflop_per_launch']  # the sub-pixel triples execute fewer MACs
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'tiles/s' and c['cores'] >= 1 and 0 < c['value'] < 10 and c['sample']


def test_score_gated_line_prices_executed_flops():
    d = _load(os.path.join(P, 'r03_bench_sparse_heads_n1.json'))
    r, c = d['roofline'], d['config']
    assert 'score-gated' in c['heads'] and 0 < c['proposal_density'] < 0.2 and c['proposals_per_step'] > 1000
    ex = r['executed_gflop_per_step']
    assert abs(ex - (r['executed_gflop_per_launch'] + r['sparse_kernel_gflop_per_step'])) < 1e-6
    assert abs(r['frac'] - ex / d['ms_per_step'] / r['peak']) < 1e-9
    assert r['frac'] < r['algorithmic_frac']  # never the reference graph's FLOPs over the gated time
    dense = _load(os.path.join(P, 'r03_bench_n1.json'))
    assert d['value'] > 1.25 * dense['value'] and ex < 0.7 * dense['roofline']['executed_gflop_per_launch']


def test_round_four_lines_are_committed():
    names = {os.path.basename(p) for p in LINES}
    for want in ('r04_bench_n1.json', 'r04_bench_slide_n1.json', 'r04_bench_slide_4096_rccl_1rank.json',
                 'r04_bench_configs1_resnet18fpn.json', 'r04_bench_configs4_resnet50fpn_bf16.json',
                 'r04_bench_configs4_resnet50fpn_fp8.json', 'r04_bench_fp8_n1.json'):
        assert want in names, want


def test_round_four_headline_carries_the_gated_default_and_the_synchronous_rate():
    """the dense reference graph stays the headline `value`; the product default (score-gated heads)
    is timed in the SAME run at two stated proposal densities with the roofline on executed FLOPs, and so is the rate a
    forward()-per-batch caller gets."""
    d = _load(os.path.join(P, 'r04_bench_n1.json'))
    assert 'configs[2]' in d['config']['workload'] and d['config']['heads'] == 'dense (reference graph)' and d['steps'] >= 50
    assert 'cpu_baseline' in d and d['cpu_baseline']['kind'] == 'port'
    dense_exec = d['roofline']['executed_gflop_per_launch']
    g = d['gated']
    assert "'auto'" in g['mode'] and len(g['lines']) == 2
    for line, target in zip(g['lines'], (.01, .10)):
        assert line['target_density'] == target and abs(line['density'] - target) < 1e-3
        tiles = d['config']['tiles_per_gpu_per_step']
        assert abs(line['value'] - tiles / (line['ms_per_step'] / 1e3)) / line['value'] < 1e-6
        ex = line['executed_gflop_per_step']
        assert abs(line['roofline_frac_executed'] - ex / line['ms_per_step'] / 2500.) < 1e-9
        assert line['sparse_kernel_gflop_per_step'] < ex < 0.7 * dense_exec        # two of the three 7x7 heads are gone
        assert line['roofline_frac_executed'] < line['algorithmic_frac'] < 0.9    # never the dense FLOPs over the gated time
        assert line['value'] > 1.25 * d['value'] and line['conv_graph_ms'] <= line['ms_per_step'] * 1.01
    assert g['lines'][0]['value'] > g['lines'][1]['value']  # more proposals, more gathered work
    sf = d['sync_forward']
    assert 0.9 * d['value'] < sf['value'] <= d['value'] * 1.02 and sf['conv_graph_ms'] < sf['ms_per_step']


def test_round_four_slide_lines():
    for name, floor in (('r04_bench_slide_n1.json', 480.), ('r04_bench_slide_4096_rccl_1rank.json', 480.)):
        d = _load(os.path.join(P, name))
        assert d['scaling'] == 'strong' and d['value'] > floor, (name, d['value'])
        g = d['gated']
        assert g['identical_to_dense'] is True and g['value'] > 1.2 * d['value']
    assert _load(os.path.join(P, 'r04_bench_slide_4096_rccl_1rank.json'))['config']['world_size_seen_by_rccl'] == 1


def test_round_five_lines_are_committed():
    names = {os.path.basename(p) for p in LINES}
    for want in ('r05_bench_n1.json', 'r05_bench_slide_n1.json', 'r05_bench_slide_4096_rccl_1rank.json',
                 'r05_bench_configs1_resnet18fpn.json', 'r05_bench_configs4_resnet50fpn_bf16.json',
                 'r05_bench_configs4_resnet50fpn_fp8.json', 'r05_bench_fp8_n1.json'):
        assert want in names, want


def test_round_five_headline_carries_the_other_baseline_configs_and_the_setup_phases():
    """BASELINE configs[1] (bf16) and configs[4] (fp8, the per-GPU share) are timed by the default run, under the
    same clock as the headline and outside its timed region; `setup_s` says where the wall time of the run went; the dominant-kernel
    statistics separate the flagship tile (three fused 7x7 heads) from MODE_S1F (the decoder convs on two workgroups per CU)."""
    d = _load(os.path.join(P, 'r05_bench_n1.json'))
    assert 'configs[2]' in d['config']['workload'] and d['config']['heads'] == 'dense (reference graph)'
    assert {'gated', 'sync_forward', 'configs', 'setup_s', 'cpu_baseline'} <= set(d)
    lines = d['configs']['lines']
    assert [(l['model'], l['batch'], l['tile'], l['dtype']) for l in lines] == [('CpnResNet18FPN', 8, 512, 'bf16'),
                                                                                ('CpnResNet50FPN', 8, 1024, 'fp8')]
    for l in lines:
        assert abs(l['value'] - l['batch'] / (l['ms_per_step'] / 1e3)) / l['value'] < 1e-6 and l['conv_graph_ms'] <= l['ms_per_step'] * 1.01
        assert l['peak'] == (5000. if l['dtype'] == 'fp8' else 2500.)
        assert abs(l['frac'] - l['algorithmic_gflop_per_launch'] / l['conv_graph_ms'] / l['peak']) < 1e-9
        assert abs(l['executed_frac'] - l['executed_gflop_per_launch'] / l['conv_graph_ms'] / l['peak']) < 1e-9
        assert 0.3 < l['executed_frac'] < l['frac'] < 0.9 and l['traffic'] and l['detections_last_step'] > 50
    assert lines[0]['value'] > 700 and lines[1]['value'] > 330
    s = d['setup_s']
    assert {'import_torch', 'build_model', 'pack_and_warm', 'timed_region', 'per_op_profile', 'extras', 'configs', 'cpu_baseline'} <= set(s)
    assert abs(s['timed_region'] - (d['steps'] + d['warmup'] + 2) * d['ms_per_step'] / 1e3) < 1.5   # the timed steps + their warm-up
    dom = d['roofline']['dominant_kernel']
    assert dom['kernel'] == 'conv_igemm_kernel<8,256,4,2,1>' and dom['launches_per_graph'] == 3 and 0.4 < dom['share_of_graph_time'] < 0.6
    sec = dom['second_kernel']
    assert 'MODE_S1F' in sec['kernel'] and sec['launches_per_graph'] == 11 and 0.45 < sec['frac'] < 0.7
    assert d['value'] > 560 and d['roofline']['frac'] > 0.54 and d['roofline']['backbone_stack']['frac'] > 0.5


def test_round_five_slide_lines():
    for name, floor in (('r05_bench_slide_n1.json', 500.), ('r05_bench_slide_4096_rccl_1rank.json', 500.)):
        d = _load(os.path.join(P, name))
        assert d['scaling'] == 'strong' and d['value'] > floor, (name, d['value'])
        assert d['gated']['identical_to_dense'] is True and d['gated']['value'] > 1.2 * d['value']


@pytest.mark.parametrize('name', ['r06_bench_n1.json', 'r06_bench_n1_box_a.json'])
def test_round_six_line_carries_numeric_parity_and_configs3(name):
    """the default line's `config.parity` is NUMBERS measured in-run (bf16 product path and fp32
    verification path vs the oracle on one tile), `configs.lines` holds three entries incl. configs[3] on one GPU with the
    gather / global-NMS milliseconds and the gated pass identical to the dense one, and the line states the host threads."""
    d = _load(os.path.join(P, name))
    par = d['config']['parity']
    for k in ('proposals_hip', 'proposals_ref', 'iou50_match_rate', 'nms_set_f1', 'max_contour_dev_matched_px',
              'median_contour_dev_matched_px', 'matched_detections'):
        assert isinstance(par[k], (int, float)), k
    assert abs(par['proposals_hip'] - par['proposals_ref']) <= 0.02 * par['proposals_ref'] and par['iou50_match_rate'] > .97
    assert par['nms_set_f1'] > .75 and par['median_contour_dev_matched_px'] < .25
    ns = par['fp32_path_vs_oracle']
    assert ns['index_sets_identical'] is True and ns['classes_identical'] is True and ns['score_max_abs_diff'] < 1e-5
    assert ns['contour_frac_off_by_more_than_1e-4'] < 5e-3  # pixel-snap flips of local_refinement
    lines = d['configs']['lines']
    assert [l['config'][:10] for l in lines] == ['configs[1]', 'configs[4]', 'configs[3]']
    c3 = lines[2]
    assert c3['tiles_total'] == 1849 and c3['slide'] == [3, 16384, 16384] and c3['batch'] == 16 and c3['stride'] == 384
    assert abs(c3['value'] - 1849 / (c3['ms_per_step'] / 1e3)) / c3['value'] < 1e-6
    assert c3['tile_loop_ms'] + c3['gather_ms'] + c3['global_nms_ms'] <= c3['ms_per_step'] * 1.001
    assert c3['detections_final'] < c3['detections_gathered'] and c3['gated']['identical_to_dense'] is True
    assert c3['gated']['value'] > c3['value']
    assert d['host_threads_per_rank'] >= 1 and 'parity' in d['setup_s'] and 'configs' in d['setup_s']


def test_final_line_carries_the_shader_clock_measured_inside_the_conv_kernels():
    """`roofline.shader_clock` (tools/clock_probe.py on libcpn_hip_clock.so, child process of bench.py): MHz and matrix-pipe duty of
    the 7x7 heads and the 3x3 convs while the flagship graph runs; duty x clock / nominal must bracket what the HIP events say about
    the dominant kernel (main loop only vs whole launch: prologue, epilogue and fused ReadOut tails are ~5 % of a workgroup)."""
    d = _load(os.path.join(P, 'r06_bench_n1.json'))
    sc = d['roofline']['shader_clock']
    assert sc['nominal_mhz'] == 2400. and sc['graph_executions'] == 20
    for k in ('conv7x7', 'conv3x3'):
        assert 1200. < sc[k]['mhz'] < 2400. and 0.5 < sc[k]['matrix_pipe_duty'] < 1., (k, sc[k])  # below nominal: the power limit
    loop_frac = sc['conv7x7']['matrix_pipe_duty'] * sc['conv7x7']['frac_of_nominal']
    launch_frac = d['roofline']['dominant_kernel']['frac']
    assert launch_frac < loop_frac < 1.12 * launch_frac, (loop_frac, launch_frac)
    assert sc['conv7x7']['launches'] == 20 * 4  # four 49-tap launches per graph execution
