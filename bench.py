"""Benchmark of the CPN inference hot path on MI355X: tiles/sec for 3x512x512 tiles with CpnResNeXt101UNet.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One *step* = one pass of the whole hot path (input conversion -> ResNeXt101-UNet conv stack -> 4 heads ->
compaction -> Fourier decode + refinement + boxes -> per-image NMS) over one batch of 16 synthetic 3x512x512 tiles
per GPU, inputs resident in HBM (BASELINE.json configs[2]: the configuration the metric is quoted on).
Weights: seeded synthetic tensors of the exact ginoro/CpnResNeXt101UNet shapes (no network => no checkpoint), heads
calibrated so that decode/NMS process a realistic number of detections.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak of MI355X (MI355X_MICROARCH.md; 2:1-sparsity figures excluded)
GFLOP_PER_TILE = {'CpnResNeXt101UNet': 2392.83, 'CpnResNet18FPN': 2124.85}  # SURVEY.md section 8a, 3x512x512 tiles
# HBM bytes of ONE conv-graph execution (batch 16 x 3x512x512, CpnResNeXt101UNet) from the PMC passes committed in
# profiles/r01_rocprofv3_summary.txt (tools/run_graph_only.py 5; sums over the conv/input/maxpool kernels / 5):
# 2 x FETCH_SIZE (gfx950 correction for wide 16-B/lane streaming reads, MI355X_MICROARCH.md section HBM) + WRITE_SIZE
# = 2 x 11.80 GB + 9.13 GB.  Algorithmic: 12.5 GB read + 9.1 GB written (every op reads/writes its tensors once).
TRAFFIC_BYTES_PER_GRAPH_B16 = 32.73e9


def build_model(name, dev, seed=0, tile=512, calib_tiles=2):
    import celldetection_amd as cda
    from celldetection_amd.synth import calibrate_heads, synth_state_dict
    model = getattr(cda.models, name)(3)
    sd = synth_state_dict(model.state_dict(), seed=seed)
    model.load_state_dict(sd)
    model = model.to(dev)
    x = torch.rand(calib_tiles, 3, tile, tile, generator=torch.Generator().manual_seed(1)).to(dev)

    def core_fn(sd_):
        model.load_state_dict(sd_)
        model.to(dev)
        s, l, r, f = model.core_forward(x)
        s = s.clamp(1e-6, 1 - 1e-6)
        return torch.log(s / (1 - s)), l, r, f

    # ~1.3 % of the 256x256 head grid above threshold -> O(1e3) proposals per tile, contours of a few px radius
    for _ in range(2):  # second pass: the first one only sees clamped logits when the raw heads saturate the sigmoid
        sd, _ = calibrate_heads(sd, core_fn, score_shift=-4.5, score_gain=3., fourier_std=1.5, location_std=.5)
    model.load_state_dict(sd)
    return model.to(dev), sd


def cpu_baseline(sd, tile, seconds_budget=20.):
    """Oracle (torch-CPU fp32 restatement, oracle/cpn_oracle.py) timed on this host's cores on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import cpn_oracle as orc
    # one tile does not scale past a few dozen oneDNN threads (256 threads measured 100x slower than 32 on the
    # 256-core GPU host): use 32 threads and report that number as `cores`
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    x = torch.rand(1, 3, tile, tile, generator=torch.Generator().manual_seed(2))
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    t0 = time.perf_counter()
    orc.cpn_forward(sd_cpu, x)  # warm-up (also bounds the sample)
    warm = time.perf_counter() - t0
    reps = max(1, min(3, int(seconds_budget / max(warm, 1e-3)) - 1))
    best = float('inf')
    for _ in range(reps):
        t0 = time.perf_counter()
        orc.cpn_forward(sd_cpu, x)
        best = min(best, time.perf_counter() - t0)
    return dict(value=1. / best, unit='tiles/s', cores=cores, kind='port',
                sample=f'{reps + 1} x 1 tile 3x{tile}x{tile} fp32 full path (conv graph + decode + NMS), best of {reps}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--tile', type=int, default=512)
    ap.add_argument('--model', default='CpnResNeXt101UNet')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp8'],
                    help="conv-graph precision; 'fp8' (e4m3, K=64 scaled MFMA) is BASELINE.json configs[4] groundwork, "
                         'the headline metric is quoted on bf16')
    ap.add_argument('--pipeline', action='store_true',
                    help='two-stream throughput mode (CPN.forward_pipelined); default: synchronous forward() per step')
    ap.add_argument('--profile-layers', action='store_true', help='print per-op timings to stderr')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = world > 1
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    if dist:
        import torch.distributed as td
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        td.init_process_group('nccl', device_id=dev)

    model, sd = build_model(args.model, dev, tile=args.tile)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(args.batch, 3, args.tile, args.tile, generator=g).to(dev)  # resident in HBM before timing
    if args.precision == 'fp8':
        model.precision = 'fp8'
        model.calibrate_fp8(x[:2])  # static activation scales from a bf16 run on two tiles

    model.engine(dev)  # pack the weights / create the native plan now (set-up, not a step), also when --warmup 0
    state = {}

    def run_step(events):
        # ONE step = conv graph (bracketed by HIP events on its launch stream) + post-processing; warm-up and timed
        # steps run this same function so that the caching allocator is in steady state (no hipMalloc while timing)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        state['maps'] = model.core_forward(x)
        e1.record()
        events.append((e0, e1))
        state['y'] = model.postprocess(*state['maps'], (args.tile, args.tile), flag=model._last_flag)
        return state['y']

    for _ in range(args.warmup):
        y = run_step([])
    if args.pipeline:
        for y in model.forward_pipelined((x for _ in range(max(args.warmup, 2)))):
            pass
    # ---- timed region
    torch.cuda.synchronize()
    if dist:
        td.barrier()
        torch.cuda.synchronize()
    mem0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
    ev = []  # HIP events around every conv-graph execution (the dominant kernel family), on its launch stream
    t0 = time.perf_counter()
    if not args.pipeline:
        for i in range(args.steps):
            y = run_step(ev)
    else:
        # throughput mode of the tile loop: conv graph of step i+1 enqueued before the post-processing of step i
        # (two HIP streams); every step's full work -- conv graph, decode, NMS, result tensors -- completes inside
        # the timed region (final synchronize below)
        for y in model.forward_pipelined((x for _ in range(args.steps)), _events=ev):
            pass
    torch.cuda.synchronize()
    if dist:
        td.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    conv_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    if args.profile_layers and rank == 0:
        print('device allocations (hipMalloc) during the timed steps:',
              torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - mem0, file=sys.stderr)
        print('conv graph per step (ms): ' + ' '.join(f'{a.elapsed_time(b):.2f}' for a, b in ev), file=sys.stderr)

    if args.profile_layers and rank == 0:
        eng = model.engine(dev)
        prof = eng.profile(x, model.core.order, True)
        tot = sum(p['ms'] for p in prof)
        for p in prof:
            if p['op'] == 'conv':
                tf = p['gflop'] / max(p['ms'], 1e-6)
                print(f"{p['index']:3d} conv k{p['k']} s{p['stride']} g{p['groups']:<2d} {p['cin']:5d}->{p['cout']:<5d} "
                      f"{p['ms']:8.3f} ms {tf:8.1f} TF/s  {p['name']}", file=sys.stderr)
            else:
                print(f"{p['index']:3d} {p['op']:8s} {p['ms']:8.3f} ms", file=sys.stderr)
        print(f'conv graph total {tot:.3f} ms', file=sys.stderr)

    if rank == 0:
        tiles = args.batch * args.steps * world
        value = tiles / dt
        gf = GFLOP_PER_TILE.get(args.model)
        if gf is None or args.tile != 512:
            from celldetection_amd.graph import reference_flops
            gf = reference_flops(model._plan, args.tile, args.tile) / 1e9
        achieved = gf * args.batch / conv_ms  # GFLOP / ms = TFLOP/s
        ndet = sum(len(s) for s in y['scores'])
        out = {
            'metric': f'tiles/sec (3x{args.tile}x{args.tile}) {args.model}', 'value': value, 'unit': 'tiles/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': f'{args.model} full CPN path, batch {args.batch} x 3x{args.tile}x{args.tile} per GPU'
                                   + (' (BASELINE.json configs[2])' if (args.model, args.batch, args.tile, args.precision)
                                      == ('CpnResNeXt101UNet', 16, 512, 'bf16') else '')
                                   + ', synthetic weights of the reference shapes',
                       'tiles_per_gpu_per_step': args.batch, 'detections_last_step': ndet,
                       'parallelism': f'tile-sharded x{world}, no data-path collective',
                       'step_mode': 'forward() per step' if not args.pipeline else
                       'forward_pipelined(): post-processing of step i overlaps the conv graph of step i+1'},
            'roofline': {'bound': 'mfma', 'achieved': achieved,
                         'peak': PEAK_BF16_TFLOPS * (2 if args.precision == 'fp8' else 1), 'unit': 'TFLOP/s',
                         'frac': achieved / (PEAK_BF16_TFLOPS * (2 if args.precision == 'fp8' else 1)),
                         'traffic': TRAFFIC_BYTES_PER_GRAPH_B16 if (args.model == 'CpnResNeXt101UNet' and args.batch == 16
                                                                    and args.tile == 512) else None,
                         'kernel': 'conv_igemm_kernel: one conv-graph execution = 126 convs in 122 launches of the kernel '
                                   'family (+ input/maxpool helpers), timed with HIP events on the launch stream',
                         'launch_ms': conv_ms, 'algorithmic_gflop_per_launch': gf * args.batch},
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(sd, args.tile)
        print(json.dumps(out))
    if dist:
        td.destroy_process_group()


if __name__ == '__main__':
    main()
