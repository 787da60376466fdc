"""Benchmark of the CPN inference hot path on MI355X: tiles/sec for 3x512x512 tiles with CpnResNeXt101UNet.

    python bench.py --gpus N --steps K --warmup W [--workload tiles|slide]

``--gpus N`` with N > 1 and no torchrun environment re-launches itself as
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` (one rank per GPU, RCCL);
started BY torchrun it checks that WORLD_SIZE equals N and fails otherwise -- a 1-GPU number can never be reported
for an N-GPU request.  Rank 0 prints ONE JSON line.

Workloads
  tiles (default; BASELINE.json configs[2], the configuration the metric is quoted on): one *step* = one pass of the
      whole hot path (input conversion -> ResNeXt101-UNet conv stack -> 4 heads -> compaction -> Fourier decode +
      refinement + boxes -> per-image NMS) over one batch of 16 synthetic 3x512x512 tiles per GPU, inputs resident in
      HBM.  Tiles are independent: no data-path collective, weak scaling.  The K timed steps run the way the tile loop
      of the product runs them (CPN.forward_pipelined: conv graph of step i+1 enqueued on one HIP stream before the
      post-processing of step i runs on a second one); every step's whole work finishes inside the timed region.
      --no-pipeline: one synchronous forward() per step.
  slide (BASELINE.json configs[3]): one *step* = the whole tiled-inference loop over a synthetic 3x16384x16384 uint8
      slide resident on every rank (1849 tiles 512/384, strided tile -> rank sharding, on-device crops, border
      removal, ONE packed all-gather of the per-rank detections over RCCL, global NMS on every rank).  Fixed total
      work: strong scaling.  The line carries the gather and global-NMS milliseconds.

Weights: seeded synthetic tensors of the exact ginoro/CpnResNeXt101UNet shapes (no network => no checkpoint), heads
calibrated so that decode/NMS process a realistic number of detections.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

_T_PROCESS = time.perf_counter()  # (before `import torch`: 1-2 minutes on a cold box while the image pages in)
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak of MI355X (MI355X_MICROARCH.md; 2:1-sparsity figures excluded)
# SURVEY.md section 8a, 2*MAC of the reference graph per 3x512x512 tile; backbone stack = body + unet (the part the
# >= 70 % MFMA target of BASELINE.json is stated on)
GFLOP_PER_TILE = {'CpnResNeXt101UNet': 2392.83, 'CpnResNet18FPN': 2124.85}
BACKBONE_GFLOP_PER_TILE = {'CpnResNeXt101UNet': 1024.04}
_JSON_OUT = sys.stdout
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'traffic.json')  # written by tools/gpu_round_pass.sh from the PMC passes


CALIB_RECIPE = 'shift-4.5_gain3_f1.5_l.5_v1'  # part of the cache key of the calibrated synthetic weights


PHASES = {'import_torch': time.perf_counter() - _T_PROCESS}
_T_PHASE = [time.perf_counter()]


def phase(name):
    """Wall seconds since the previous mark (the driver's clock around a cold-box run is mostly set-up: reported in the line)."""
    now = time.perf_counter()
    PHASES[name] = PHASES.get(name, 0.) + now - _T_PHASE[0]
    _T_PHASE[0] = now


def build_model(name, dev, seed=0, tile=512, calib_tiles=2):
    """Synthetic weights of the reference shapes, heads calibrated on the GPU.  The calibrated state dict is cached on disk
    for the run (CPN_BENCH_CACHE, default /tmp/cpn_bench_cache): the calibration costs ~8 repack + forward rounds, which
    every rank of an N-GPU launch and every further bench.py call on the box would otherwise repeat."""
    import celldetection_amd as cda
    from celldetection_amd.synth import calibrate_heads, synth_state_dict
    model = getattr(cda.models, name)(3)
    model.sparse_heads = False  # calibration reads the dense head maps
    # the calibration only rescales the eight final 1x1 head tensors (synth.HEAD_FINAL_KEYS): its result for the bench
    # configurations is committed (tools/bench_calibration/*.npz, written by a run with CPN_BENCH_SAVE_CALIB=<dir>), so a
    # cold box builds the SAME workload -- same weights, same proposal density -- without ~10 repack + forward rounds
    fix = os.path.join(ROOT, 'tools', 'bench_calibration', f'{name}_t{tile}_s{seed}_c{calib_tiles}_{CALIB_RECIPE}.npz')
    if os.path.isfile(fix) and os.environ.get('CPN_BENCH_RECALIBRATE') != '1':
        import numpy as np
        with np.load(fix) as z:
            ov = {k: torch.as_tensor(z[k]) for k in z.files}
        sd = synth_state_dict(model.state_dict(), seed=seed, overrides=ov)
        model.load_state_dict(sd)
        return model.to(dev), sd
    cache_dir = os.environ.get('CPN_BENCH_CACHE', '/tmp/cpn_bench_cache')
    cache = os.path.join(cache_dir, f'{name}_t{tile}_s{seed}_c{calib_tiles}_{CALIB_RECIPE}.pt') if cache_dir != '0' else None
    if cache and os.path.isfile(cache):
        try:
            sd = torch.load(cache, map_location='cpu', weights_only=True)
            model.load_state_dict(sd)
            return model.to(dev), sd
        except Exception as e:  # a torn / stale file: rebuild
            print(f'bench.py: ignoring the cached weights {cache} ({type(e).__name__}: {e})', file=sys.stderr)
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

    # score logits standardised to mean -4.5, std 3: a Gaussian would put 1.3 % of the head grid above the 0.9 threshold; the
    # synthetic nets' heavier tail gives ~5 % (the measured density travels in the JSON line) -> O(1e3) proposals per tile,
    # contours of a few px radius
    # deep synthetic nets (ResNet50FPN) saturate the sigmoid completely (all logits clamp to +-13.8, so their statistics
    # say nothing): shrink the final score conv until the logits are observable
    for _ in range(12):
        if float((core_fn(sd)[0].abs() > 12.).float().mean()) < .05:
            break
        for p_ in ('weight', 'bias'):
            sd['core.score_head.block.4.' + p_] = sd['core.score_head.block.4.' + p_] * 0.1
    for _ in range(2):  # second pass: the first one only sees clamped logits when the raw heads saturate the sigmoid
        sd, _ = calibrate_heads(sd, core_fn, score_shift=-4.5, score_gain=3., fourier_std=1.5, location_std=.5)
    if float((core_fn(sd)[0] > 2.197).float().mean()) < 1e-3:
        # a heavily skewed score map whose tail points the wrong way (nothing above the threshold): mirror the head
        for p_ in ('weight', 'bias'):
            sd['core.score_head.block.4.' + p_] = -sd['core.score_head.block.4.' + p_]
        sd, _ = calibrate_heads(sd, core_fn, score_shift=-4.5, score_gain=3., fourier_std=1.5, location_std=.5)
    model.load_state_dict(sd)
    if os.environ.get('CPN_BENCH_SAVE_CALIB'):
        import numpy as np
        from celldetection_amd.synth import HEAD_FINAL_KEYS
        os.makedirs(os.environ['CPN_BENCH_SAVE_CALIB'], exist_ok=True)
        np.savez(os.path.join(os.environ['CPN_BENCH_SAVE_CALIB'], os.path.basename(fix)),
                 **{k: sd[k].detach().cpu().numpy() for k in HEAD_FINAL_KEYS})
    if cache:
        try:
            os.makedirs(cache_dir, exist_ok=True)
            tmp = f'{cache}.{os.getpid()}.tmp'
            torch.save({k: v.detach().cpu() for k, v in sd.items()}, tmp)
            os.replace(tmp, cache)  # atomic: concurrent ranks either see the whole file or none
        except OSError as e:
            print(f'bench.py: could not cache the calibrated weights ({e})', file=sys.stderr)
    return model.to(dev), sd


def _physical_cores():
    try:
        import psutil
        return int(psutil.cpu_count(logical=False) or 0)
    except Exception:
        return 0


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


_CPU_WORKER = r"""
import json, os, sys, time
cores, tile, reps, sd_path, root = json.loads(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
try:
    os.sched_setaffinity(0, cores)
except (AttributeError, OSError):
    pass
import torch
torch.set_num_threads(len(cores))
sys.path.insert(0, os.path.join(root, 'oracle'))
import cpn_oracle as orc
sd = torch.load(sd_path, map_location='cpu', weights_only=True)
x = torch.rand(1, 3, tile, tile, generator=torch.Generator().manual_seed(2))
orc.cpn_forward(sd, x)
print('READY', flush=True)
sys.stdin.readline()
t0 = time.perf_counter()
for _ in range(reps):
    orc.cpn_forward(sd, x)
print('DONE', time.perf_counter() - t0, flush=True)
"""


def _cpu_quota():
    """CPUs this container may keep busy (cgroup CPU bandwidth limit), or None when unlimited / unknown.  Round 5 measured on the
    GPU hosts: 4 pinned 32-thread oracle processes ran 10x SLOWER each than one alone (22-41 s instead of 3.6 s for two tiles) --
    the signature of CFS throttling, and the reason one oneDNN process on 256 threads had measured ~100x slower than on 32: the
    container's quota, not the host's 128 physical cores, is what a CPU baseline can use here."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:  # cgroup v2: "<quota> <period>" | "max <period>"
            q, per = f.read().split()
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
            q = float(f.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _cpu_baseline_all_cores(sd_cpu, tile, phys, threads, single_rate, budget_s=25.):
    """BASELINE.md asks for n = physical cores.  One oneDNN process does not scale a batch of <= 2 tiles past a few dozen threads
    on the 2-socket GPU hosts, so the cores are used the way a CPU deployment of the reference would use them: `phys // threads`
    worker processes (the oracle, `threads` threads each, pinned to their own block of cores), one tile at a time each, released
    together; value = tiles of all workers / the slowest worker's time.  -> dict, or None when it cannot run here."""
    import tempfile
    procs_n = max(1, min(phys // threads, 8))
    if procs_n < 2:
        return None
    reps = max(1, min(4, int(6. * single_rate)))  # ~6 s of timed work per worker at the single-process rate
    shm = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
    fd, path = tempfile.mkstemp(suffix='.pt', dir=shm)
    os.close(fd)
    procs = []
    try:
        torch.save(sd_cpu, path)
        try:
            avail = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            avail = list(range(os.cpu_count() or 1))
        avail = avail[:phys] if len(avail) >= phys else avail  # (logical CPUs 0 .. phys - 1: one per physical core on these hosts)
        blocks = [avail[i * threads:(i + 1) * threads] for i in range(procs_n)]
        blocks = [b for b in blocks if len(b) == threads]
        if len(blocks) < 2:
            return None
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='')
        for b in blocks:
            procs.append(subprocess.Popen([sys.executable, '-c', _CPU_WORKER, json.dumps(b), str(tile), str(reps), path, ROOT],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env))
        deadline = time.perf_counter() + budget_s

        def line(p_):
            while time.perf_counter() < deadline:
                ln = p_.stdout.readline()
                if not ln:
                    raise RuntimeError('worker ended early')
                if ln.startswith(('READY', 'DONE')):
                    return ln
            raise TimeoutError('cpu baseline worker timed out')
        for p_ in procs:
            line(p_)
        for p_ in procs:
            p_.stdin.write('\n')
            p_.stdin.flush()
        times = [float(line(p_).split()[1]) for p_ in procs]
        return dict(value=len(procs) * reps / max(times), processes=len(procs), threads_per_process=threads,
                    cores=len(procs) * threads, tiles_per_process=reps, slowest_s=max(times), fastest_s=min(times))
    except (OSError, RuntimeError, TimeoutError, ValueError) as e:
        print(f'bench.py: all-core CPU baseline skipped ({type(e).__name__}: {e})', file=sys.stderr)
        return None
    finally:
        for p_ in procs:  # (exactly the processes started above)
            if p_.poll() is None:
                p_.kill()
            try:
                p_.wait(timeout=5)
            except subprocess.TimeoutExpired:
                pass
        try:
            os.unlink(path)
        except OSError:
            pass


def cpu_baseline(sd, tile, seconds_budget=12.):
    """Oracle (torch-CPU fp32 restatement, oracle/cpn_oracle.py) timed on this host's cores on a bounded sample: ONE process on
    min(physical cores, 32) threads (oneDNN's conv does not scale a batch of <= 2 tiles past a few dozen threads on the 2-socket
    GPU hosts: 256 threads measured ~100x SLOWER than 32), then -- BASELINE.md asks for n = physical cores -- one such process per
    32 physical cores, all released together (`all_cores`); `value` / `cores` are the all-core figures when that run succeeded."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import cpn_oracle as orc
    phys = _physical_cores() or (os.cpu_count() or 1)
    quota = _cpu_quota()
    # threads: min(physical cores, 32, the CPUs the container may keep busy) -- round 6 measured on a GPU host (16-CPU quota, 128
    # physical cores; profiles/r06_cpu_baseline_threads.txt): 8 threads 0.41, 16: 0.60, 32: 0.58, 48: 0.50 tiles/s
    cores = max(1, min(phys, 32, int(quota) if quota and quota >= 1 else 32))
    torch.set_num_threads(cores)
    batch = 2
    x = torch.rand(batch, 3, tile, tile, generator=torch.Generator().manual_seed(2))
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    t0 = time.perf_counter()
    orc.cpn_forward(sd_cpu, x[:1])  # warm-up on one tile (also bounds the sample)
    warm = time.perf_counter() - t0
    if warm * batch * 2 > seconds_budget:
        batch, x = 1, x[:1]
    reps = max(1, min(5, int(seconds_budget / max(warm * batch, 1e-3))))
    best = float('inf')
    for _ in range(reps):
        t0 = time.perf_counter()
        orc.cpn_forward(sd_cpu, x)
        best = min(best, time.perf_counter() - t0)
    single = dict(value=batch / best, cores=cores,
                  sample=f'1 warm-up tile + {reps} x batch of {batch} tile(s) 3x{tile}x{tile}, fp32 full path (conv graph + '
                         f'decode + NMS) on {cores} threads, best of {reps}')
    out = dict(value=single['value'], unit='tiles/s', cores=cores, kind='port', physical_cores=phys, cpu_model=_cpu_model(),
               sample=single['sample'], single_process=single)
    # the oracle's detections for tile 0 of the sample (one more core forward, both post-processings): the checker side of the
    # `parity` object of the line
    s_, l_, r_, f_, u_ = orc.core_forward(sd_cpu, x[:1], with_uncertainty=True)
    size = (tile, tile)
    ORACLE_TILE['x'] = x[:1]
    ORACLE_TILE['proposals'] = orc.cpn_postprocess(s_, l_, r_, f_, input_size=size, uncertainty=u_, nms=False)
    ORACLE_TILE['detections'] = orc.cpn_postprocess(s_, l_, r_, f_, input_size=size, uncertainty=u_, nms=True)
    out['cpu_quota'] = quota  # (CPUs the container's cgroup lets it keep busy; None = no limit found)
    if quota is not None:
        out['sample'] += f'; the container may keep {quota:g} CPUs busy (cgroup cpu.max) of the host\'s {phys} physical cores'

    # all physical cores only where the container may actually use them: a CPU-bandwidth quota below the worker threads makes the
    # workers throttle each other (see _cpu_quota) -- then the one-process figure IS the baseline this box can give
    usable = phys if quota is None else min(phys, int(quota))
    allc = _cpu_baseline_all_cores(sd_cpu, tile, usable, cores, single['value']) if usable >= 2 * cores else None
    if allc is not None and allc['value'] < 1.2 * single['value']:
        out['all_cores_rejected'] = allc  # (no gain over one process: throttled or memory-bound -- reported, not used)
        allc = None
    if allc is not None:
        out.update(value=allc['value'], cores=allc['cores'], all_cores=allc,
                   sample=f"{allc['processes']} oracle processes x {allc['threads_per_process']} threads (one block of physical cores "
                          f"each), released together after one warm-up tile: {allc['tiles_per_process']} tile(s) 3x{tile}x{tile} per "
                          f"process, fp32 full path; value = all tiles / slowest process ({allc['slowest_s']:.2f} s); one process "
                          f"alone on {cores} threads: {single['value']:.3f} tiles/s")
    return out


ORACLE_TILE = {}  # filled by cpu_baseline(): input tile + the oracle's proposals / detections for it


def _iou_matrix(a, b):
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
    return inter / (area(a)[:, None] + area(b)[None] - inter + 1e-9)


def parity_numbers(model, dev):
    """Reference-vs-HIP agreement at the headline configuration, MEASURED in this run on one tile (the oracle side comes from the
    `cpu_baseline` leg): the bf16 product path against the fp32 CPU oracle -- proposal counts, IoU > 0.5 match rate of the
    proposals, F1 of the post-NMS detection sets (one-to-one IoU > 0.5 matches), contour deviation of the matched detections --
    and the fp32 verification path of the same library against the same oracle (the north-star statement: identical index sets,
    contour coordinates within 1e-4)."""
    import numpy as np
    x = ORACLE_TILE['x'].to(dev)
    ref_p, ref_d = ORACLE_TILE['proposals'], ORACLE_TILE['detections']
    t = lambda v: torch.as_tensor(np.asarray(v))
    prev = model.precision, model.sparse_heads
    out = {}
    try:
        model.sparse_heads = 'auto'  # the product default
        model.precision = 'bf16'
        gp, gd = model(x, nms=False), model(x, nms=True)
        iou = _iou_matrix(gp['boxes'][0].cpu(), t(ref_p['boxes'][0]))
        out['proposals_hip'], out['proposals_ref'] = int(iou.shape[0]), int(iou.shape[1])
        out['iou50_match_rate'] = float(((iou.max(1).values > .5).float().mean() + (iou.max(0).values > .5).float().mean()) / 2)
        iou = _iou_matrix(gd['boxes'][0].cpu(), t(ref_d['boxes'][0]))
        best = iou.argmax(1)
        mutual = (iou.argmax(0)[best] == torch.arange(iou.shape[0])) & (iou.max(1).values > .5)
        tp = int(mutual.sum())
        out['detections_hip'], out['detections_ref'] = int(iou.shape[0]), int(iou.shape[1])
        out['nms_set_f1'] = 2. * tp / max(iou.shape[0] + iou.shape[1], 1)
        dev_px = (gd['contours'][0].cpu()[mutual] - t(ref_d['contours'][0])[best[mutual]]).norm(dim=-1)  # [matched, S]
        out['matched_detections'] = tp
        out['max_contour_dev_matched_px'] = float(dev_px.max()) if tp else None
        out['median_contour_dev_matched_px'] = float(dev_px.median()) if tp else None
        out['mean_contour_dev_matched_px'] = float(dev_px.mean()) if tp else None
        model.precision = 'fp32'  # verification path of the same library
        model.sparse_heads = False
        g32 = model(x, nms=True)
        same = all(tuple(g32[k][0].shape) == tuple(np.asarray(ref_d[k][0]).shape) for k in ('scores', 'contours', 'boxes'))
        ns = {'index_sets_identical': bool(same), 'detections': int(len(ref_d['scores'][0]))}
        if same and len(ref_d['scores'][0]):
            ns['classes_identical'] = bool(torch.equal(g32['classes'][0].cpu(), t(ref_d['classes'][0])))
            d = (g32['contours'][0].cpu().double() - t(ref_d['contours'][0]).double()).abs()
            ns['contour_frac_off_by_more_than_1e-4'] = float((d > 1e-4).double().mean())  # pixel-snap flips of local_refinement
            ns['contour_max_abs_diff_no_flip'] = float(d[d < .1].max()) if bool((d < .1).any()) else 0.
            ns['score_max_abs_diff'] = float((g32['scores'][0].cpu() - t(ref_d['scores'][0])).abs().max())
        out['fp32_path_vs_oracle'] = ns
    finally:
        model.precision, model.sparse_heads = prev
        model._engine = None  # (drop the fp32 engine's packed weights)
    out['sample'] = 'tile 0 of the cpu_baseline sample (seeded rand 3x%dx%d), bf16 product path (sparse_heads auto) and fp32 ' \
                    'verification path vs oracle/cpn_oracle.py fp32 on the host' % tuple(x.shape[-2:])
    return out


def _sclk_mhz(dev):
    """Current shader clock of this rank's GPU in MHz (None when no SMI binding is available): N GPUs of one node share
    its power budget, so a bent weak-scaling curve shows up here first."""
    try:
        return float(torch.cuda.clock_rate(dev))
    except Exception:
        pass
    try:
        out = subprocess.run(['rocm-smi', '-d', str(dev.index or 0), '--showclocks', '--json'], capture_output=True,
                             text=True, timeout=20).stdout
        for card in json.loads(out).values():
            for k, v in card.items():
                if 'sclk' in k.lower() and 'mhz' in str(v).lower():
                    return float(''.join(c for c in str(v).split('Mhz')[0].split('(')[-1] if c.isdigit() or c == '.'))
    except Exception:
        pass
    return None


def shader_clock_probe(args):
    """roofline.shader_clock: shader MHz and matrix-pipe duty of the bf16 conv kernels while the conv graph of this workload runs,
    measured inside the kernels by the probe library (tools/clock_probe.py, include/cpn_hip.h cpn_debug_clock_probe) in a child
    process -- this process keeps the product library.  Outside every timed region; the GPU is idle here while the child runs."""
    lib = os.path.join(ROOT, 'celldetection_amd', 'libcpn_hip_clock.so')
    if not os.path.isfile(lib):
        return {'error': 'celldetection_amd/libcpn_hip_clock.so is not built (python -m celldetection_amd.build)'}
    env = dict(os.environ, CPN_HIP_LIB=lib)
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'clock_probe.py'), '20', args.model, str(args.batch), str(args.tile)],
                           capture_output=True, text=True, timeout=300, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode or not line:
            return {'error': f'clock probe failed (rc {r.returncode}): {r.stderr.strip().splitlines()[-1:]}'}
        return json.loads(line[-1])
    except Exception as e:
        return {'error': f'{type(e).__name__}: {e}'}


def host_threads(world):
    """CPU threads one rank may use: the CPUs this container may keep busy (cgroup quota, else the visible CPUs) shared by the
    ranks of the node.  8 ranks x `os.cpu_count() // 8` = 32 threads each on a 16-CPU quota is the CFS throttling measured in
    round 5 (10-100x slow-downs of weight synthesis / packing): the quota, not the host's core count, is what can run."""
    cpus = os.cpu_count() or 1
    try:
        cpus = min(cpus, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = _cpu_quota()
    if quota is not None:
        cpus = min(cpus, max(1, int(quota)))
    return max(1, cpus // max(world, 1))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args):
    """python bench.py --gpus N (N > 1) outside torchrun: start one rank per GPU and relay the JSON line."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def setup_per_rank(dev, world, dist, keys=('import_torch', 'import_and_init', 'build_model', 'pack_and_warm')):
    """Set-up seconds of every rank (a rank throttled by the container's CPU quota shows up here), or None on one rank."""
    if not dist or world < 2:
        return None
    import torch.distributed as td
    t = torch.tensor([PHASES.get(k, 0.) for k in keys], dtype=torch.float64, device=dev)
    allt = [torch.zeros_like(t) for _ in range(world)]
    td.all_gather(allt, t)
    return {k: [round(float(a[i]), 2) for a in allt] for i, k in enumerate(keys)}


def load_traffic(model, batch, tile, precision):
    """HBM bytes per conv-graph execution from the committed PMC passes (profiles/traffic.json), or None."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, None
    key = f'{model}/b{batch}/t{tile}/{precision}'
    e = t.get(key)
    if not e:
        return None, None
    return float(e['traffic_bytes_per_graph']), e.get('source')


def dry_run_slide(args, world, rank):
    """--dry-run --workload slide: the slide loop of the product (inference.tiled_inference: tiling, strided tile -> rank
    sharding, batching, packed variable-length all-gather, redundant global NMS) on CPU ranks with a synthetic per-tile
    forward (a deterministic function of the tile offset; some tiles yield NO detections) and pure-torch stand-ins for the
    border rule / NMS kernels.  Ranks may own no tile at all (slide smaller than world x crop).  Every rank checks that
    its final result equals rank 0's bit for bit.  NOT a measurement."""
    import hashlib
    import torch.distributed as td
    from celldetection_amd import inference
    S_, O_ = 8, 3

    def forward(tiles, offsets, **kw):
        out = {k: [] for k in inference.KEYS}
        for n in range(tiles.shape[0]):
            ox, oy = int(offsets[n, 0]), int(offsets[n, 1])
            g = torch.Generator().manual_seed(ox * 7919 + oy + 1)
            k = ((ox // max(args.stride, 1)) + (oy // max(args.stride, 1))) % 3 * 4  # 0, 4 or 8 detections
            ctr = torch.rand(k, 1, 2, generator=g) * tiles.shape[-1]
            con = ctr + torch.rand(k, S_, 2, generator=g) * 12 - 6 + torch.tensor([ox, oy], dtype=torch.float32)
            out['contours'].append(con)
            out['contour_proposals'].append(con + 1)
            out['boxes'].append(torch.cat((con.min(1).values, con.max(1).values), 1))
            out['scores'].append(torch.rand(k, generator=g))
            out['classes'].append(torch.ones(k, dtype=torch.int64))
            out['locations'].append(ctr[:, 0] + torch.tensor([ox, oy], dtype=torch.float32))
            out['fourier'].append(torch.randn(k, O_, 4, generator=g))
        return out

    def nms(boxes, scores, thr):  # greedy box NMS, descending score (stable); one vector pass per kept box
        order = torch.argsort(scores, descending=True, stable=True)
        b = boxes[order]
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        dead = torch.zeros(b.shape[0], dtype=torch.bool)
        keep = []
        for i in range(b.shape[0]):
            if dead[i]:
                continue
            keep.append(int(order[i]))
            lt, rb = torch.maximum(b[i, :2], b[i + 1:, :2]), torch.minimum(b[i, 2:], b[i + 1:, 2:])
            inter = (rb - lt).clamp(min=0).prod(1)
            dead[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > thr
        return torch.tensor(keep, dtype=torch.int64)

    class Model:
        nms_thresh, samples, order = .3, S_, O_

        class core:
            order = O_

    keep_all = lambda contours, image_index, sides, offsets, size, pad: torch.ones(contours.shape[0], dtype=torch.bool)
    img = torch.zeros(3, args.slide, args.slide, dtype=torch.uint8)
    t = {}
    res = inference.tiled_inference(Model(), img, crop_size=(args.tile, args.tile), strides=(args.stride, args.stride),
                                    batch_size=args.batch, forward_fn=forward,
                                    ops_fns=(keep_all, inference.stitch_rule_batched, nms), timings=t)
    digest = hashlib.sha256(b''.join(res[k].contiguous().numpy().tobytes() for k in inference.KEYS)).digest()[:8]
    mine = torch.tensor([int.from_bytes(digest, 'little') >> 1, t['tiles_local'], t['detections_local']], dtype=torch.int64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        td.all_gather(allr, mine)
        td.barrier()
    else:
        allr = [mine]
    assert all(int(a[0]) == int(allr[0][0]) for a in allr), 'ranks disagree on the final detections'
    if rank == 0:
        print(json.dumps({'dry_run': True, 'workload': 'slide', 'n_gpus': world, 'world_size_seen': world,
                          'backend': args.backend, 'tiles_per_rank': [int(a[1]) for a in allr],
                          'detections_per_rank': [int(a[2]) for a in allr],
                          'gathered_detections': t['detections_gathered'], 'final_detections': t['detections_final'],
                          'identical_on_all_ranks': True}), file=_JSON_OUT, flush=True)


def dry_run(args, world, rank):
    """CPU exercise of the launch / rendezvous / sharding / packed all-gather plumbing (no GPU, no kernels): used by
    tests/test_bench_launch.py with --backend gloo.  Prints a JSON line marked dry_run; NOT a measurement."""
    import torch.distributed as td
    from celldetection_amd import inference, util
    if args.workload == 'slide':
        return dry_run_slide(args, world, rank)
    S, O = 8, 3
    n_tiles = len(list(util.get_tiling_slices((2048, 2048), (512, 512), (384, 384))[0]))
    mine = inference.shard_tiles(n_tiles, rank, world)
    k = 3 * len(mine) + rank  # deterministic fake detections: 3 per tile (+rank: ragged counts)
    g = torch.Generator().manual_seed(rank)
    d = dict(contours=torch.rand(k, S, 2, generator=g), boxes=torch.rand(k, 4, generator=g),
             scores=torch.rand(k, generator=g), classes=torch.ones(k, dtype=torch.int64),
             locations=torch.rand(k, 2, generator=g), fourier=torch.rand(k, O, 4, generator=g),
             contour_proposals=torch.rand(k, S, 2, generator=g))
    buf = inference.gather_detections(inference.pack_detections(d))
    total = buf.shape[0]
    expect = sum(3 * len(inference.shard_tiles(n_tiles, r, world)) + r for r in range(world))
    assert total == expect, (total, expect)
    if world > 1:
        td.barrier()
    if rank == 0:
        print(json.dumps({'dry_run': True, 'n_gpus': world, 'world_size_seen': world, 'backend': args.backend,
                          'tiles': n_tiles, 'gathered_detections': total}), file=_JSON_OUT, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--tile', type=int, default=512)
    ap.add_argument('--model', default='CpnResNeXt101UNet')
    ap.add_argument('--workload', default='tiles', choices=['tiles', 'slide'])
    ap.add_argument('--slide', type=int, default=16384, help='slide edge length of the slide workload')
    ap.add_argument('--stride', type=int, default=384, help='tile stride of the slide workload')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help="'nccl' = RCCL; 'gloo' only with --dry-run")
    ap.add_argument('--dry-run', action='store_true', help='CPU plumbing check of launch + sharding + gather (no GPU)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp8'],
                    help="conv-graph precision; 'fp8' (e4m3, K=64 scaled MFMA) is BASELINE.json configs[4] groundwork, "
                         'the headline metric is quoted on bf16')
    ap.add_argument('--pipeline', dest='pipeline', action='store_true', default=True,
                    help='(default) throughput mode of the tile loop, CPN.forward_pipelined: the conv graph of step i+1 is '
                         'enqueued before the post-processing of step i; all work of the K steps completes inside the timed region')
    ap.add_argument('--no-pipeline', dest='pipeline', action='store_false', help='synchronous forward() per step')
    ap.add_argument('--profile-layers', action='store_true', help='print per-op timings to stderr')
    ap.add_argument('--sparse-heads', action='store_true',
                    help='score-gated location / Fourier heads -- evaluated at the proposal pixels only, identical outputs '
                         '(csrc/sparse_heads.hip); bf16 only.  The default line stays the dense reference graph')
    ap.add_argument('--no-extras', action='store_true', help='skip the sync_forward / gated sub-measurements (profiling runs)')
    ap.add_argument('--no-subpixel', action='store_true',
                    help='A/B switch: run the UNet decoder convs over upsampled maps as the reference states them instead '
                         'of their sub-pixel decomposition (model.subpixel = False)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 50 if args.workload == 'tiles' else 1
    if args.warmup is None:
        args.warmup = 5 if args.workload == 'tiles' else 1

    in_torchrun = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ
    if args.gpus > 1 and not in_torchrun:
        sys.exit(self_launch(args))
    # stdout carries ONE JSON line: everything else a library prints there (RCCL writes its version banner to stdout when
    # the communicator is created) goes to stderr
    global _JSON_OUT
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    # CPN_BENCH_FORCE_DIST=1: create the RCCL process group on a one-rank launch too (single-GPU hardware smoke of the
    # nccl init / barrier / all-reduce calls of the N > 1 path)
    dist = world > 1 or (in_torchrun and os.environ.get('CPN_BENCH_FORCE_DIST') == '1')
    if args.dry_run:
        torch.set_num_threads(host_threads(world))
        if dist:
            import torch.distributed as td
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            td.init_process_group(args.backend)
        dry_run(args, world, rank)
        if dist:
            td.destroy_process_group()
        return
    if args.backend != 'nccl':
        raise SystemExit('bench.py: measurements run on RCCL (--backend nccl); gloo is for --dry-run only')
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    # N ranks on one host share what the container's CPU quota lets it keep busy (weight synthesis / packing are host work)
    torch.set_num_threads(host_threads(world))
    if dist:
        import torch.distributed as td
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        td.init_process_group('nccl', device_id=dev)
        assert td.get_world_size() == args.gpus, (td.get_world_size(), args.gpus)

    phase('import_and_init')
    model, sd = build_model(args.model, dev, tile=args.tile)
    phase('build_model')
    if args.workload == 'slide':
        return slide_workload(args, model, dev, world, rank, dist)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(args.batch, 3, args.tile, args.tile, generator=g).to(dev)  # resident in HBM before timing
    if args.precision == 'fp8':
        model.precision = 'fp8'
        model.calibrate_fp8(x[:2])  # static activation scales from a bf16 run on two tiles

    if args.no_subpixel:
        model.subpixel = False
    # the headline is the reference's DENSE graph; the product default ('auto': score-gated location / Fourier heads in the
    # forward paths, identical outputs) is measured next to it, in the same run, as the `gated` sub-object
    model.sparse_heads = False
    if args.sparse_heads:
        if args.precision != 'bf16':
            raise SystemExit('--sparse-heads: bf16 only')
        model.sparse_heads = True
    state = {}

    def warm_engine():
        # engine set-up, independent of --warmup: pack the weights / create the native plan, and capture the conv graph of
        # this shape into its GRAPH_SLOTS hipGraph instances (cpn._Engine.run captures the 2nd..4th time a shape is seen) --
        # done here so that no capture can fall into a timed region
        e_ = model.engine(dev)
        for _ in range(e_.GRAPH_SLOTS + 1):
            model.core_forward(x, _static_ok=True)
        torch.cuda.synchronize()
        return e_

    def run_step(events):
        # ONE step = conv graph (bracketed by HIP events on its launch stream) + post-processing; warm-up and timed
        # steps run this same function so that the caching allocator is in steady state (no hipMalloc while timing)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        state['maps'] = model.core_forward(x, _static_ok=True)  # (consumed right below, before the slot is re-used)
        e1.record()
        events.append((e0, e1))
        state['y'] = model.postprocess(*state['maps'], (args.tile, args.tile), flag=model._last_flag,
                                       sparse=model._last_sparse)
        return state['y']

    def timed(steps, warmup, pipeline):
        """-> (seconds for `steps` steps (MAX over ranks), HIP-event pairs around every conv-graph execution, last output)"""
        y_ = None
        for _ in range(warmup):
            y_ = run_step([])
        if pipeline:
            for y_ in model.forward_pipelined((x for _ in range(max(warmup, 2)))):
                pass
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()
        ev_ = []  # HIP events around every conv-graph execution (the dominant kernel family), on its launch stream
        t0 = time.perf_counter()
        if not pipeline:
            for _ in range(steps):
                y_ = run_step(ev_)
        else:
            # throughput mode of the tile loop: conv graph of step i+1 enqueued before the post-processing of step i
            # (two HIP streams); every step's full work -- conv graph, decode, NMS, result tensors -- completes inside
            # the timed region (final synchronize below)
            for y_ in model.forward_pipelined((x for _ in range(steps)), _events=ev_):
                pass
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if dist:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, ev_, y_

    eng = warm_engine()
    phase('pack_and_warm')
    mem0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
    # ---- timed region of the headline value
    dt, ev, y = timed(args.steps, args.warmup, args.pipeline)
    mem1 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
    phase('timed_region')
    sclk = _sclk_mhz(dev)  # right after the timed region, per rank
    setup_ranks = setup_per_rank(dev, world, dist)
    if dist:
        t = torch.tensor([sclk if sclk is not None else -1.], dtype=torch.float64, device=dev)
        allc = [torch.zeros_like(t) for _ in range(world)]
        td.all_gather(allc, t)
        sclk_all = [None if float(c.item()) < 0 else float(c.item()) for c in allc]
    else:
        sclk_all = [sclk]
    conv_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    if args.profile_layers and rank == 0:
        print('device allocations (hipMalloc) during the timed steps:', mem1 - mem0, file=sys.stderr)
        print('conv graph per step (ms): ' + ' '.join(f'{a.elapsed_time(b):.2f}' for a, b in ev), file=sys.stderr)

    # per-op timing of ONE more graph execution (outside the timed region): backbone-stack fraction of the roofline
    backbone = None
    dominant = None
    prof = None
    prof_batch = args.batch
    if rank == 0:
        from celldetection_amd.graph import reference_flops
        peng = eng
        if eng.sparse:  # per-op timing needs a plan whose heads all run inside the graph: profile the dense plan's ops
            model.sparse_heads = False
            peng = model.engine(dev)
        prof_batch = peng.max_batch(args.batch, args.tile, args.tile)  # (the engine splits larger batches: 2^31-byte tensors)
        peng.profile(x[:prof_batch], model.core.order, True)  # (warm: a fresh engine's first run pays one-time set-up)
        prof = peng.profile(x[:prof_batch], model.core.order, True)
        if eng.sparse:
            model.sparse_heads = True
            model._engine = eng
    if prof is not None:
        tot = sum(p['ms'] for p in prof)
        bb_ms = sum(p['ms'] for p in prof if p['op'] != 'conv' or 'backbone' in p['name'])
        bb_gf = BACKBONE_GFLOP_PER_TILE.get(args.model) if args.tile == 512 else None
        if bb_gf is None:
            bb_gf = reference_flops(model._plan, args.tile, args.tile, only=lambda op: 'backbone' in op['w']) / 1e9
        if bb_gf:
            backbone = {'ms': bb_ms, 'achieved': bb_gf * prof_batch / bb_ms,
                        'frac': bb_gf * prof_batch / bb_ms / (PEAK_BF16_TFLOPS * (2 if args.precision == 'fp8' else 1)),
                        'algorithmic_gflop': bb_gf * prof_batch, 'batch': prof_batch,
                        'note': 'backbone conv stack (encoder + decoder incl. input/maxpool helpers) of one per-op-timed graph '
                                'execution outside the timed region; algorithmic FLOPs of the reference graph'}
        # the dominant kernel instantiation, conv_igemm_kernel<8,256,4,2,1>: dense k x k stride-1 convs with >= 256 output channels that
        # do NOT take the two-workgroups-per-CU tiles (round 5: MODE_S1F runs the decoder's single-source 3x3 / 2x2 convs; what
        # stays on the flagship tile are the three fused 7x7 heads of CpnResNeXt101UNet and any conv with a second / resized source),
        # from the same per-op timing (a 16-pixel-wide output runs the MODE_N instantiation <8,256,4,2,6>: not part of the statistics)
        pops = peng.plan.ops

        def s1f(p):  # mirrors csrc/conv_igemm.hip flat_ok() for this batch
            o = pops[p['index']]
            return (args.precision == 'bf16' and o.get('dst') is not None and o.get('src1') is None and not o.get('up0')
                    and p['k'] in (2, 3) and p['stride'] == 1 and p['groups'] == 1 and o.get('fuse') is None
                    and ((p['cout'] or 0) + 31) // 32 * 32 % 128 == 0 and os.environ.get('CPN_S1F', '1') != '0')

        convs = [p for p in prof if p['op'] == 'conv' and (p['k'] or 0) > 1 and p['groups'] == 1 and p['stride'] == 1
                 and (p['cin'] or 0) >= 64 and p['gflop'] > 0 and p['out_w'] != 16]
        dom = [p for p in convs if (p['cout'] or 0) >= 256 and not s1f(p)]
        flat = [p for p in convs if s1f(p)]
        pk = PEAK_BF16_TFLOPS * (2 if args.precision == 'fp8' else 1)

        def kstats(ps, name):
            ms_, gf_ = sum(p['ms'] for p in ps), sum(p['gflop'] for p in ps)
            return {'kernel': name, 'launches_per_graph': len(ps), 'avg_launch_us': 1e3 * ms_ / len(ps), 'gflop_per_graph': gf_,
                    'batch': prof_batch, 'achieved': gf_ / ms_, 'frac': gf_ / ms_ / pk, 'share_of_graph_time': ms_ / tot}
        if dom:
            dominant = kstats(dom, 'conv_igemm_kernel<8,256,4,2,1>' if args.precision == 'bf16' else 'cpn_fp8::conv_igemm_kernel<8,256,4,2,1>')
            if flat:
                dominant['second_kernel'] = kstats(flat, 'conv_igemm_kernel<8,128,4,2,7> (MODE_S1F: two workgroups per CU)')
        if args.profile_layers:
            for p in prof:
                if p['op'] == 'conv_bridge':
                    print(f"{p['index']:3d} conv_bridge (scatter conv -> conv 3x3, one kernel) {p['ms']:8.3f} ms "
                          f"{p['gflop'] / max(p['ms'], 1e-6):8.1f} TF/s executed  {p['name']}", file=sys.stderr)
                elif p['op'] == 'conv_pair':
                    print(f"{p['index']:3d} conv_pair (conv1 1x1 -> grouped conv2 3x3) {p['ms']:8.3f} ms "
                          f"{p['gflop'] / max(p['ms'], 1e-6):8.1f} TF/s executed  {p['name']}", file=sys.stderr)
                elif p['op'] == 'conv':
                    tf = p['gflop'] / max(p['ms'], 1e-6)
                    print(f"{p['index']:3d} conv k{p['k']} s{p['stride']} g{p['groups']:<2d} {p['cin']:5d}->{p['cout']:<5d} "
                          f"{p['ms']:8.3f} ms {tf:8.1f} TF/s  {p['name']}", file=sys.stderr)
                else:
                    print(f"{p['index']:3d} {p['op']:8s} {p['ms']:8.3f} ms", file=sys.stderr)
            print(f'conv graph total {tot:.3f} ms', file=sys.stderr)

    phase('per_op_profile')
    if rank == 0:
        tiles = args.batch * args.steps * world
        value = tiles / dt
        gf = GFLOP_PER_TILE.get(args.model)
        if gf is None or args.tile != 512:
            from celldetection_amd.graph import reference_flops
            gf = reference_flops(model._plan, args.tile, args.tile) / 1e9
        peak = PEAK_BF16_TFLOPS * (2 if args.precision == 'fp8' else 1)
        achieved = gf * args.batch / conv_ms  # GFLOP / ms = TFLOP/s
        executed = eng.executed_flops(args.batch, args.tile, args.tile) / 1e9
        traffic, traffic_src = load_traffic(args.model, args.batch, args.tile, args.precision)
        ndet = sum(len(s) for s in y['scores'])
        # conv launches of one graph execution (sub-pixel triples run either their head or their two member ops: per-op
        # executed FLOPs of the profiled run tell which)
        n_launch = sum(1 for p in prof if p['op'] in ('conv', 'conv_pair', 'conv_bridge') and p['gflop'] > 0) - (2 if args.sparse_heads and eng.sparse else 0)
        rccl_world = td.get_world_size() if dist else None  # None: no process group exists (one rank, RCCL never initialised)
        assert rccl_world is None or rccl_world == world == args.gpus, (rccl_world, world, args.gpus)
        gated = None
        if args.sparse_heads and eng.sparse:
            # score-gated line: the roofline fraction counts EXECUTED FLOPs (conv graph without the two deferred heads + the
            # gathered kernel's work on the proposals, padded to its 128-proposal workgroups) over the whole step
            sc = model.core_forward(x)[0]
            n_prop = int((sc > model.score_thresh).sum().item())
            head_px = sc.shape[0] * sc.shape[-2] * sc.shape[-1]
            per_prop = sum(2. * op['cout'] * op['cin'] * op['k'] ** 2 + 2. * op['fuse']['cout'] * op['cout']
                           for op in eng.plan.ops if op.get('deferred'))
            sparse_gf = per_prop * ((n_prop + 127) // 128 * 128) / 1e9
            gated = dict(proposals=n_prop, head_pixels=head_px, density=n_prop / head_px, sparse_kernel_gflop=sparse_gf)
        out = {
            'metric': f'tiles/sec (3x{args.tile}x{args.tile}) {args.model}', 'value': value, 'unit': 'tiles/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': f'{args.model} full CPN path, batch {args.batch} x 3x{args.tile}x{args.tile} per GPU'
                                   + (' (BASELINE.json configs[2])' if (args.model, args.batch, args.tile, args.precision)
                                      == ('CpnResNeXt101UNet', 16, 512, 'bf16') else '')
                                   + ', synthetic weights of the reference shapes',
                       'tiles_per_gpu_per_step': args.batch, 'detections_last_step': ndet,
                       'world_size_seen_by_rccl': rccl_world, 'sclk_mhz_per_rank_after_timing': sclk_all,
                       'parity_statement': 'bf16 conv graph: matched against the fp32 reference (gates = 2 x the measured error, '
                                           'tests/golden/bf16_measured.json); decode / NMS bit-exact on identical head maps; '
                                           'precision=fp32 reproduces the reference within 1e-4; `parity` = numbers measured in this run',
                       'parallelism': f'tile-sharded x{world}, one process per GPU, no data-path collective',
                       'step_mode': 'forward() per step' if not args.pipeline else
                       'forward_pipelined(): post-processing of step i overlaps the conv graph of step i+1',
                       'heads': 'score-gated location / Fourier heads (exact; evaluated at the proposals only, outside the '
                                'HIP-event bracket of the conv graph)' if args.sparse_heads else 'dense (reference graph)'},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                         'traffic': traffic if not (args.sparse_heads or args.no_subpixel) else None, 'traffic_source': traffic_src,
                         'kernel': f'conv_igemm_kernel: one conv-graph execution = {n_launch} launches of the kernel family '
                                   '(+ input/maxpool helpers), timed with HIP events on the launch stream',
                         'launch_ms': conv_ms, 'algorithmic_gflop_per_launch': gf * args.batch,
                         'executed_gflop_per_launch': executed, 'executed_frac': executed / conv_ms / peak,
                         'backbone_stack': backbone, 'dominant_kernel': dominant,
                         # not measured by this run: what the matrix pipes sustain with NO memory instruction in the loop
                         # (tools/probes/mfma_ceiling.hip, 8 accumulators per wave, 2 x 256 threads per CU), as fractions of `peak`
                         'register_only_mfma_loop_frac_of_peak': {
                             'zero_operands': 0.995 if args.precision == 'fp8' else 0.985,
                             'random_operands': 0.794 if args.precision == 'fp8' else 0.701,
                             'source': 'profiles/r05_kernel_experiments.txt #8 (one MI355X box; power / clock bound)'},
                         # ... and with the operand fragments re-read from LDS every K step (the conv kernels' 2 x 4 wave tile, 6 reads per
                         # 8 MFMAs, two fragment sets; no DMA, no barrier, no addressing): the bound of any LDS-fed main loop
                         'lds_fed_mfma_loop_frac_of_peak': {
                             'zero_operands': 0.899 if args.precision == 'fp8' else 0.905,
                             'random_operands': 0.677 if args.precision == 'fp8' else 0.615,
                             'source': 'profiles/r06_kernel_experiments.txt #14 (tools/probes/mfma_lds_ratio.hip)'}},
        }
        if gated is not None:
            step_ms = 1e3 * dt / args.steps
            ex = executed + gated['sparse_kernel_gflop']
            out['config']['proposal_density'] = gated['density']
            out['config']['proposals_per_step'] = gated['proposals']
            out['roofline'].update({
                'achieved': ex / step_ms, 'frac': ex / step_ms / peak,
                'frac_basis': 'EXECUTED FLOPs (conv graph without the two deferred heads + the gathered head kernel on the '
                              'proposals) / whole step time: the gathered kernel runs in the post-processing, outside the '
                              'HIP-event bracket of the conv graph',
                'executed_gflop_per_step': ex, 'sparse_kernel_gflop_per_step': gated['sparse_kernel_gflop'],
                'algorithmic_frac': gf * args.batch / step_ms / peak})
        if world == 1 and args.precision == 'bf16' and not args.sparse_heads and not args.no_extras:
            # ---- same run, same box, outside the headline's timed region
            steps2 = max(5, args.steps // 2)
            if args.pipeline:  # the rate a caller of forward() per batch gets (LitCpn._predict_step, model(x)): no overlap of
                dts, evs, _ = timed(steps2, 2, False)  # the post-processing with the next conv graph
                out['sync_forward'] = {'value': args.batch * steps2 / dts, 'unit': 'tiles/s', 'steps': steps2,
                                       'ms_per_step': 1e3 * dts / steps2,
                                       'conv_graph_ms': sum(a.elapsed_time(b) for a, b in evs) / steps2,
                                       'step_mode': 'forward() per step (dense graph), synchronous'}
            try:
                out['gated'] = gated_lines(model, x, args, timed, warm_engine, gf, peak)
            except Exception as e:
                out['gated'] = {'error': f'{type(e).__name__}: {e}'}
            phase('extras')
            out['roofline']['shader_clock'] = shader_clock_probe(args)
            phase('shader_clock')
            if (args.model, args.batch, args.tile) == ('CpnResNeXt101UNet', 16, 512):
                # (sub-measurements outside the headline's timed region: a failure here must not cost the line)
                try:
                    out['configs'] = configs_lines(dev)
                    out['configs']['lines'].append(slide_line(model, dev, args))
                except Exception as e:
                    out.setdefault('configs', {'lines': []})['error'] = f'{type(e).__name__}: {e}'
                    torch.cuda.empty_cache()
                phase('configs')
        if not args.no_cpu_baseline and world == 1:
            try:
                out['cpu_baseline'] = cpu_baseline(sd, args.tile)
            except Exception as e:  # (the contract asks for the object: say why it is missing rather than lose the line)
                out['cpu_baseline'] = {'value': None, 'unit': 'tiles/s', 'cores': 0, 'kind': 'port', 'sample': f'failed: {type(e).__name__}: {e}'}
            phase('cpu_baseline')
            if args.precision == 'bf16' and ORACLE_TILE:
                try:
                    out['config']['parity'] = parity_numbers(model, dev)
                except Exception as e:  # (the line must still be printed)
                    out['config']['parity'] = {'error': f'{type(e).__name__}: {e}'}
                phase('parity')
        out['setup_s'] = {k: round(v, 2) for k, v in PHASES.items()}
        if setup_ranks is not None:
            out['setup_s_per_rank'] = setup_ranks
        out['host_threads_per_rank'] = host_threads(world)
        print(json.dumps(out), file=_JSON_OUT, flush=True)
    if dist:
        td.destroy_process_group()


def measure_pipelined(model, x, steps, warmup=2):
    """One GPU, no process group: engine warm-up (hipGraph slots captured), `steps` pipelined steps between synchronizes ->
    (seconds, HIP-event pairs around every conv-graph execution, last output, engine)."""
    eng = model.engine(x.device)
    for _ in range(eng.GRAPH_SLOTS + 1):
        model.core_forward(x, _static_ok=True)
    y = None
    for y in model.forward_pipelined((x for _ in range(max(warmup, 2)))):
        pass
    torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    for y in model.forward_pipelined((x for _ in range(steps)), _events=ev):
        pass
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ev, y, eng


def configs_lines(dev):
    """BASELINE.json configs[1] and configs[4] (its per-GPU share: 32 tiles over 4 GPUs, no data-path collective) under the
    SAME clock as the headline: dense reference graph, pipelined tile loop, inputs resident in HBM.
    `frac` prices the REFERENCE graph's FLOPs over the conv-graph time, `executed_frac` the FLOPs the MFMA loops execute (the
    bilinear phase head executes 0.55-0.68 of the reference's): the hardware-utilisation figure is the second one."""
    from celldetection_amd.graph import reference_flops
    out = []
    for label, name, batch, tile, precision, steps in (
            ('configs[1]', 'CpnResNet18FPN', 8, 512, 'bf16', 20),
            ('configs[4] per-GPU share (8 of 32 tiles)', 'CpnResNet50FPN', 8, 1024, 'fp8', 10)):
        t_setup = time.perf_counter()
        model, _ = build_model(name, dev, tile=tile, calib_tiles=1 if tile > 512 else 2)
        model.sparse_heads = False
        x = torch.rand(batch, 3, tile, tile, generator=torch.Generator().manual_seed(100)).to(dev)
        if precision == 'fp8':
            model.precision = 'fp8'
            model.calibrate_fp8(x[:1])
        t_setup = time.perf_counter() - t_setup
        dt, ev, y, eng = measure_pipelined(model, x, steps)
        conv_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
        gf = reference_flops(model._plan, tile, tile) / 1e9
        executed = eng.executed_flops(batch, tile, tile) / 1e9
        peak = PEAK_BF16_TFLOPS * (2 if precision == 'fp8' else 1)
        traffic, traffic_src = load_traffic(name, batch, tile, precision)
        out.append({'config': label, 'model': name, 'batch': batch, 'tile': tile, 'dtype': precision,
                    'value': batch * steps / dt, 'unit': 'tiles/s', 'steps': steps, 'ms_per_step': 1e3 * dt / steps,
                    'conv_graph_ms': conv_ms, 'algorithmic_gflop_per_launch': gf * batch,
                    'executed_gflop_per_launch': executed, 'peak': peak, 'frac': gf * batch / conv_ms / peak,
                    'executed_frac': executed / conv_ms / peak, 'traffic': traffic, 'traffic_source': traffic_src,
                    'detections_last_step': sum(len(v) for v in y['scores']), 'setup_s': t_setup})
        del model, eng, x, y
        torch.cuda.empty_cache()
    return {'note': 'same run and box as the headline, outside its timed region; one GPU; dense reference graph, '
                    'forward_pipelined()', 'lines': out}


def slide_line(model, dev, args, S=16384, stride=384):
    """BASELINE.json configs[3] on ONE GPU under the same clock as the headline: the slide loop of the product
    (inference.tiled_inference: on-device crops of a resident uint8 slide, pipelined conv graphs, batched border rule, packed
    gather -- a no-op on one rank --, global NMS) over a synthetic 3 x S x S slide, 1849 tiles 512 / 384, batch 16; the headline's
    model (dense heads) and, in a second pass, the product default (score-gated heads; identical detections)."""
    from celldetection_amd import inference, util
    t_setup = time.perf_counter()
    crop, strd = (args.tile, args.tile), (stride, stride)
    slide = torch.randint(0, 256, (3, S, S), dtype=torch.uint8, device=dev, generator=torch.Generator(dev).manual_seed(3))
    ntiles = len(list(util.get_tiling_slices((S, S), crop, strd)[0]))
    kw = dict(crop_size=crop, strides=strd, batch_size=args.batch)
    edge = min(S, args.tile + stride * (int((4 * args.batch) ** .5 + 1) - 1))
    prev = model.sparse_heads
    res = {}
    try:
        for mode in (False, 'auto'):
            model.sparse_heads = mode
            inference.tiled_inference(model, slide[:, :edge, :edge], **kw)   # engine of this mode + its hipGraph slots
            inference.tiled_inference(model, slide[:, :2048, :2048], **kw)  # (ragged batch shapes of a small slide)
            torch.cuda.synchronize()
            if mode is False:
                t_setup = time.perf_counter() - t_setup
            t = {}
            t0 = time.perf_counter()
            out = inference.tiled_inference(model, slide, timings=t, **kw)
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0, t, out)
    finally:
        model.sparse_heads = prev
    dt, t, out = res[False]
    dtg, _, outg = res['auto']
    gf = GFLOP_PER_TILE.get(args.model)
    line = {'config': 'configs[3] on 1 GPU', 'model': args.model, 'batch': args.batch, 'tile': args.tile, 'dtype': 'bf16',
            'slide': [3, S, S], 'stride': stride, 'tiles_total': ntiles, 'value': ntiles / dt, 'unit': 'tiles/s', 'steps': 1,
            'ms_per_step': 1e3 * dt, 'tile_loop_ms': 1e3 * t['tiles'], 'gather_ms': 1e3 * t['gather'],
            'global_nms_ms': 1e3 * t['nms'], 'detections_gathered': t['detections_gathered'],
            'detections_final': t['detections_final'], 'peak': PEAK_BF16_TFLOPS,
            'frac': (ntiles / dt) * gf / 1e3 / PEAK_BF16_TFLOPS if gf else None,
            'gated': {'value': ntiles / dtg, 'ms_per_step': 1e3 * dtg,
                      'identical_to_dense': bool(all(torch.equal(outg[k], out[k]) for k in inference.KEYS)),
                      'mode': "model.sparse_heads = 'auto' (product default)"},
            'setup_s': t_setup}
    del slide, out, outg, res
    torch.cuda.empty_cache()
    return line


def gated_lines(model, x, args, timed, warm_engine, gf_tile, peak, densities=(.01, .10)):
    """The product default (``model.sparse_heads = 'auto'``: the forward paths evaluate the location / Fourier heads at the
    proposal pixels only -- CPN.forward reads nothing else of their maps, celldetection/models/cpn.py:613-637,710-734;
    outputs identical to the dense graph) timed in the same run at STATED proposal densities.  The density is set through the
    model's own ``score_thresh`` (the (1 - d) quantile of the synthetic score map), nothing else changes.  The roofline
    fraction prices EXECUTED FLOPs: conv graph without the two deferred heads + the gathered kernel's work on the
    proposals (padded to its 128-proposal workgroups)."""
    thresh0, sh0 = model.score_thresh, model.sparse_heads
    model.sparse_heads = False
    sc = model.core_forward(x)[0].flatten().float()
    head_px = sc.numel()
    res = []
    try:
        for d in densities:
            k = max(1, int(round(d * head_px)))
            model.score_thresh = float(torch.topk(sc, k + 1).values[-1].item())  # strictly-greater threshold: k proposals
            model.sparse_heads = 'auto'
            warm_engine()  # ('auto': the dense engine of the public core_forward; the gated one is captured by the warm-up below)
            geng = model.engine(x.device, _forward_path=True)
            if not geng.sparse:
                return {'note': 'the plan does not qualify for score-gated heads (heads on different features / strides)'}
            steps = max(5, args.steps // 2)
            dtg, evg, yg = timed(steps, geng.GRAPH_SLOTS + 2, True)
            n_prop = int((sc > model.score_thresh).sum().item())
            per_prop = sum(2. * op['cout'] * op['cin'] * op['k'] ** 2 + 2. * op['fuse']['cout'] * op['cout']
                           for op in geng.plan.ops if op.get('deferred'))
            sparse_gf = per_prop * ((n_prop + 127) // 128 * 128) / 1e9
            graph_gf = geng.executed_flops(args.batch, args.tile, args.tile) / 1e9
            step_ms = 1e3 * dtg / steps
            res.append({'target_density': d, 'density': n_prop / head_px, 'proposals_per_step': n_prop,
                        'score_thresh': model.score_thresh, 'value': args.batch * steps / dtg, 'unit': 'tiles/s',
                        'steps': steps, 'ms_per_step': step_ms,
                        'conv_graph_ms': sum(a.elapsed_time(b) for a, b in evg) / steps,
                        'detections_last_step': sum(len(v) for v in yg['scores']),
                        'executed_gflop_per_step': graph_gf + sparse_gf, 'sparse_kernel_gflop_per_step': sparse_gf,
                        'roofline_frac_executed': (graph_gf + sparse_gf) / step_ms / peak,
                        'algorithmic_frac': gf_tile * args.batch / step_ms / peak})
    finally:
        model.score_thresh, model.sparse_heads = thresh0, sh0
    return {'mode': "model.sparse_heads = 'auto' (product default): forward_pipelined(), location / Fourier heads evaluated at "
                    'the proposal pixels only; outputs identical to the dense graph (tests/test_gpu_sparse_heads.py)',
            'density_set_by': 'score_thresh = (1 - d) quantile of the synthetic score map',
            'frac_basis': 'EXECUTED FLOPs (conv graph without the two deferred heads + gathered head kernel) / whole step time',
            'lines': res}


def slide_workload(args, model, dev, world, rank, dist):
    """BASELINE.json configs[3]: tiled inference over one synthetic slide, tiles sharded over the ranks."""
    from celldetection_amd import inference, util
    if dist:
        import torch.distributed as td
    S, crop, stride = args.slide, (args.tile, args.tile), (args.stride, args.stride)
    # every rank holds the same slide (seeded, generated ON the device: no 805 MB host tensor per rank) resident in HBM as
    # uint8: 3 x 16384^2 = 805 MB
    slide = torch.randint(0, 256, (3, S, S), dtype=torch.uint8, device=dev, generator=torch.Generator(dev).manual_seed(3))
    ntiles = len(list(util.get_tiling_slices((S, S), crop, stride)[0]))
    kw = dict(crop_size=crop, strides=stride, batch_size=args.batch)
    warm = slide[:, :min(S, 4 * args.tile), :min(S, 4 * args.tile)]
    # warm-up: allocator / RCCL communicator / kernels / hipGraph slots.  A shape is captured when it is seen twice in a row
    # (cpn._Engine._graph_slot), so the warm-up needs a slide with several full batches per rank (the small one only yields
    # one full and one ragged batch: round 4 measured the three captures inside the first timed slide, 3 x 28 ms)
    edge = min(S, args.tile + args.stride * (int((4 * args.batch * world) ** .5 + 1) - 1))
    inference.tiled_inference(model, slide[:, :edge, :edge], **kw)
    for _ in range(max(args.warmup, 1)):
        inference.tiled_inference(model, warm, **kw)
    torch.cuda.synchronize()
    if dist:
        td.barrier()
        torch.cuda.synchronize()
    tim = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t = {}
        res = inference.tiled_inference(model, slide, timings=t, **kw)
        tim.append(t)
    torch.cuda.synchronize()
    if dist:
        td.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt] + [sum(t[k] for t in tim) for k in ('tiles', 'gather', 'nms')], dtype=torch.float64, device=dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt, t_tiles, t_gather, t_nms = (float(v) for v in tt.tolist())
    else:
        t_tiles, t_gather, t_nms = (sum(t[k] for t in tim) for k in ('tiles', 'gather', 'nms'))
    # the product default (sparse_heads = 'auto': score-gated location / Fourier heads, identical detections) over the same
    # slide, every rank, outside the headline's timed region
    gated = None
    if not args.no_extras:
        model.sparse_heads = 'auto'
        inference.tiled_inference(model, slide[:, :edge, :edge], **kw)  # (packs the gated plan, captures its graphs)
        inference.tiled_inference(model, warm, **kw)
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        res_g = inference.tiled_inference(model, slide, **kw)
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()
        dtg = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([dtg], dtype=torch.float64, device=dev)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            dtg = float(tt.item())
        gated = {'value': ntiles / dtg, 'unit': 'tiles/s', 'ms_per_step': 1e3 * dtg,
                 'detections_final': int(res_g['scores'].shape[0]),
                 'identical_to_dense': bool(res_g['scores'].shape == res['scores'].shape and torch.equal(res_g['boxes'], res['boxes'])),
                 'mode': "model.sparse_heads = 'auto' (product default), one pass over the same slide"}
        model.sparse_heads = False
    setup_ranks = setup_per_rank(dev, world, dist, keys=('import_torch', 'import_and_init', 'build_model'))
    if rank == 0:
        assert not dist or td.get_world_size() == world == args.gpus
        gf = GFLOP_PER_TILE.get(args.model)
        value = ntiles * args.steps / dt
        peak = PEAK_BF16_TFLOPS
        out = {
            'metric': f'tiles/sec (3x{args.tile}x{args.tile}) {args.model}', 'value': value, 'unit': 'tiles/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'{args.model} tiled inference over a synthetic 1x3x{S}x{S} uint8 slide, {ntiles} tiles '
                                   f'{args.tile}/{args.stride}, batch {args.batch}, tile-shard + RCCL gather + global NMS'
                                   + (' (BASELINE.json configs[3])' if (S, args.tile, args.stride) == (16384, 512, 384) else ''),
                       'tiles_total': ntiles, 'world_size_seen_by_rccl': td.get_world_size() if dist else None,
                       'parallelism': f'tile i -> rank i mod {world}; one packed all-gather of the detections; global NMS on every rank',
                       'tile_loop_ms': 1e3 * t_tiles / args.steps, 'gather_ms': 1e3 * t_gather / args.steps,
                       'global_nms_ms': 1e3 * t_nms / args.steps,
                       'detections_gathered': tim[-1].get('detections_gathered'),
                       'detections_final': int(res['scores'].shape[0])},
            'roofline': {'bound': 'mfma', 'achieved': value * gf / 1e3 if gf else None, 'peak': peak * world, 'unit': 'TFLOP/s',
                         'frac': (value * gf / 1e3) / (peak * world) if gf else None, 'traffic': None,
                         'kernel': 'whole slide loop (conv graphs + crops + post-processing + exchange): '
                                   'tiles/s x algorithmic GFLOP per tile'},
        }
        if gated is not None:
            out['gated'] = gated
        out['setup_s'] = {k: round(v, 2) for k, v in PHASES.items()}
        if setup_ranks is not None:
            out['setup_s_per_rank'] = setup_ranks
        out['host_threads_per_rank'] = host_threads(world)
        print(json.dumps(out), file=_JSON_OUT, flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == '__main__':
    main()
