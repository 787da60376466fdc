cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/calib
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r05_pytest_gpu_pass2.log 2>&1; tail -5 gpurun_out/r05_pytest_gpu_pass2.log
grep -h "north-star\|configs\[0\]" gpurun_out/r05_pytest_gpu_pass2.log | head
# calibration fixtures of the three bench models (one calibration run each), then the default line from the fixtures
for m in "CpnResNeXt101UNet 512 2" "CpnResNet18FPN 512 2" "CpnResNet50FPN 1024 1"; do set -- $m
  CPN_BENCH_CACHE=0 CPN_BENCH_SAVE_CALIB=gpurun_out/calib python -c "
import sys, torch; sys.path.insert(0, '.')
import bench
bench.build_model('$1', torch.device('cuda:0'), tile=$2, calib_tiles=$3)
print('calibrated $1')" 2>&1 | tail -1
done
mkdir -p tools/bench_calibration && cp gpurun_out/calib/*.npz tools/bench_calibration/
/usr/bin/time -v timeout 900 python bench.py --profile-layers > gpurun_out/r05_bench_n1_pass2.json 2> gpurun_out/r05_per_layer_timing_pass2.txt
cut -c1-600 gpurun_out/r05_bench_n1_pass2.json; grep -E "Elapsed|Maximum resident" gpurun_out/r05_per_layer_timing_pass2.txt
python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_n1_pass2.json').read().strip().splitlines()[-1])
print('setup_s', d.get('setup_s')); print('configs', json.dumps(d.get('configs'))[:1500]); print('gated', [(l['density'], l['value']) for l in d['gated']['lines']]); print('sync', d['sync_forward']['value'])"
