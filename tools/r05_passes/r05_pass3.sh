cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
S=$(date +%s)
timeout 900 python bench.py --profile-layers > gpurun_out/r05_bench_n1_pass3.json 2> gpurun_out/r05_per_layer_timing_pass3.txt
echo "bench wall: $(( $(date +%s) - S )) s"
cut -c1-400 gpurun_out/r05_bench_n1_pass3.json
python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_n1_pass3.json').read().strip().splitlines()[-1])
print('setup_s', d.get('setup_s')); print('configs', json.dumps(d.get('configs'))[:2500]); print('gated', [(l['density'], l['value']) for l in d['gated']['lines']]); print('sync', d['sync_forward']['value']); print('roofline', {k: v for k, v in d['roofline'].items() if k in ('frac','launch_ms','executed_frac')}, d['roofline']['backbone_stack']['frac'], d['roofline']['dominant_kernel'])"
python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "full_size_properties or full_width" 2>&1 | grep -h "north-star\|configs\[\|passed\|failed"
