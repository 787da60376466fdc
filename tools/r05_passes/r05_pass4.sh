cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_model.py -m gpu -x -q -s -k "fp8" 2>&1 | grep -v "^$" | tail -60
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fp8 or two_workgroups" 2>&1 | tail -3
for sp in 1 0; do echo "### fp8 configs[2], subpixel=$sp"; python bench.py --precision fp8 --no-cpu-baseline --steps 20 $( [ $sp = 0 ] && echo --no-subpixel ) 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['executed_frac'])"; done
