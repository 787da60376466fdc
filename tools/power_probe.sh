#!/bin/bash
# Samples socket power / clocks with rocm-smi while the bench loop runs (NOT the committed bench line: polling the
# SMU perturbs the run).  tools/power_probe.sh  -> gpurun_out/power_probe.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/power_probe.txt
(rocm-smi --showmaxpower 2>&1 | grep -i "max graphics") > $OUT
python bench.py --steps 500 --warmup 5 --no-cpu-baseline > /tmp/pp_bench.json 2>/dev/null &
BP=$!
t0=$(date +%s.%N)
while kill -0 $BP 2>/dev/null; do
  s=$(rocm-smi --showpower --showclocks 2>&1 | grep -i "Package Power\|sclk" | sed 's/.*: //' | tr '\n' ' ')
  echo "t=$(echo "$(date +%s.%N) - $t0" | bc | cut -c1-5)s  $s" >> $OUT
  sleep 0.5
done
echo "--- bench line" >> $OUT; cut -c1-200 /tmp/pp_bench.json >> $OUT
sort -t' ' -k4 -n $OUT | tail -3; grep -c "t=" $OUT; tail -2 $OUT | cut -c1-200
