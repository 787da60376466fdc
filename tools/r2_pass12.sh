#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do python tools/conv_microbench.py head7 head7x3 2>&1 | grep -v amdgpu; done > gpurun_out/r2l_head7x3.txt; cat gpurun_out/r2l_head7x3.txt
python bench.py --pipeline --no-cpu-baseline > gpurun_out/r02_bench_pipeline_n1.json 2>/dev/null; cut -c1-200 gpurun_out/r02_bench_pipeline_n1.json
P=/tmp/prof_cfg1; mkdir -p $P
G="python tools/run_graph_only_cfg1.py 5"
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/gtrace -o b -- $G) > $P/gtrace.log 2>&1
(timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -o b -- $G) > $P/fetch.log 2>&1
(timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -o b -- $G) > $P/write.log 2>&1
(echo "### tools/run_graph_only_cfg1.py 5 (CpnResNet18FPN, 5 conv-graph executions, batch 8 x 3x512x512, bf16): kernel trace, FETCH_SIZE, WRITE_SIZE"
 python tools/summarize_rocprof.py $P/gtrace $P/fetch $P/write) > gpurun_out/r02_configs1_rocprofv3_summary.txt 2>&1
python tools/summarize_rocprof.py --traffic-json $P/fetch $P/write 5 CpnResNet18FPN/b8/t512/bf16 gpurun_out/r02_traffic_cfg1.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/run_graph_only_cfg1.py 5 (r02); profiles/r02_configs1_rocprofv3_summary.txt"
grep -n "bilinear\|kernel  \|conv_igemm\|TOTAL" gpurun_out/r02_configs1_rocprofv3_summary.txt | cut -c1-160 | head -20
