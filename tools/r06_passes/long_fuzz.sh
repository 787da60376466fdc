#!/bin/bash
# Round 6: long runs of the seeded fuzzers (tests/fuzz_*.py; the checkers are the oracle and the fp32 torch conv) -> profiles/r06_fuzz_long.txt
cd "$GRAFT_REPO_ROOT"
(
for s in 31 32 33 34 35 36; do timeout 900 python -c "import sys; sys.path.insert(0,'tests'); import fuzz_conv; sys.exit(1 if fuzz_conv.run(1000, $s) else 0)" 2>&1 | grep -v amdgpu.ids | tail -4; done
for s in 41 42 43 44; do timeout 900 python -c "import sys; sys.path.insert(0,'tests'); import fuzz_conv; sys.exit(1 if fuzz_conv.run_fp8(500, $s) else 0)" 2>&1 | grep -v amdgpu.ids | tail -4; done
for s in 51 52 53; do timeout 1200 python tests/fuzz_model.py 200 $s 2>&1 | grep -v amdgpu.ids | grep -v " ok (" | tail -6; done
for s in 61 62 63 64; do timeout 900 python tests/fuzz_tiled.py 500 $s 2>&1 | grep -v amdgpu.ids | grep -v " ok (" | tail -4; done
for s in 71 72 73 74; do timeout 900 python tests/fuzz_post.py 500 $s 2>&1 | grep -v amdgpu.ids | tail -4; done
) > gpurun_out/r06_fuzz_long.txt 2>&1
tail -30 gpurun_out/r06_fuzz_long.txt
