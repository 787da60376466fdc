#!/bin/bash
# Round 6: long runs of the seeded fuzzers (tests/fuzz_*.py; the checkers are the oracle and the fp32 torch conv) -> profiles/r06_fuzz_long.txt
cd "$GRAFT_REPO_ROOT"
O=${FUZZ_OFF:-0}   # seed offset: FUZZ_OFF=100 tools/r06_passes/long_fuzz.sh = another run
OUT=gpurun_out/r06_fuzz_long${FUZZ_OFF:+_$FUZZ_OFF}.txt
(
for s in $((31+O)) $((32+O)) $((33+O)) $((34+O)) $((35+O)) $((36+O)); do timeout 900 python -c "import sys; sys.path.insert(0,'tests'); import fuzz_conv; sys.exit(1 if fuzz_conv.run(1000, $s) else 0)" 2>&1 | grep -v amdgpu.ids | tail -4; done
for s in $((41+O)) $((42+O)) $((43+O)) $((44+O)); do timeout 900 python -c "import sys; sys.path.insert(0,'tests'); import fuzz_conv; sys.exit(1 if fuzz_conv.run_fp8(500, $s) else 0)" 2>&1 | grep -v amdgpu.ids | tail -4; done
for s in $((51+O)) $((52+O)) $((53+O)); do timeout 1200 python tests/fuzz_model.py 200 $s 2>&1 | grep -v amdgpu.ids | grep -v " ok (" | tail -6; done
for s in $((61+O)) $((62+O)) $((63+O)) $((64+O)); do timeout 900 python tests/fuzz_tiled.py 500 $s 2>&1 | grep -v amdgpu.ids | grep -v " ok (" | tail -4; done
for s in $((71+O)) $((72+O)) $((73+O)) $((74+O)); do timeout 900 python tests/fuzz_post.py 500 $s 2>&1 | grep -v amdgpu.ids | tail -4; done
) > $OUT 2>&1
tail -30 $OUT
