cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
V=celldetection_amd/build/variants
S="python tools/summarize_rocprof.py"
run() {  # name, env...
  local name=$1; shift
  echo "##### $name: time"; env "$@" python tools/conv_microbench.py dec3b dec3 k3 2>&1 | grep -v amdgpu
  (env "$@" timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$name -o b -- python tools/conv_microbench.py dec3b dec3 k3) > /tmp/pf_$name.log 2>&1
  echo "##### $name: FETCH_SIZE per dispatch (KiB raw)"; $S /tmp/pf_$name 2>&1 | grep -E "conv_igemm" | cut -c1-140
}
run s1f_off CPN_S1F=0
run s1f_on CPN_S1F=1
run s1f_map1_adjacent_ids CPN_S1F=1 CPN_HIP_LIB=$V/libcpn_s1f_map1.so
run s1f_map2_block_major CPN_S1F=1 CPN_HIP_LIB=$V/libcpn_s1f_map2.so
