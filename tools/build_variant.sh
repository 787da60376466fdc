#!/bin/bash
# builds a tuning variant of libcpn_hip.so:  [SRC=conv_pair] tools/build_variant.sh NAME [-DCPN_... flags]
# (SRC: the translation unit the flags apply to, default conv_igemm)
# -> celldetection_amd/build/variants/libcpn_NAME.so  (select with CPN_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
D=celldetection_amd/build/variants; mkdir -p $D
SRC=${SRC:-conv_igemm}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c celldetection_amd/csrc/$SRC.hip -o $D/${SRC}_$NAME.o "$@"
OTHERS=$(ls celldetection_amd/build/*.o | grep -v "/$SRC.o\$" | grep -v "/conv_igemm_clock.o\$")  # (the probe library's object is not part of a variant)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/${SRC}_$NAME.o $OTHERS -o $D/libcpn_$NAME.so
echo $D/libcpn_$NAME.so
