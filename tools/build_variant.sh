#!/bin/bash
# builds a tuning variant of libcpn_hip.so:  tools/build_variant.sh NAME [-DCPN_... flags]
# -> celldetection_amd/build/variants/libcpn_NAME.so  (select with CPN_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
D=celldetection_amd/build/variants; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c celldetection_amd/csrc/conv_igemm.hip -o $D/conv_igemm_$NAME.o "$@"
OTHERS=$(ls celldetection_amd/build/*.o | grep -v '/conv_igemm.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/conv_igemm_$NAME.o $OTHERS -o $D/libcpn_$NAME.so
echo $D/libcpn_$NAME.so
