// fp8 (OCP e4m3) build of the implicit-GEMM convolution: same source as conv_igemm.hip, compiled with CPN_FP8 = 1
// (namespace cpn_fp8; entry point cpn::launch_conv_fp8).  See the header comment of conv_igemm.hip.
#define CPN_FP8 1
#include "conv_igemm.hip"
