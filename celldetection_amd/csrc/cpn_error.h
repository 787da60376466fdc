// Error plumbing of libcpn_hip.so: int status codes + thread-local message (no exceptions across the C ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace cpn {
int fail(int code, const char *msg);             // records msg, returns code
int check_hip(hipError_t e, const char *where);  // 0 on success, else records "<where>: <hip error>" and returns (int) e
}  // namespace cpn
