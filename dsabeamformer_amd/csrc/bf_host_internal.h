// bf_host_internal.h -- helpers shared by the host-mirror translation units (not installed, not part of the ABI).
#pragma once
#include <functional>

namespace dsabf {
// Splits [0, n) over the machine's hardware threads (std::thread; OpenMP is not assumed in the product build).
void parallel_for(long n, const std::function<void(long, long)>& body);
// Sets the calling thread's bf_last_error() text; returns `code` (bf_runtime.cpp).
int set_error(int code, const char* msg);
// getenv(name) in a process that says DSABF_LAB=1, NULL anywhere else (bf_kernels.hip): the measurement / test switches.  The
// production allow-list read with plain getenv is DSABF_RCCL_LIB, DSABF_THREADS, DSABF_COALESCE, DSABF_PAIRED.
bool lab_mode();
const char* lab_getenv(const char* name);
}  // namespace dsabf
