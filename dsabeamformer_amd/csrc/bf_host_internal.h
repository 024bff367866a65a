// bf_host_internal.h -- helpers shared by the host-mirror translation units (not installed, not part of the ABI).
#pragma once
#include <functional>

struct bf_comm;

namespace dsabf {
// Splits [0, n) over the machine's hardware threads (std::thread; OpenMP is not assumed in the product build).
void parallel_for(long n, const std::function<void(long, long)>& body);
// Sets the calling thread's bf_last_error() text; returns `code` (bf_runtime.cpp).
int set_error(int code, const char* msg);
// getenv(name) in a process that says DSABF_LAB=1, NULL anywhere else (bf_kernels.hip): the measurement / test switches.  The
// production allow-list read with plain getenv is DSABF_RCCL_LIB, DSABF_THREADS, DSABF_COALESCE, DSABF_PAIRED.
bool lab_mode();
const char* lab_getenv(const char* name);
// Control plane of a sharded run (bf_comm.cpp): every rank contributes a flag, every rank learns whether ALL are set -- one tiny
// all-gather on the communicator (16 bytes per rank), blocking.  world 1 / NULL: *all = ok.
int comm_all_ok(bf_comm* c, bool ok, bool* all);
}  // namespace dsabf
