#!/bin/bash
# One-off ThreadSanitizer run of the host-side C++ on the CPU: the asynchronous sinks (delivery thread), the shared-memory
# rings, the threaded generator.  Swaps the built library, restores it however the script ends, exits with the tests' status.
cd "$(dirname "$0")/.."
CXX=/opt/rocm/lib/llvm/bin/clang++
python -m dsabeamformer_amd.build >/dev/null || exit 1
cp dsabeamformer_amd/libdsabf.so /tmp/libdsabf_keep.so
restore() { cp /tmp/libdsabf_keep.so dsabeamformer_amd/libdsabf.so; touch dsabeamformer_amd/libdsabf.so dsabeamformer_amd/beam; }
trap restore EXIT
C=dsabeamformer_amd/csrc
$CXX -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=thread -fno-omit-frame-pointer -Iinclude \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -shared -o /tmp/libhost_tsan.so $C/bf_geometry.cpp $C/bf_generator.cpp \
    $C/bf_scheduler.cpp $C/bf_sinks.cpp $C/bf_host_c.cpp $C/bf_runtime.cpp $C/bf_comm.cpp $C/bf_shmring.cpp $C/bf_dada.cpp \
    dsabeamformer_amd/build/bf_kernels.hip.o dsabeamformer_amd/build/bf_dm_wide.hip.o dsabeamformer_amd/build/bf_fusedg.hip.o dsabeamformer_amd/build/bf_fused16_*.hip.o -lpthread -lrt -ldl || exit 1
cp /tmp/libhost_tsan.so dsabeamformer_amd/libdsabf.so
TSAN_OPTIONS="report_signal_unsafe=0 halt_on_error=0 exitcode=66 log_path=/tmp/tsan_report" LD_PRELOAD=$($CXX -print-file-name=libclang_rt.tsan-x86_64.so) \
    python -m pytest tests/test_host_cpu.py -x -q -k "sink or ring or shm or generator or observation" 2>&1 | tee /tmp/tsan.log | tail -5
status=${PIPESTATUS[0]}
grep -c "WARNING: ThreadSanitizer" /tmp/tsan.log
exit $status
