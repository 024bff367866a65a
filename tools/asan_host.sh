#!/bin/bash
# One-off sanitizer run of the host-side C++ (the host mirror, bf_runtime.cpp, bf_comm.cpp, bf_shmring.cpp) and of the oracle
# on the CPU (GPU AddressSanitizer is not available on this pool).  Temporarily swaps the built libraries; restores them
# however the script ends, and exits with the tests' status.
cd "$(dirname "$0")/.."
CXX=/opt/rocm/lib/llvm/bin/clang++
python -m dsabeamformer_amd.build >/dev/null || exit 1
cp dsabeamformer_amd/libdsabf.so /tmp/libdsabf_keep.so
cp oracle/liborc.so /tmp/liborc_keep.so
restore() {
    cp /tmp/libdsabf_keep.so dsabeamformer_amd/libdsabf.so; touch dsabeamformer_amd/libdsabf.so dsabeamformer_amd/beam
    cp /tmp/liborc_keep.so oracle/liborc.so; touch oracle/liborc.so
}
trap restore EXIT
C=dsabeamformer_amd/csrc
$CXX -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -shared -o /tmp/libhost_asan.so $C/bf_geometry.cpp $C/bf_generator.cpp \
    $C/bf_scheduler.cpp $C/bf_sinks.cpp $C/bf_host_c.cpp $C/bf_runtime.cpp $C/bf_comm.cpp $C/bf_shmring.cpp $C/bf_dada.cpp \
    dsabeamformer_amd/build/bf_kernels.hip.o dsabeamformer_amd/build/bf_dm_wide.hip.o dsabeamformer_amd/build/bf_fusedg.hip.o dsabeamformer_amd/build/bf_fused16_*.hip.o -lpthread -lrt -ldl || exit 1
cp /tmp/libhost_asan.so dsabeamformer_amd/libdsabf.so
status=0
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$($CXX -print-file-name=libclang_rt.asan-x86_64.so) \
    python -m pytest tests/test_host_cpu.py tests/test_abi_cpu.py tests/test_gather_plan_cpu.py -x -q \
    -k "not exports_every and not does_not_reference and not stand_in_library" || status=$?
cp /tmp/libdsabf_keep.so dsabeamformer_amd/libdsabf.so; touch dsabeamformer_amd/libdsabf.so
gcc -O1 -g -mavx2 -fopenmp -ffp-contract=off -fPIC -fsanitize=address,undefined -shared -o /tmp/liborc_asan.so oracle/dsabf_oracle.c -lm || exit 1
cp /tmp/liborc_asan.so oracle/liborc.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python -m pytest tests/test_oracle.py -x -q || status=$?
exit $status
