#!/bin/bash
# resource usage + instruction census of the wide DM kernel (bf_dm_wide.hip) -> /tmp/isa/dw.s
mkdir -p /tmp/isa; cd /root/repo/dsabeamformer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -I../../include $EXTRA -c bf_dm_wide.hip -o /tmp/isa/dw.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "wide_kernel" | grep -E "VGPRs:|Spill" | sed 's/.*remark: //; s/\[-R.*//'
S=/tmp/isa/bf_dm_wide-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN5dsabf12_GLOBAL__N_125dedisperse_dm_wide_kernel.*:/{on=1} on{print} on&&/s_endpgm/{exit}' $S > /tmp/isa/dw.s
echo "lines $(wc -l < /tmp/isa/dw.s) ds_read $(grep -c ds_read /tmp/isa/dw.s) ds_write $(grep -c ds_write /tmp/isa/dw.s) pk_add $(grep -c v_pk_add_f32 /tmp/isa/dw.s) v_add_f32 $(grep -c 'v_add_f32' /tmp/isa/dw.s) scratch $(grep -c scratch_ /tmp/isa/dw.s) barrier $(grep -c s_barrier /tmp/isa/dw.s)"
