#!/bin/bash
# usage: tools_isa.sh <mangled-substring>   -- dumps resource usage + ISA of one kernel variant to /tmp/isa/sel.s
set -e
mkdir -p /tmp/isa && cd /root/repo/dsabeamformer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC $EXTRA -c bf_kernels.hip -o /tmp/isa/k.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name.*$1" | grep -E "Function Name|VGPRs:|Spill|Occupancy" | sed 's/.*remark: //; s/\[-Rpass.*//' 
S=/tmp/isa/bf_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v pat="^_ZN.*$1.*:" '$0 ~ pat {on=1} on {print} on && /s_endpgm/ {exit}' $S > /tmp/isa/sel.s
echo "lines $(wc -l < /tmp/isa/sel.s) mfma $(grep -c v_mfma /tmp/isa/sel.s) scratch $(grep -c scratch_ /tmp/isa/sel.s) valu_f32 $(grep -cE 'v_(fma|fmac|fmaak|fmamk|mul|add)_f32' /tmp/isa/sel.s) pk $(grep -c v_pk_ /tmp/isa/sel.s) ds_read $(grep -c ds_read /tmp/isa/sel.s) ds_write $(grep -c ds_write /tmp/isa/sel.s)"
