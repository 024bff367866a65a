#!/bin/bash
# usage: tools/isa.sh <mangled-substring> [class]   -- resource usage + ISA of one fused16_kernel instantiation -> /tmp/isa/sel.s
#   class = a64 (default) | a100 | a128 | k1p16 | k1p4 | k2p16 | k2p4: the translation unit bf_fused16_<class>.hip
#   e.g.  tools/isa.sh 'Li64ELi32ELb0ELi0ELb0E'        (64 antennas, n_ipo 32, canonical, general)
#         EXTRA=-DDSABF_X=1 tools/isa.sh 'Li100ELi32ELb0ELi0ELb1E' a100
set -e
CLS=${2:-a64}
mkdir -p /tmp/isa && cd /root/repo/dsabeamformer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -I../../include $EXTRA -c bf_fused16_$CLS.hip -o /tmp/isa/k.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name.*$1" | grep -E "Function Name|VGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: //; s/\[-Rpass.*//'
S=/tmp/isa/bf_fused16_$CLS-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v pat="^_ZN.*$1.*:" '$0 ~ pat {on=1} on {print} on && /s_endpgm/ {exit}' $S > /tmp/isa/sel.s
echo "lines $(wc -l < /tmp/isa/sel.s) mfma $(grep -c v_mfma /tmp/isa/sel.s) scratch $(grep -c scratch_ /tmp/isa/sel.s) valu_f32 $(grep -cE 'v_(fma|fmac|fmaak|fmamk|mul|add)_f32' /tmp/isa/sel.s) pk $(grep -c v_pk_ /tmp/isa/sel.s) ds_read $(grep -c ds_read /tmp/isa/sel.s) ds_write $(grep -c ds_write /tmp/isa/sel.s)"
