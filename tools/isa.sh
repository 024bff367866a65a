#!/bin/bash
# usage: tools/isa.sh <mangled-substring> [unit]   -- resource usage + ISA of one fused16_kernel instantiation -> /tmp/isa/sel.s
#   unit = the translation unit bf_fused16_<unit>.hip: a64 (default) | a100 | a128 | k1p16 | k1p4 | k2p16 | k2p4, and the wide
#          launches of the two-k-step classes <class>_w8 (general kernel, 8-wave workgroups), <class>_w8p (conjugate-pair kernel,
#          8-wave workgroups), <class>_s8 (conjugate-pair kernel, 8 output slots per wave)
#   e.g.  tools/isa.sh 'Li64ELi32ELb0ELi0ELb0E'        (64 antennas, n_ipo 32, canonical, general)
#         EXTRA=-DDSABF_X=1 tools/isa.sh 'Li100ELi32ELb0ELi0ELb1E' a100_s8
# The flags are the SHIPPED ones: dsabeamformer_amd.build.flags_for(source), i.e. the per-file LLVM scheduling strategy included
# (max-ilp; *_w8: iterative-maxocc; *_w8p: iterative-ilp).  For the shipped objects themselves (no re-compile): tools/isa_report.py.
set -e
CLS=${1:+${2:-a64}}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/dsabeamformer_amd/csrc/bf_fused16_$CLS.hip
FLAGS=$(cd $ROOT && python3 -c "from dsabeamformer_amd import build; print(' '.join(build.flags_for('$SRC')))")
mkdir -p /tmp/isa && cd $ROOT/dsabeamformer_amd/csrc
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $SRC -o /tmp/isa/k.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Function Name.*$1" | grep -E "Function Name|VGPRs:|Spill|Occupancy|LDS Size" | sed 's/.*remark: //; s/\[-Rpass.*//'
S=/tmp/isa/bf_fused16_$CLS-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v pat="^_ZN.*$1.*:" '$0 ~ pat {on=1} on {print} on && /s_endpgm/ {exit}' $S > /tmp/isa/sel.s
echo "lines $(wc -l < /tmp/isa/sel.s) mfma $(grep -c v_mfma /tmp/isa/sel.s) scratch $(grep -c scratch_ /tmp/isa/sel.s) valu_f32 $(grep -cE 'v_(fma|fmac|fmaak|fmamk|mul|add)_f32' /tmp/isa/sel.s) pk $(grep -c v_pk_ /tmp/isa/sel.s) ds_read $(grep -c ds_read /tmp/isa/sel.s) ds_write $(grep -c ds_write /tmp/isa/sel.s)"
