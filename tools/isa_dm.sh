#!/bin/bash
# resource usage + instruction census of the wide DM kernel (bf_dm_wide.hip) -> /tmp/isa/dw.s; flags = build.flags_for(source)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/dsabeamformer_amd/csrc/bf_dm_wide.hip
FLAGS=$(cd $ROOT && python3 -c "from dsabeamformer_amd import build; print(' '.join(build.flags_for('$SRC')))")
mkdir -p /tmp/isa; cd $ROOT/dsabeamformer_amd/csrc
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $SRC -o /tmp/isa/dw.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "wide_kernel" | grep -E "VGPRs:|Spill" | sed 's/.*remark: //; s/\[-R.*//'
S=/tmp/isa/bf_dm_wide-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN5dsabf12_GLOBAL__N_125dedisperse_dm_wide_kernel.*:/{on=1} on{print} on&&/s_endpgm/{exit}' $S > /tmp/isa/dw.s
echo "lines $(wc -l < /tmp/isa/dw.s) ds_read $(grep -c ds_read /tmp/isa/dw.s) ds_write $(grep -c ds_write /tmp/isa/dw.s) pk_add $(grep -c v_pk_add_f32 /tmp/isa/dw.s) v_add_f32 $(grep -c 'v_add_f32' /tmp/isa/dw.s) scratch $(grep -c scratch_ /tmp/isa/dw.s) barrier $(grep -c s_barrier /tmp/isa/dw.s)"
