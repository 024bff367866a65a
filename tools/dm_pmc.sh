#!/bin/bash
# usage (GPU box, repo root): tools/dm_pmc.sh <outdir-under-gpurun_out>   -- PMC passes over tools/dm_one.py (DM-trial dedispersion)
R=$PWD; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/tools/dm_one.py > $OUT/pass$i.log 2>&1
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "dedisperse_dm" not in k: continue
        agg[k.replace("(anonymous namespace)::", "").split("(")[0][-70:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fp:
    for k, d in agg.items():
        fp.write(k + "\n")
        for c, v in sorted(d.items()):
            fp.write("  %-32s mean %.6g  (n=%d)\n" % (c, sum(v)/len(v), len(v)))
print(open(out + "/summary.txt").read())
PY
