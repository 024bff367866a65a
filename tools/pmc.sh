#!/bin/bash
# usage (on the GPU box, from the repo root): tools_pmc.sh <outdir-under-gpurun_out> [bench args...]
# Runs separate rocprofv3 --pmc passes (never combined with trace domains) and leaves CSVs under gpurun_out/<outdir>/.
R=$PWD; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ARGS="${PMC_BENCH_ARGS:---steps 6 --warmup 2 --no-cpu-baseline --no-extras} $@"   # PMC_BENCH_ARGS: e.g. without --no-extras for the auxiliary kernels
export PMC_FILTER="${PMC_FILTER:-fused|expand_kernel}"                            # kernels kept in the summary (regex on the name)
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pass$i -- python3 $R/bench.py $ARGS > $OUT/pass$i.log 2>&1
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os, re
keep = re.compile(os.environ.get("PMC_FILTER", "fused|expand_kernel"))
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not keep.search(k): continue
        agg[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        # the dispatch's own duration in the pass that counted the cycles: clock = GRBM_GUI_ACTIVE / 8 / this
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            agg[k.split("(")[0][-60:]]["_KERNEL_NS_GRBM_PASS"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
# first line: the key of this pass -- kernel build id (bf_version()), instantiation, launch -- as bench.py printed it on stderr;
# bench.py quotes a committed summary beside its timings only if this line is its own run's (pmc_for_launch)
key = None
for f in sorted(glob.glob(out + "/pass*.log")):
    for line in open(f, errors="replace"):
        if line.startswith("bench.py: pmc_key "):
            key = line[len("bench.py: pmc_key "):].strip()
            break
    if key:
        break
with open(out + "/summary.txt", "w") as fp:
    fp.write("# pmc_key %s\n" % (key or "unknown"))
    for k, d in agg.items():
        fp.write(k + "\n")
        for c, v in sorted(d.items()):
            v = v[len(v)//3:] if len(v) > 3 else v   # skip warm-up dispatches
            fp.write("  %-32s mean %.6g  (n=%d)\n" % (c, sum(v)/len(v), len(v)))
print(open(out + "/summary.txt").read())
PY
