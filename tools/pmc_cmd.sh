#!/bin/bash
# usage (GPU box, repo root): tools/pmc_cmd.sh <outdir-under-gpurun_out> <kernel-name-substring> <program> [args...]
# the issue-side counters of ONE program (a micro-benchmark binary, or python3 tools/generic_one.py ...), one rocprofv3 --pmc pass
# per group (never combined with trace domains); prints mean counter values per kernel whose name contains the substring
R=$PWD; OUT=$R/gpurun_out/$1; PAT=$2; shift 2
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- "$@" > $OUT/stats.log 2>&1
cd $R
python3 - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        agg[k.split("(")[0][-70:] + " grid " + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]: dur[r["Name"].split("(")[0][-70:]] = float(r["AverageNs"])
with open(out + "/summary.txt", "w") as fp:
    for k, d in sorted(agg.items()):
        fp.write(k + "\n")
        m = {c: sum(v[len(v)//3:]) / len(v[len(v)//3:]) for c, v in d.items()}
        for c, v in sorted(m.items()): fp.write("  %-28s %.6g\n" % (c, v))
        if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 1024
            fp.write("  -> matrix pipe busy %.3f of SIMD-cycles; VALU issue active %.3f; waves in s_waitcnt %.3f, waiting to issue %.3f of wave-cycles\n" % (
                m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4 * m.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles,
                m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
    for k, v in dur.items(): fp.write("avg ns %-70s %.0f\n" % (k, v))
print(open(out + "/summary.txt").read())
PY
