set -x
cd /root/repo
python bench.py > gpurun_out/r01_c3_bench.json 2> gpurun_out/r01_c3_bench.err
python bench.py --workload prod --no-cpu-baseline > gpurun_out/r01_prod_bench.json 2>/dev/null
python bench.py --workload c2 --no-cpu-baseline > gpurun_out/r01_c2_bench.json 2>/dev/null
python bench.py --workload c5 --units 16 --no-cpu-baseline > gpurun_out/r01_c5_bench.json 2>/dev/null
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_c3.log 2>&1
for WL in c5 c2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$WL -- python3 $R/bench.py --workload $WL --units $([ $WL = c5 ] && echo 16 || echo 128) --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_$WL.log 2>&1
  find $R/gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|fused16' {} > $R/gpurun_out/r01_${WL}_kernel_stats.csv"
  rm -rf $R/gpurun_out/prof_$WL
done
cd $R
bash tools/pmc.sh pmc_c3 > /dev/null 2>&1
find gpurun_out/prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r01_c3_kernel_stats.csv
find gpurun_out/prof_c3 -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'head -40 {} > gpurun_out/r01_c3_kernel_trace_head.csv'
cp gpurun_out/pmc_c3/summary.txt gpurun_out/r01_c3_pmc_summary.txt
rm -rf gpurun_out/prof_c3
cat gpurun_out/r01_c3_bench.json
