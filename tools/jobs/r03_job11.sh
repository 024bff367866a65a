#!/bin/bash
# round 3, GPU job 11: DM kernel with the fit decision folded in (tests + A/B + trace + counters); resident-workgroup sweep of
# the fused kernel (DSABF_LDS_PAD); long fuzz of the fused kernel on the final build
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "dedisperse or dm or pulse or burst or gather_detected" > $O/gputest11.log 2>&1; echo "pytest rc $?" | tee -a $O/gputest11.log; tail -2 $O/gputest11.log
timeout 600 python tools/dm_ab.py 2>&1 | grep -v amdgpu.ids > $O/dm_ab.txt; cat $O/dm_ab.txt | cut -c1-220
mkdir -p gpurun_out/r03p; python tools/bench_stages.py 2>/dev/null > gpurun_out/r03p/r03_stage_kernels.json; cat gpurun_out/r03p/r03_stage_kernels.json
R=$PWD; P=$R/gpurun_out/r03p; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P/prof_dm -- python3 $R/tools/dm_one.py > $P/prof_dm.log 2>&1
find $P/prof_dm -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c "grep -E 'Name|dedisperse' {} > $P/r03_dm_kernel_stats.csv"
rm -rf $P/prof_dm $P/prof_dm.log
cd $R; bash tools/dm_pmc.sh r03p/pmc_dm > /dev/null 2>&1 && cp $P/pmc_dm/summary.txt $P/r03_dm_pmc_summary.txt; rm -rf $P/pmc_dm
cat $P/r03_dm_kernel_stats.csv
cat $O/ab_c3_paired_resident.txt $O/ab_c3_general_resident.txt
