#!/bin/bash
# round 3, GPU job 18: C5 pair kernel with 8 column tiles per wave (4-wave workgroups of 512 beams) against the 8-wave workgroups
O=gpurun_out/r03; mkdir -p $O
python tools/ab_libs.py --workload c5 --paired 1 --rounds 5 w8=product ns8=variants/ns8/libdsabf.so,DSABF_WG_WAVES=4 ns8t1=variants/ns8/libdsabf.so,DSABF_WG_WAVES=4,DSABF_TSPLIT=1 ns8t4=variants/ns8/libdsabf.so,DSABF_WG_WAVES=4,DSABF_TSPLIT=4 2>&1 | tee -a $O/ab_c5_ns8.txt
python tools/ab_libs.py --workload c5 --paired 1 --detect contracted --rounds 3 w8=product ns8=variants/ns8/libdsabf.so,DSABF_WG_WAVES=4 2>&1 | tee -a $O/ab_c5_ns8.txt
