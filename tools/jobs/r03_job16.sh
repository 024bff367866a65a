#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
V=variants/occ4/libdsabf.so
timeout 900 python tools/ab_libs.py --workload c3 --paired 0 --rounds 3 occ3=product occ4=$V > $O/ab_c3_general_occ4.txt 2>&1
timeout 900 python tools/ab_libs.py --workload c3 --paired 0 --detect contracted --rounds 3 occ3=product occ4=$V >> $O/ab_c3_general_occ4.txt 2>&1
timeout 900 python tools/ab_libs.py --workload c3 --paired 0 --weights calibrated --rounds 3 occ3=product occ4=$V >> $O/ab_c3_general_occ4.txt 2>&1
timeout 900 python tools/ab_libs.py --workload c3 --paired 1 --rounds 2 occ3=product occ4=$V >> $O/ab_c3_general_occ4.txt 2>&1
cat $O/ab_c3_general_occ4.txt
