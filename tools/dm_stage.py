#!/usr/bin/env python3
"""The DM-trial stage of the loop by itself (bench.dm_stage_record), for rocprofv3: a few dozen pushes of a production block through
a bf_dm_stream with the zero-copy feed.  What the kernel trace must show: dedisperse_dm_wide_kernel (+ dedisperse_dm_kernel for trial
groups outside a window) and, every few blocks, the slide of the carried rows -- and NO device-to-device copy of the pushed rows.
  python tools/dm_stage.py [n_dm] [rank world [blocks_per_push [queues]]]          (GPU box, repo root)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import dsabeamformer_amd as bfm

n_dm = int(sys.argv[1]) if len(sys.argv) > 1 else 64
share = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 1)
bpp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
queues = int(sys.argv[5]) if len(sys.argv) > 5 else 1
print(json.dumps(bench.dm_stage_record(torch, bfm, 0, n_dm, share, blocks_per_push=bpp, queues=queues, pushes=24 if queues == 1 else 48)))
