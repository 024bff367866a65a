#!/bin/bash
# In-kernel shader clock of the fused kernel under the bench load (MI355X_MICROARCH.md, DVFS item 6): a diagnostic build
# (-DDSABF_CLOCKPROBE=1: every workgroup stamps s_memtime / s_memrealtime around its chunk loop and overwrites
# out[blockIdx.x] with the ratio in GHz), two seconds of back-to-back launches on random data, median over workgroups.
# Restores the product build afterwards.  usage (GPU box, repo root): tools/clock_probe.sh [workload] [units]
WL=${1:-c3}; UNITS=${2:-128}
restore() { DSABF_EXTRA_FLAGS="" python -m dsabeamformer_amd.build --force > /dev/null 2>&1; }
trap restore EXIT     # the probe build corrupts out[blockIdx.x]: never leave it in place, however the script ends
DSABF_EXTRA_FLAGS="-DDSABF_CLOCKPROBE=1" python -m dsabeamformer_amd.build --force > /dev/null 2>&1 || exit 1
export DSABF_EXTRA_FLAGS="-DDSABF_CLOCKPROBE=1"   # the python below must see the same flag set (build/flags.stamp)
python - "$WL" "$UNITS" <<'PY'
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
import bench
import dsabeamformer_amd as bfm
wl, units = sys.argv[1], int(sys.argv[2])
n_avg, n_out = bench.geometry(wl)
for label, env in (("paired", None), ("general", "0")):
    import os
    if env is None: os.environ.pop("DSABF_PAIRED", None)
    else: os.environ["DSABF_PAIRED"] = env
    cfg = bfm.production_config(n_avg=n_avg, n_out_per_gemm=n_out, n_freq=1024 if wl == "c5" else 256)
    if wl == "c5":
        cfg.n_ant, cfg.n_beams = 100, 512
    bf = bfm.Beamformer(cfg)
    bf.set_weights(bench.product_weights(cfg, 0))
    n_time = n_out * cfg.n_pol * cfg.n_avg
    d_in = [torch.randint(0, 256, (units * cfg.n_freq * n_time * cfg.n_ant,), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_out = torch.empty(units * n_out * cfg.n_freq * cfg.n_beams, dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t0 = time.time(); i = 0
    while time.time() - t0 < 2.0:
        bf.beamform(d_in[i & 1], units, d_out, s); i += 1
        if i % 50 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
    info = bf.kernel_info(units)
    ghz = d_out[:info["grid"]].cpu().numpy()
    print(json.dumps({"workload": wl, "kernel": info["kernel"], "launches": i, "in_kernel_clock_GHz": {
        "median": float(np.median(ghz)), "p10": float(np.percentile(ghz, 10)), "p90": float(np.percentile(ghz, 90))}}))
    bf.close()
PY
