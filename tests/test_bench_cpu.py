"""bench.py's host logic without a GPU: defaults of the contract, the committed PMC summaries it quotes, the watchdog."""
import json
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_defaults_are_the_contract(monkeypatch):
    """No flags = 1 GPU and a K / W that finish within minutes; --gpus N without a launcher starts one (below)."""
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.workload, a.detect, a.units) == (1, "c3", "canonical", 128)
    assert a.steps * a.units * 16 >= 1000 and a.warmup == 25          # >= 1000 gemm-units after 25 warm-ups
    assert bench.geometry("c3") == (16, 16) and bench.geometry("c2") == (1, 8) and bench.geometry("c5") == (16, 8)
    pos, dirs = bench.grid_100()
    assert pos.shape == (100, 3) and dirs.shape == (512, 2)


def test_committed_pmc_summaries_give_the_quoted_traffic_and_mfma_busy():
    """roofline.traffic / mfma_busy_frac come from profiles/r04_*_pmc_summary.txt: the files must parse, and the derived
    numbers must be what DESIGN.md section 4 quotes (traffic within 1 % of the algorithmic bytes for C3 and C2)."""
    c3 = bench.pmc_summary("r04_c3_paired_pmc_summary.txt")
    gen = bench.pmc_summary("r04_c3_general_pmc_summary.txt")
    c5 = bench.pmc_summary("r04_c5_pmc_summary.txt")
    c5g = bench.pmc_summary("r04_c5_general_pmc_summary.txt")
    c2 = bench.pmc_summary("r04_c2_pmc_summary.txt")
    assert c3 and gen and c5 and c5g and c2
    # the general kernel of the 100-antenna geometry keeps the matrix pipe busy more than half the time (22 % of that on the
    # zero weights behind antenna 99); round 3: 3-fragment image, 8-wave workgroups, iterative-maxocc scheduling: 58 -> 63 %
    assert 0.55 < bench.pmc_mfma_busy(c5g) < 0.68 and abs(c5g["SQ_INSTS_VALU_MFMA_I8"] / (2 * c5["SQ_INSTS_VALU_MFMA_I8"]) - 1) < 1e-4
    alg_c3 = (64 * 32 * 256 + 4 * 256 * 256) * 128 * 16          # bytes per launch: SURVEY.md 8d x 2048 beam-blocks
    assert abs(bench.pmc_traffic(c3) / alg_c3 - 1) < 0.01 and abs(bench.pmc_traffic(gen) / alg_c3 - 1) < 0.02
    alg_c2 = (64 * 2 * 256 + 4 * 256 * 256) * 128 * 8
    assert abs(bench.pmc_traffic(c2) / alg_c2 - 1) < 0.02
    assert 0.25 < bench.pmc_mfma_busy(c3) < 0.32 and 0.47 < bench.pmc_mfma_busy(gen) < 0.55      # pair: half the MFMAs
    assert 0.36 < bench.pmc_mfma_busy(c5) < 0.45 and bench.pmc_mfma_busy(c2) < 0.15     # C5: 8 slots per wave, fewer parked waves
    # the pair kernel executes half the algorithmic int8 ops: SQ_INSTS_VALU_MFMA_I8 x 16*16*64*2 ops
    assert abs(c3["SQ_INSTS_VALU_MFMA_I8"] * 32768 / (8 * 256 * 64 * 32 * 256 * 2048 / 2) - 1) < 0.01
    assert abs(gen["SQ_INSTS_VALU_MFMA_I8"] * 32768 / (8 * 256 * 64 * 32 * 256 * 2048) - 1) < 0.01
    assert bench.pmc_summary("no_such_file.txt") == {} and bench.pmc_traffic({}) is None and bench.pmc_mfma_busy({}) is None


def test_watchdog_prints_exactly_one_line(capsys):
    out = {"metric": "m", "value": 1.0, "roofline": {"frac": 0.5}}
    w = bench.Watchdog(out, 3600.0)
    out["extra"] = {"a": 1}
    w.finish()
    w.finish()                      # a second call (or a late timer) prints nothing
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["extra"] == {"a": 1} and "truncated" not in lines[0]
    w2 = bench.Watchdog(None, 3600.0)   # ranks other than 0 hold no record
    w2.cancel()


def test_watchdog_expiry_prints_the_partial_record_and_exits_zero():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "out = {'metric': 'm', 'value': 2.0, 'roofline': {'frac': 0.5}}\n"
            "w = bench.Watchdog(out, 0.2)\n"
            "out['half'] = object()      # a record the main thread has not finished: not serialisable\n"
            "time.sleep(30)\n"
            "print('never reached')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "never reached" not in r.stdout
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] == 2.0 and "truncated" in d and "half" not in d


def test_multi_gpu_request_without_a_launcher_starts_the_launcher_itself(monkeypatch):
    """`python bench.py --gpus N` (the shape of the driver's N = 1 command) must not die at argument
    parsing.  With no WORLD_SIZE in the environment it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD (never an exec) with the caller's own flags, on 127.0.0.1, and exits with the child's status."""
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    at = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[at + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    # the hand-over itself, on this GPU-less box: the launcher is a subprocess, gets the flags, and its status comes back
    calls = {}

    def fake_call(c, env=None):
        calls["cmd"], calls["env"] = c, env
        return 7

    import subprocess as sp
    monkeypatch.setattr(sp, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DSABF_BENCH_ONE_GPU", "1")            # (the GPU count is not asked for in the ranks-share-GPU-0 test mode)
    try:
        bench.main()
        raise AssertionError("self_launch returned")
    except SystemExit as e:
        assert e.code == 7
    assert calls["cmd"][-4:] == ["--gpus", "2", "--steps", "3"] and calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert int(calls["cmd"][calls["cmd"].index("--master-port") + 1]) > 0


def test_multi_gpu_request_on_a_box_with_fewer_gpus_says_so():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSABF_BENCH_ONE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "this node shows 0 GPU(s)" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_committed_bench_lines_agree_with_the_committed_rocprof_kernel_stats():
    """profiles/: the HIP-event average of the dominant kernel in each committed bench line must agree with the AverageNs of
    the same kernel in the rocprofv3 --kernel-trace --stats summary committed beside it (same command, another run and,
    for rocprof, a profiled one: 5 %; 8 % for the 50-us launches of C2), and roofline.frac must follow from it."""
    import csv

    for wl, stats, units_blocks in (("c3", "r04_c3_paired_kernel_stats.csv", 2048), ("c5", "r04_c5_kernel_stats.csv", 128),
                                    ("c2", "r04_c2_kernel_stats.csv", 1024)):
        line = [l for l in open(os.path.join(ROOT, "profiles", "r04_%s_bench.json" % wl)) if l.startswith("{")][-1]
        d = json.loads(line)
        roof = d["roofline"]
        rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", stats))) if "fused16_kernel" in r["Name"]]
        assert len(rows) == 1, (wl, rows)
        rocprof_ms = float(rows[0]["AverageNs"]) * 1e-6
        # C2's launches last 50 us: the events' own cost (~2-3 us between the two records) shows as 4-6 %
        assert abs(rocprof_ms / roof["kernel_ms_avg"] - 1) < (0.08 if wl == "c2" else 0.05), (wl, rocprof_ms, roof["kernel_ms_avg"])
        assert d["config"]["beam_blocks_per_step"] == units_blocks
        per_launch = roof["algorithmic_ops_per_launch"] if roof["bound"] == "mfma" else roof["algorithmic_bytes_per_launch"]
        scale = 1e12 if roof["bound"] == "mfma" else 1e9
        assert abs(per_launch / (roof["kernel_ms_avg"] * 1e-3) / scale / roof["peak"] / roof["frac"] - 1) < 1e-6
        assert roof["traffic"] is not None and roof["pmc_source"].startswith("profiles/r0")
        assert set(roof["from_committed_profile"]) == {"traffic", "mfma_busy_frac"}     # labelled: not observed in that run
        assert d["ms_per_step"] >= roof["kernel_ms_avg"] * 0.999          # the whole step cannot be shorter than its kernel
        # the measured peak beside the nominal one, from the same run (a pure MFMA loop / a pure streaming kernel)
        assert 0.6 * roof["peak"] < roof["peak_measured"] <= roof["peak"] * 1.02
        assert abs(roof["achieved"] / roof["peak_measured"] / roof["frac_of_measured_peak"] - 1) < 1e-9
        assert roof["frac"] < roof["frac_of_measured_peak"] < 1.05
