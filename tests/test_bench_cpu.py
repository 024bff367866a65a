"""bench.py's host logic without a GPU: defaults of the contract, the committed PMC summaries it quotes, the watchdog."""
import json
import os
import subprocess
import sys

import pytest

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
    """roofline.traffic / mfma_busy_frac come from profiles/r06_*_pmc_summary.txt: the files must parse, and the derived
    numbers must be what DESIGN.md section 4 quotes (traffic within 1 % of the algorithmic bytes for C3 and C2)."""
    c3 = bench.pmc_summary("r06_c3_paired_pmc_summary.txt")
    gen = bench.pmc_summary("r06_c3_general_pmc_summary.txt")
    c5 = bench.pmc_summary("r06_c5_pmc_summary.txt")
    c5g = bench.pmc_summary("r06_c5_general_pmc_summary.txt")
    c2 = bench.pmc_summary("r06_c2_pmc_summary.txt")
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


def test_counters_are_quoted_only_for_the_build_they_were_taken_from(tmp_path):
    """the line's counters (traffic, mfma_busy_frac, ...) come from a committed rocprofv3 pass -- resolved by the
    key stored IN the summary (kernel build id = the tail of bf_version(), the instantiation, the launch), not by a per-round file
    name.  A summary of another build is not used: traffic null, pmc_stale true, and the line names what it left aside."""
    body = ("void dsabf::fused16_kernel<-1, 32, false, 0, true, 4, 4>\n  FETCH_SIZE                       mean 526739  (n=752)\n"
            "  WRITE_SIZE                       mean 524288  (n=731)\ndsabf::expand_kernel\n  FETCH_SIZE                       mean 1  (n=3)\n")
    key = bench.pmc_key("0123456789abcdef", "fused16_kernel<-1, 32, false, 0, true, 4, 4>", "c3", 128, True, "canonical")
    assert key == "kernels=0123456789abcdef variant=fused16_kernel<-1,32,false,0,true,4,4> workload=c3 units=128 paired=1 detect=canonical"
    (tmp_path / "r09_c3_paired_pmc_summary.txt").write_text("# pmc_key %s\n%s" % (key, body))
    (tmp_path / "r08_c3_paired_pmc_summary.txt").write_text("# pmc_key %s\n%s" % (key.replace("0123456789abcdef", "an_older_build_id"), body))
    (tmp_path / "r05_c3_paired_pmc_summary.txt").write_text(body)                       # no key: a summary from before round 6
    vals, name, stale = bench.pmc_for_launch(key, str(tmp_path))
    assert not stale and name == "r09_c3_paired_pmc_summary.txt" and vals["FETCH_SIZE"] == 526739 and vals["WRITE_SIZE"] == 524288
    assert bench.pmc_traffic(vals) == (2 * 526739 + 524288) * 1024
    # the kernels changed and nobody refreshed the profile: a doctored key = a stale summary
    vals, name, stale = bench.pmc_for_launch(key.replace("0123456789abcdef", "fedcba9876543210"), str(tmp_path))
    assert stale and vals == {} and name == "r09_c3_paired_pmc_summary.txt" and bench.pmc_traffic(vals) is None
    # another launch of the same build: nothing to fall back to by name
    vals, name, stale = bench.pmc_for_launch(key.replace("units=128", "units=32"), str(tmp_path))
    assert stale and vals == {} and name is None
    # ... and the committed summaries of THIS tree carry keys (the refresh script writes them; bf_version() reports the same id)
    from dsabeamformer_amd import build
    import dsabeamformer_amd as bfm

    kid = build.kernel_build_id()
    assert bfm.load().bf_version().decode().endswith("kernels %s)" % kid)
    # A kernel edit without a profile refresh fails HERE: the five committed counter passes of the BASELINE launches are of this build
    for name, variant in (("r06_c3_paired", "fused16_kernel<-1,32,false,0,true,4,4>"), ("r06_c3_general", "fused16_kernel<-1,32,false,0,false,4,4>"),
                          ("r06_c5", "fused16_kernel<100,32,false,0,true,4,8>"), ("r06_c5_general", "fused16_kernel<100,32,false,0,false,8,4>"),
                          ("r06_c2", "fused16_kernel<-1,2,false,0,true,4,4>")):
        first = open(os.path.join(ROOT, "profiles", name + "_pmc_summary.txt")).readline().strip()
        assert first.startswith("# pmc_key kernels=%s variant=%s " % (kid, variant)), (name, first, "run tools/refresh_profiles_r06.sh on the final kernels")
        vals, src, stale = bench.pmc_for_launch(first[len("# pmc_key "):])
        assert not stale and src == name + "_pmc_summary.txt" and vals


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

    class FakeChild:
        pid = 0

        def __init__(self, c, env=None, start_new_session=False):
            calls["cmd"], calls["env"], calls["session"] = c, env, start_new_session

        def wait(self, timeout=None):
            calls["timeout"] = timeout
            return 7

    import subprocess as sp
    monkeypatch.setattr(sp, "Popen", FakeChild)
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
    assert calls["session"] is True and calls["timeout"] > 1500      # its own session, under the parent's deadline


def _fake_topology(tmp_path, gpus, render_present):
    """A KFD topology tree like /sys/class/kfd/kfd/topology/nodes: two CPU nodes, then `gpus` GPU nodes; render_present = the
    GPU indices whose /dev/dri/renderD* exists (a container that was handed only some of the host's GPUs)."""
    nodes, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir()
    for i in range(2 + gpus):
        d = nodes / str(i)
        d.mkdir(parents=True)
        if i < 2:
            (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        else:
            (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % (128 + i - 2))
            if i - 2 in render_present:
                (dri / ("renderD%d" % (128 + i - 2))).write_text("")
    return {"DSABF_KFD_TOPOLOGY": str(nodes), "DSABF_DRI_DIR": str(dri)}


def test_gpu_count_comes_from_sysfs_not_from_the_hip_runtime(tmp_path, monkeypatch):
    """VERDICT r04 item 1d / ADVICE: the parent of the ranks counts devices without loading HIP -- KFD topology nodes with
    SIMDs whose render node this container can open, cut to HIP_VISIBLE_DEVICES."""
    for k, v in _fake_topology(tmp_path, 8, {0, 1, 2, 3, 4, 5, 6, 7}).items():
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert bench.count_gpus() == 8
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert bench.count_gpus() == 2
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    (tmp_path / "dri" / "renderD130").unlink()            # a GPU of the host that was not handed to this container
    assert bench.count_gpus() == 7
    monkeypatch.setenv("DSABF_KFD_TOPOLOGY", str(tmp_path / "nothing"))
    assert bench.count_gpus() == 0
    import inspect
    assert "import torch" not in inspect.getsource(bench.self_launch) and "import torch" not in inspect.getsource(bench.count_gpus)


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DSABF_BENCH_ONE_GPU", "HIP_VISIBLE_DEVICES",
                                                              "ROCR_VISIBLE_DEVICES", "DSABF_BENCH_CHILD_COUNT")}


@pytest.mark.parametrize("child", ["1", "none"])
def test_multi_gpu_request_on_a_box_with_fewer_gpus_says_so(tmp_path, child):
    """`python bench.py --gpus 2` on a box that shows one GPU (an 8-GPU host, one render node in the container) -- and a child
    process's torch.cuda.device_count() agrees, or does not answer: refused with ONE JSON line on stdout like every other outcome
    (value null, the error, BOTH counts), a non-zero status, no launcher -- whatever hardware the test itself runs on (the
    topology is a fixture, the child's answer too: DSABF_BENCH_CHILD_COUNT)."""
    env = _clean_env()
    env.update(_fake_topology(tmp_path, 8, {3}), DSABF_BENCH_CHILD_COUNT=child)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 6 and "this node shows 1 GPU(s)" in r.stderr
    assert "starting" not in r.stderr                                   # no launcher was started
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "this node shows 1 GPU(s)" in d["error"] and d["rank"] is None
    assert d["gpus_seen"] == {"sysfs": 1, "child_device_count": 1 if child == "1" else None}


def test_a_container_without_the_sysfs_tree_goes_by_the_childs_count(tmp_path):
    """/dev/kfd without /sys/class/kfd/kfd/topology (or render nodes under another permission model) made the
    sysfs walk see 0 GPUs and the run was lost to a hard refusal.  Now a short-lived child's torch.cuda.device_count() is the
    second opinion: if IT sees enough GPUs the launcher starts (here a stand-in that answers at once)."""
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "bench.launcher_command = lambda n, argv, port: [sys.executable, '-c', 'print(\"LAUNCHED %%d\" %% ' + str(n) + ')']\n"
            "sys.argv = ['bench.py', '--gpus', '4']\n"
            "bench.main()\n" % ROOT)
    env = _clean_env()
    env.update(DSABF_KFD_TOPOLOGY=str(tmp_path / "no_such_tree"), DSABF_DRI_DIR=str(tmp_path), DSABF_BENCH_CHILD_COUNT="8")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "LAUNCHED 4" in r.stdout and "going by the child" in r.stderr and "sysfs shows 0 GPU(s)" in r.stderr
    import inspect
    import re
    assert not re.search(r"^\s*(import|from) torch", inspect.getsource(bench.count_gpus_in_child), re.M)   # the PARENT never imports torch


def test_a_launcher_that_never_returns_is_killed_and_reported(tmp_path):
    """The parent's own deadline: the child (its own session) is killed as a process group, one diagnostic line, status 5."""
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "bench.launcher_command = lambda n, argv, port: [sys.executable, '-c', 'import time; time.sleep(600)']\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--deadline-seconds', '1']\n"
            "bench.main()\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(DSABF_BENCH_ONE_GPU="1", DSABF_LAUNCH_GRACE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 5, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] is None and d["n_gpus"] == 2 and "did not return" in d["error"]


def test_deadline_before_the_headline_is_a_failure_with_a_diagnostic_line():
    """armed before anything can block; when a stage never finishes rank 0 prints ONE line naming the
    stage and the process ends NON-ZERO (os._exit from the timer thread: the main thread is stuck by definition)."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "d = bench.Deadline(rank=0, n_gpus=8, total_seconds=60.0)\n"
            "d.stage('control plane: init_process_group(gloo)', 0.3)\n"
            "time.sleep(30)\n"
            "print('never reached')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "never reached" not in r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 8 and d["rank"] == 0 and d["stage"].startswith("control plane")
    assert "did not finish" in d["error"] and d["metric"].startswith("beam-blocks/sec")
    # ranks other than 0 print no line (the launcher's stdout carries rank 0's only); they say why on stderr and end non-zero
    r2 = subprocess.run([sys.executable, "-c", code.replace("rank=0", "rank=3")], capture_output=True, text=True, timeout=60)
    assert r2.returncode == 3 and not [l for l in r2.stdout.splitlines() if l.startswith("{")] and "rank 3" in r2.stderr


def test_deadline_after_the_kernel_only_record_keeps_it_and_names_the_gather():
    """kernel-only scaling is recorded before any RCCL traffic; a gather stage that hangs leaves a line
    with gather_modes.none, gather_error and value null; the exit status is the gather's failure code."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "d = bench.Deadline(rank=0, n_gpus=2, total_seconds=60.0)\n"
            "out = {'metric': 'm', 'value': 123.0, 'n_gpus': 2, 'gather_modes': {'none': {'value': 123.0}}, 'roofline': {'frac': 0.4}}\n"
            "d.adopt(out); d.fail_code = 4\n"
            "d.stage(\"gather 'alltoall_rank_major': warm-up\", 0.3)\n"
            "time.sleep(30)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 4
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] is None and d["gather_modes"]["none"]["value"] == 123.0 and "alltoall_rank_major" in d["gather_error"]
    assert d["stage"].endswith("warm-up") and d["roofline"]["frac"] == 0.4


def test_deadline_reports_a_sigterm_from_the_launcher():
    """The launcher stops the other ranks with SIGTERM when one fails: rank 0 must still say what it had -- even while its main
    thread sits in a call that never returns (sigwait on a thread of its own, not a Python-level handler)."""
    code = ("import os, sys, time, signal; sys.path.insert(0, %r); import bench\n"
            "signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})\n"
            "d = bench.Deadline(rank=0, n_gpus=2, total_seconds=60.0)\n"
            "bench.watch_sigterm(d)\n"
            "d.stage('RCCL communicator (bf_comm_create)', 50.0)\n"
            "print('ready', flush=True)\n"
            "time.sleep(30)\n" % ROOT)
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "ready"
    p.terminate()
    out, _ = p.communicate(timeout=30)
    assert p.returncode == 143
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert "signal 15" in d["error"] and d["stage"].startswith("RCCL communicator")


def test_issue_model_says_what_binds_the_kernel():
    """from the committed PMC pass of the headline launch -- 17.4 VALU ops per MFMA (17.2 with round 5's
    compile-time 64-antenna class: the run-time class tests its staging pieces against the antenna count), 13 + 2.45 K cycles of
    issue per MFMA account for the launch's cycles (the SIMDs' instruction issue is the bound, the matrix pipe is 28 % busy),
    and the clock the chip held."""
    c3 = bench.pmc_summary("r06_c3_paired_pmc_summary.txt")
    m = bench.issue_model(c3, 0.9475)
    assert abs(m["valu_per_mfma"] - 17.4) < 0.15 and abs(m["issue_model_cycles_per_mfma"] - (13 + 2.45 * m["valu_per_mfma"])) < 1e-9
    assert 0.9 < m["issue_occupancy"] < 1.05 and m["bound_measured"] == "simd-issue"
    assert 1.9 < m["clock_ghz_under_load"] < 2.45      # (2.04 ... 2.20 by box; nominal 2.4)
    gen = bench.issue_model(bench.pmc_summary("r06_c3_general_pmc_summary.txt"), 1.05)
    assert 6.0 < gen["valu_per_mfma"] < 7.5 and gen["bound_measured"] in ("simd-issue", "unclear")
    assert bench.issue_model({}, 1.0) == {"bound_measured": None}


def test_committed_bench_lines_agree_with_the_committed_rocprof_kernel_stats():
    """profiles/: the HIP-event average of the dominant kernel in each committed bench line must agree with the AverageNs of
    the same kernel in the rocprofv3 --kernel-trace --stats summary committed beside it (same command, another run and,
    for rocprof, a profiled one: 5 %; 8 % for the 50-us launches of C2), and roofline.frac must follow from it."""
    import csv

    for wl, stats, units_blocks in (("c3", "r06_c3_paired_kernel_stats.csv", 2048), ("c5", "r06_c5_kernel_stats.csv", 128),
                                    ("c2", "r06_c2_kernel_stats.csv", 1024)):
        line = [l for l in open(os.path.join(ROOT, "profiles", "r06_%s_bench.json" % wl)) if l.startswith("{")][-1]
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
        assert roof["traffic"] is not None and roof["pmc_source"].startswith("profiles/r06_") and roof["pmc_stale"] is False
        assert open(os.path.join(ROOT, roof["pmc_source"])).readline().strip() == "# pmc_key " + roof["pmc_key"]
        # labelled: not observed in that run (round 5 adds what binds the kernel, from the same committed passes)
        assert {"traffic", "mfma_busy_frac"} <= set(roof["from_committed_profile"]) <= {"traffic", "mfma_busy_frac", "valu_per_mfma",
                                                                                        "issue_occupancy", "bound_measured", "clock_ghz_under_load"}
        assert d["ms_per_step"] >= roof["kernel_ms_avg"] * 0.999          # the whole step cannot be shorter than its kernel
        # the measured peak beside the nominal one, from the same run (a pure MFMA loop / a pure streaming kernel)
        assert 0.6 * roof["peak"] < roof["peak_measured"] <= roof["peak"] * 1.02
        assert abs(roof["achieved"] / roof["peak_measured"] / roof["frac_of_measured_peak"] - 1) < 1e-9
        assert roof["frac"] < roof["frac_of_measured_peak"] < 1.05


def test_a_signal_to_the_parent_ends_the_ranks_too(tmp_path):
    """The launcher and its ranks run in a session of their own (so that the parent's deadline can kill them as a group): a
    SIGTERM to the parent must therefore be passed on, or the ranks would outlive it and keep the GPUs."""
    import signal
    import time

    marker = tmp_path / "child.pid"
    child_code = "import os, time; open(%r, 'w').write(str(os.getpid())); time.sleep(600)" % str(marker)
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "bench.launcher_command = lambda n, argv, port: [sys.executable, '-c', %r]\n"
            "sys.argv = ['bench.py', '--gpus', '2']\n"
            "bench.main()\n" % (ROOT, child_code))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(DSABF_BENCH_ONE_GPU="1")
    p = subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not marker.exists() and time.time() - t0 < 60:
        time.sleep(0.05)
    assert marker.exists()
    time.sleep(0.2)
    pid = int(marker.read_text())
    p.send_signal(signal.SIGTERM)
    p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM
    for _ in range(100):                       # the child is gone (ESRCH), not orphaned
        try:
            os.kill(pid, 0)
        except OSError:
            break
        time.sleep(0.05)
    else:
        os.kill(pid, signal.SIGKILL)
        raise AssertionError("the launcher's child outlived its parent")


def test_stdout_is_reserved_for_the_line():
    """bench.claim_stdout(): after it, what python prints and what native code writes to fd 1 (gloo's connection notes at N > 1)
    lands on stderr; the line -- print_line -- is the only thing on the real stdout.  A closed stderr does not break it."""
    import subprocess

    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); print('python noise'); "
            "os.write(1, b'native noise\\n'); bench.print_line('{\"ok\": 1}')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == '{"ok": 1}\n' and "native noise" in r.stderr and "python noise" in r.stderr
    code = ("import os, sys; sys.path.insert(0, %r); import bench; os.close(2); bench.claim_stdout(); bench.print_line('{\"ok\": 1}')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == '{"ok": 1}\n'
