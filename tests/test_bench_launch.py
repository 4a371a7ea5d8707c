"""bench.py's launch logic on the CPU (no GPU here: the ranks themselves must refuse to run, loudly).

* `--gpus 2` with no launcher: bench.py starts `python -m torch.distributed.run` as a child, the two ranks fail ("needs an MI355X"),
  and the parent exits NON-ZERO without printing a JSON line -- the only 8-GPU run this project gets must not measure a world of one.
* `--gpus 2` inside a launcher's world of another size: refused.
* round 6: supervisor / worker / watchdog / ONE conservative second attempt, driven with a STUB worker (VTMC_BENCH_STUB: the same stages,
  reports, gloo rendezvous and final reduction as a real worker, no GPU, no number -- its line says "stub": true, value 0): a rank that stops
  answering in its warm-up or timed region costs its stage's bound, every rank's watchdog fires, and the supervisors -- which never
  imported torch -- start the fallback ranks, which meet through a prefix of the launcher's store and deliver a line labelled "fallback".
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VTMC_BENCH_SELF_LAUNCHED", "TORCHELASTIC_RUN_ID", "VTMC_BENCH_ROLE",
              "VTMC_BENCH_REPORT_FD", "VTMC_BENCH_STUB", "VTMC_BENCH_FALLBACK", "TORCHELASTIC_USE_AGENT_STORE"):
        e.pop(k, None)
    return e


def test_self_launch_starts_two_ranks_and_hands_their_failure_on():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU form of this is tests/test_bench_modes.py::test_gpus_n_without_a_launcher_starts_its_own_ranks")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--grid", "256", "--steps", "1", "--warmup", "1"], cwd=ROOT, env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node 2" in p.stderr      # the child launcher was started ...
    assert p.stderr.count("bench.py needs an MI355X") == 2                                  # ... and both ranks ran bench.py and refused (exit code 4: no second attempt)
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]                   # no line for a run that did not happen


def test_a_launcher_world_of_another_size_is_refused():
    e = _clean_env()
    e.update({"RANK": "0", "WORLD_SIZE": "3", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29711"})
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--grid", "256"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "refusing" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_a_killed_parent_takes_its_launcher_and_ranks_with_it():
    """SIGTERM to `python bench.py --gpus 2` (no launcher around it) is forwarded to the child launcher's process group: nothing of the
    run is left behind holding GPUs (advisor, round 4)."""
    import signal
    import time

    import psutil
    parent = subprocess.Popen([sys.executable, "bench.py", "--gpus", "2", "--grid", "256", "--steps", "1", "--warmup", "1"], cwd=ROOT, env=_clean_env(),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    kids = []
    for _ in range(200):    # until the child launcher exists
        try:
            kids = psutil.Process(parent.pid).children(recursive=True)
        except psutil.NoSuchProcess:
            break
        if kids or parent.poll() is not None:
            break
        time.sleep(0.05)
    assert kids, "the child launcher never appeared"
    parent.send_signal(signal.SIGTERM)
    parent.wait(timeout=60)
    assert parent.returncode != 0
    gone, alive = psutil.wait_procs(kids, timeout=30)
    assert not alive, "left behind: %r" % alive


# ---------------------------------------------------------------------------------------------------------------------------------
# round 6: supervisor, watchdog, fallback (stub worker)
# ---------------------------------------------------------------------------------------------------------------------------------
def _stub(spec, extra_args=(), launcher=None, env=None, timeout=300):
    e = _clean_env()
    e["VTMC_BENCH_STUB"] = json.dumps(spec)
    e.update(env or {})
    cmd = (launcher or [sys.executable]) + ["bench.py", "--steps", "2", "--warmup", "1"] + list(extra_args)
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def _torchrun(n, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port)]


def test_supervisor_prints_the_workers_line_once_and_never_imports_torch():
    p, lines = _stub({"budget_s": 5})
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout, p.stderr[-2000:])
    j = lines[0]
    assert j["stub"] is True and j["value"] == 0.0 and j["fallback"] is False     # a stub line can never pass for a measurement
    assert j["worker"] == {"attempts": 1, "exit_code": 0, "last_stage": "teardown"}
    assert j["pipeline_depth"] == 4 and j["streams_arg"] == 2 and j["gather_stream"] == "side" and j["assign"] == "balanced" and j["place_outputs"] == 8   # the shipped defaults
    # the supervisor's own process: bench.py's module level must not pull torch in (a supervisor that touched the GPU could not start a fallback)
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.argv = ['bench.py']; import bench; print('torch' in sys.modules)"], cwd=ROOT,
                           env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert probe.stdout.strip() == "False", probe.stdout + probe.stderr


def test_a_hanging_warmup_costs_its_bound_and_the_fallback_delivers_the_line():
    import time
    t0 = time.time()
    p, lines = _stub({"hang_stage": "warmup", "hang_rank": 0, "attempts": [0], "budget_s": 2})
    assert time.time() - t0 < 60
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout, p.stderr[-2000:])
    j = lines[0]
    assert j["fallback"] is True and "watchdog in stage 'warmup'" in j["fallback_reason"] and j["worker"]["attempts"] == 2
    assert j["pipeline_depth"] == 2 and j["streams_arg"] == 1 and j["gather_stream"] == "main" and j["assign"] == "modulo" and j["place_outputs"] == 0      # the conservative configuration
    assert "WATCHDOG: stage 'warmup' exceeded its bound" in p.stderr and "bench-watchdog" not in p.stdout
    assert "Thread 0x" in p.stderr or "Current thread" in p.stderr          # faulthandler's dump of every thread's stack


def test_a_second_failure_is_final_and_no_line_is_printed():
    p, lines = _stub({"hang_stage": "timed", "hang_rank": 0, "attempts": [0, 1], "budget_s": 2})
    assert p.returncode == 3 and not lines, (p.returncode, p.stdout)
    assert p.stderr.count("WATCHDOG: stage 'timed'") == 2


def test_no_fallback_flag_and_refusals_do_not_start_a_second_attempt():
    p, lines = _stub({"hang_stage": "setup", "hang_rank": 0, "attempts": [0], "budget_s": 2}, ["--no-fallback"])
    assert p.returncode == 3 and not lines and "starting ONE fresh worker" not in p.stderr
    # a worker that refuses to run at all (no GPU here) ends with exit code 4 and is not retried
    e = _clean_env()
    q = subprocess.run([sys.executable, "bench.py", "--grid", "256", "--steps", "1", "--warmup", "1"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert q.returncode == 4 and q.stderr.count("bench.py needs an MI355X") == 1 and "starting ONE fresh worker" not in q.stderr
        assert not [ln for ln in q.stdout.splitlines() if ln.startswith("{")]


def test_a_crash_behind_the_timed_region_keeps_the_line():
    """The worker sends its line, then dies in its tear-down (round 5's aborts): the measurement is not lost, no second attempt, the
    line says how the worker ended."""
    p, lines = _stub({"crash_stage": "teardown", "crash_rank": 0, "crash_code": 134, "attempts": [0], "budget_s": 5})
    assert p.returncode == 0 and len(lines) == 1
    assert lines[0]["fallback"] is False and lines[0]["worker"]["exit_code"] == 134 and lines[0]["worker"]["attempts"] == 1
    assert "delivered its line and then ended with exit code 134" in p.stderr


def test_two_ranks_one_hangs_in_the_timed_region_both_fall_back_and_meet_again():
    """A stubbed hanging rank under the launcher the driver uses: rank 1 never enters the final reduction, rank 0 waits in it; both watchdogs fire,
    both supervisors start their fallback worker, the two meet through the `vtmc_fallback/` prefix of the agent's store and rank 0 prints ONE line."""
    p, lines = _stub({"hang_stage": "timed", "hang_rank": 1, "attempts": [0], "budget_s": 4}, ["--gpus", "2"], launcher=_torchrun(2, 29733))
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout, p.stderr[-3000:])
    j = lines[0]
    assert j["n_gpus"] == 2 and j["fallback"] is True and j["ranks_sum"] == 3 and j["worker"]["attempts"] == 2
    assert p.stderr.count("starting ONE fresh worker") == 2


def test_two_ranks_one_crashes_early_the_other_times_out_both_fall_back():
    p, lines = _stub({"crash_stage": "setup", "crash_rank": 1, "crash_code": 134, "attempts": [0], "budget_s": 4}, ["--gpus", "2"], launcher=_torchrun(2, 29735))
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout, p.stderr[-3000:])
    assert lines[0]["fallback"] is True and lines[0]["ranks_sum"] == 3
    # rank 1 crashed; rank 0 was at the barrier: its collective failed (gloo sees the closed connection) or its watchdog fired -- either way it fell back
    assert "exit code 134 in stage 'setup'" in p.stderr and ("in stage 'warmup'" in p.stderr) and p.stderr.count("starting ONE fresh worker") == 2


def test_self_launched_ranks_fall_back_too():
    """`python bench.py --gpus 2` without a launcher (--standalone rendezvous): the same second attempt."""
    p, lines = _stub({"hang_stage": "warmup", "hang_rank": 0, "attempts": [0], "budget_s": 3}, ["--gpus", "2"])
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout, p.stderr[-3000:])
    assert lines[0]["fallback"] is True and lines[0]["n_gpus"] == 2 and lines[0]["ranks_sum"] == 3
