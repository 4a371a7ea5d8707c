"""bench.py's launch logic on the CPU (no GPU here: the ranks themselves must refuse to run, loudly).

* `--gpus 2` with no launcher: bench.py starts `python -m torch.distributed.run` as a child, the two ranks fail ("needs an MI355X"),
  and the parent exits NON-ZERO without printing a JSON line -- the only 8-GPU run this project gets must not measure a world of one.
* `--gpus 2` inside a launcher's world of another size: refused.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VTMC_BENCH_SELF_LAUNCHED", "TORCHELASTIC_RUN_ID"):
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
    assert p.stderr.count("bench.py needs an MI355X") >= 2                                  # ... and both ranks ran bench.py and refused
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
