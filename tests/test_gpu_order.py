"""tests/conftest.py orders the GPU suite by a hand-kept list of patterns (the driver runs `pytest -m gpu -x`: what comes after the first
failure is never reached).  A renamed test would silently fall to the end of that order; this CPU test collects the GPU suite and fails
when a pattern matches nothing or a GPU test of the core parity files matches no pattern."""
import os
import re
import subprocess
import sys

import conftest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_order_pattern_matches_and_core_tests_are_ranked():
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "--collect-only", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    ids = [l.strip() for l in p.stdout.splitlines() if "::" in l]
    assert len(ids) > 50, p.stdout[-2000:] + p.stderr[-2000:]
    for pat in conftest.GPU_ORDER:
        assert any(re.search(pat, i) for i in ids), "GPU_ORDER pattern matches no collected GPU test: " + pat
    unranked = [i for i in ids if conftest.gpu_rank(i) == len(conftest.GPU_ORDER)]
    core = [i for i in unranked if re.search(r"test_(gpu_parity|golden|lifecycle|indexed|terrain)\.py", i)]
    assert not core, "GPU tests of the core parity files without a place in conftest.GPU_ORDER: %r" % core
