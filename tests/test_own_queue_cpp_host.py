"""The own-queue stream of a context (vtmc_context_stream(ctx, 1, &s)) from a plain C++ / HIP host on ROCm's own runtime -- the runtime
under which round 5 saw a C++ host hang at process exit when such a stream carried pinned staging.  tools/calib/own_queue_host.hip drives two
contexts the documented way (sampler + extract queued on the own-queue streams, two steps in flight, the library's staging on its ordinary
stream), compares the triangle totals with the same work on ordinary streams, destroys the contexts and EXITS: the test fails on a wrong
total, on a non-zero exit code and on a process that does not come back."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.gpu
def test_cpp_host_on_own_queue_streams_runs_and_exits(tmp_path):
    import volumetricterrain_amd as vt
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    vt.load()
    libdir = os.path.dirname(vt.library_path())
    exe = str(tmp_path / "own_queue_host")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tools", "calib", "own_queue_host.hip"), "-L", libdir, "-lvtmc", "-Wl,-rpath," + libdir],
                   check=True, capture_output=True, text=True, timeout=600)
    try:
        p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired as e:
        pytest.fail("the C++ host did not exit within 120 s (output so far: %r)" % ((e.stdout or b"")[-400:],))
    assert p.returncode == 0 and "OWN-QUEUE-HOST-OK" in p.stdout, (p.returncode, p.stdout[-600:], p.stderr[-600:])
