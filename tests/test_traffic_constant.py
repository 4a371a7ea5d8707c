"""bench.py's roofline.traffic is a committed constant (PMC passes taken on the builder's lease).  It carries the SHA-256 of the kernel
sources it was measured on; bench.py reports it only while the sources still hash to that value -- a kernel change silently turns it
into `null` with the reason in traffic_source instead of a stale number."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_is_reported_only_for_the_sources_it_was_measured_on(monkeypatch):
    import bench
    from volumetricterrain_amd import build as vt_build
    j = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert len(j.get("kernel_source_sha256", "")) == 64
    real = vt_build.kernel_source_hash()
    assert len(real) == 64
    monkeypatch.setattr(vt_build, "kernel_source_hash", lambda: j["kernel_source_sha256"])
    t, src = bench.pmc_traffic("emit_kernel", True)
    assert t == j["emit_kernel_hbm_bytes"] and "committed constant" in src
    monkeypatch.setattr(vt_build, "kernel_source_hash", lambda: "0" * 64)
    t, src = bench.pmc_traffic("emit_kernel", True)
    assert t is None and src.startswith("none:")
    assert bench.pmc_traffic("emit_kernel", False) == (None, None)      # another workload: no constant applies


def test_committed_traffic_belongs_to_the_current_sources():
    """Not a parity check: a reminder.  When the extract path's kernel sources changed after the PMC passes were taken, bench.py prints
    `roofline.traffic: null` -- correct, but the round's bench line then lacks the figure.  Retake: profiles/r05/scripts/r05_pmc_retake.sh
    on the GPU box, then tools/summarize_profiles.py."""
    import pytest
    from volumetricterrain_amd import build as vt_build
    j = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    if j["kernel_source_sha256"] != vt_build.kernel_source_hash():
        pytest.skip("profiles/pmc_traffic.json was measured on other kernel sources: bench.py will report traffic = null until the PMC passes are retaken")
