#!/usr/bin/env python3
"""Builds the diagnostic variants of libvtmc.so next to the A/B libraries (tools/_ab/, untracked): -DVTMC_EMIT_TIMING (tools/emit_phases.py)
and -DVTMC_TIMELINE (tools/classify_timeline.py).  Run before a gpurun call that uses them: built files travel with the snapshot."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volumetricterrain_amd import build  # noqa: E402

for name, flag in (("libvtmc_phases.so", "-DVTMC_EMIT_TIMING"), ("libvtmc_timeline.so", "-DVTMC_TIMELINE")):
    print(build.build_variant(os.path.join(ROOT, "tools", "_ab", name), [flag]))
