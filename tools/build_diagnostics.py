#!/usr/bin/env python3
"""Builds the diagnostic variants of libvtmc.so next to the A/B libraries (tools/_ab/, untracked): -DVTMC_DIAGNOSTICS (the *_ablate tuning
keys, which the product library refuses: VTMC_LIB=tools/_ab/libvtmc_diag.so), -DVTMC_EMIT_TIMING (tools/emit_phases.py) and -DVTMC_TIMELINE
(tools/classify_timeline.py).  Run before a gpurun call that uses them: built files travel with the snapshot."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volumetricterrain_amd import build  # noqa: E402

for name, flags in (("libvtmc_diag.so", ["-DVTMC_DIAGNOSTICS"]), ("libvtmc_phases.so", ["-DVTMC_DIAGNOSTICS", "-DVTMC_EMIT_TIMING"]),
                    ("libvtmc_timeline.so", ["-DVTMC_DIAGNOSTICS", "-DVTMC_TIMELINE"])):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    print(build.build_variant(os.path.join(ROOT, "tools", "_ab", name), flags))
