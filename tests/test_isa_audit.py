"""The emit kernels issue their tile prefetch and ticket atomics from inline asm, outside the compiler's vmcnt bookkeeping
(emit_kernels.hip).  hipcc does not know that those destination registers are still in flight after ;;#ASMEND: a copy, a spill
or a reuse it inserts before the counted wait would be silent corruption (one such copy was caught by this audit while the code
was written).  The audit compiles the kernels for gfx950 and walks the ISA; it needs hipcc, not a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not installed")
def test_no_compiler_instruction_touches_a_register_with_an_asm_load_in_flight():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_audit.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-4000:] + p.stderr[-2000:]
    assert "asm loads, 0 findings" in p.stdout and "FINDING" not in p.stdout
