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
    assert "0 compiler-emitted lines mention it, 0 kernels spill" in p.stdout   # density.hip: m0 belongs to the sign words' v_writelane alone


def test_m0_audit_tells_asm_from_compiler_code():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_audit
    asm = ["\t;;#ASMSTART", "\ts_mov_b32 m0, s12", "\tv_writelane_b32 v30, vcc_lo, m0", "\t;;#ASMEND"]
    bad, n = isa_audit.m0_outside_asm("\n".join(["k:"] + asm + ["\tv_add_f32_e32 v0, v0, v1 ; m0 in a comment", "\ts_endpgm"]))
    assert not bad and n == 2
    bad, n = isa_audit.m0_outside_asm("\n".join(["k:"] + asm + ["\ts_mov_b32 m0, s4", "\tds_read_b32 v1, v2", "\ts_endpgm"]))
    assert bad == ["s_mov_b32 m0, s4"] and n == 2


def test_audit_follows_control_flow_and_still_catches_a_touch():
    """The audit walks the control-flow graph, not the text: hipcc may place the block that stores the landed tile BEHIND the block that
    issues the next tile's loads.  (i) such a placement is clean; (ii) a compiler move of a register whose load is in flight is found
    on whichever path reaches it."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_audit
    wait = ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(48)", "\ts_waitcnt vmcnt(0)", "\t;;#ASMEND"]
    load = ["\t;;#ASMSTART", "\tglobal_load_dword v13, v1, s[22:23]", "\t;;#ASMEND"]
    clean = ["k:"] + load + [".LBB0_1:"] + wait + ["\ts_cbranch_execnz .LBB0_3", ".LBB0_2:"] + load + ["\ts_cbranch_scc1 .LBB0_4", "\ts_branch .LBB0_1",
             ".LBB0_3:", "\tds_write_b32 v58, v13", "\ts_branch .LBB0_2", ".LBB0_4:"] + wait + ["\ts_endpgm"]
    n, problems = isa_audit.audit_kernel("k", clean)
    assert n == 2 and not [p for p in problems if "in flight" in p], problems
    dirty = list(clean)
    dirty.insert(dirty.index("\ts_cbranch_scc1 .LBB0_4"), "\tv_mov_b32_e32 v5, v13")
    _, problems = isa_audit.audit_kernel("k", dirty)
    assert any("touches v[13]" in p for p in problems), problems
