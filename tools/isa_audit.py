#!/usr/bin/env python3
"""Audit of the asynchronous tile prefetch in the compiled emit kernels (cdna_hip_programming.md, "What hipcc does not do", item 1).

The emit kernels issue their tile loads and ticket atomics from inline asm, so hipcc neither counts nor waits for them
(emit_kernels.hip: gload_async / ticket_async / wait_vm_at_most).  An asm load's destination register counts as written at
;;#ASMEND for the compiler: a copy, a spill or a reuse of it before the data has landed would be silent corruption.  This script
compiles emit_kernels.hip to gfx950 assembly and checks, for every kernel with asm loads, that

  * the destination registers of the asm loads / atomics are touched by nothing but inline asm, the ds_write of the tile into LDS,
    v_readfirstlane of the ticket and (before the first asm statement) their initialisation;
  * the kernel has no scratch and no VGPR spills;
  * inside the main loop the compiler itself emits no `s_waitcnt vmcnt` at all (the dirty list is read by scalar loads).

usage: tools/isa_audit.py [--keep file.s]      exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volumetricterrain_amd import build as vt_build  # noqa: E402

# every translation unit with kernels that issue loads from inline asm, compiled with EXACTLY the product's flags (build.FLAGS minus the
# link options): a flag the audit does not share could schedule, spill or merge differently from the library that ships
SRCS = [os.path.join(vt_build.CSRC, f) for f in ("emit_kernels.hip",)]
FLAGS = [f for f in vt_build.FLAGS if f not in ("-shared", "-fPIC")] + ["-Wno-unused-command-line-argument", "-S", "--cuda-device-only"]


def compile_asm(out, src):
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc] + FLAGS + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)


def kernels(text):
    """-> {mangled name: [lines]} for every kernel body (label .. s_endpgm)"""
    out, cur, name = {}, None, None
    for line in text.splitlines():
        m = re.match(r"^(_ZN4vtmc\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                out[name] = cur
                cur = None
    return out


def regs_of(operand_text):
    """VGPR numbers an operand string mentions (v5, v[4:7])"""
    r = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", operand_text):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", operand_text):
        r.add(int(m.group(1)))
    return r


def audit_kernel(name, lines):
    """Dataflow over the kernel's control-flow graph (hipcc places basic blocks freely: the tile's way into LDS may stand BEHIND the next
    tile's loads in the text although it runs before them, so a linear walk of the text is not the order of execution).  A register is
    `pending` from the asm statement that loads it until an asm statement that holds an `s_waitcnt vmcnt` (the counted wait at the head
    of the main loop, and once more behind it); the pending sets are propagated along every edge to a fixed point, and nothing but
    inline asm may touch a register that is pending on ANY path that reaches the instruction."""
    # units: ("asm", [lines], line no) | ("ins", code, line no) | ("label", name, line no)
    units, in_asm, block, start = [], False, [], 0
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm, block, start = True, [], i
            continue
        if "#ASMEND" in l:
            in_asm = False
            units.append(("asm", block, start))
            continue
        if in_asm:
            block.append(l)
            continue
        code = l.split(";")[0].strip()
        if not code or code.startswith("."):
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                units.append(("label", m.group(1), i))
            continue
        if code.endswith(":"):
            units.append(("label", code[:-1], i))
            continue
        units.append(("ins", code, i))
    # basic blocks
    blocks, cur = [], []
    label_of = {}
    for u in units:
        if u[0] == "label":
            if cur:
                blocks.append(cur)
                cur = []
            label_of[u[1]] = len(blocks)
            continue
        cur.append(u)
        if u[0] == "ins" and re.match(r"s_(cbranch\w*|branch|endpgm|setpc_b64)\b", u[1]):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for bi, blk in enumerate(blocks):
        last = blk[-1] if blk else None
        out = []
        if last and last[0] == "ins":
            op = last[1].split()[0]
            tgt = last[1].split()[-1]
            if op == "s_branch":
                out = [label_of[tgt]] if tgt in label_of else []
            elif op.startswith("s_cbranch"):
                out = ([label_of[tgt]] if tgt in label_of else []) + ([bi + 1] if bi + 1 < len(blocks) else [])
            elif op in ("s_endpgm", "s_setpc_b64"):
                out = []
            else:
                out = [bi + 1] if bi + 1 < len(blocks) else []
        else:
            out = [bi + 1] if bi + 1 < len(blocks) else []
        succ.append(out)

    n_loads = 0

    def transfer(blk, pending, report):
        nonlocal n_loads
        pending = set(pending)
        for kind, body, i in blk:
            if kind == "asm":
                if any("s_waitcnt vmcnt" in b for b in body):
                    pending = set()
                for b in body:
                    m = re.match(r"\s*(global_load_dword(?:x2)?|global_atomic_add)\s+(v\d+|v\[\d+:\d+\])\s*,", b)
                    if m:
                        pending |= regs_of(m.group(2))
                        if report is not None:
                            n_loads += 1
                continue
            touched = regs_of(body) & pending
            if touched and report is not None:
                report.append("%s: line %d touches v%s while its asm load is in flight: %s" % (name[:60], i, sorted(touched), body))
        return pending

    pend_in = [set() for _ in blocks]
    work = list(range(len(blocks)))
    while work:
        bi = work.pop(0)
        out = transfer(blocks[bi], pend_in[bi], None)
        for sj in succ[bi]:
            if not out <= pend_in[sj]:
                pend_in[sj] |= out
                if sj not in work:
                    work.append(sj)
    problems = []
    for bi, blk in enumerate(blocks):
        transfer(blk, pend_in[bi], problems)
    problems += sgpr_hazards(name, lines)
    if n_loads:
        problems += compiler_waits_in_loop(name, lines)
        problems += counted_stores(name, lines)
    return (n_loads or None), problems


def counted_stores(name, lines):
    """The counted wait (wait_vm_at_most(vm_issued)) is right only while vm_issued never EXCEEDS the vector-memory instructions really issued
    behind the tile loads.  The source adds one per store instruction it is sure of: a pass of stream_out_range (one 16-byte store per
    lane: global_store_dwordx4, three unrolled passes), the two 12-byte vertex stores and the one 12-byte index store of the indexed
    output (global_store_dwordx3).  A compiler that split, merged or narrowed one of them (two dwordx2 for a dwordx4, a dwordx4 for two
    dwordx3 ...) would change the instruction count under the source's feet.  Checked here: in the main loop the record stores appear in
    exactly the widths the source counts, in at least the multiplicity of one inlined copy, and nowhere as narrower pieces of them."""
    head, in_asm, start = 0, False, 0   # the main loop starts at the last counted wait of the text (as compiler_waits_in_loop)
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm, start = True, i
        elif "#ASMEND" in l:
            in_asm = False
        elif in_asm and "s_waitcnt vmcnt(48)" in l:
            head = start
    body = [l.split(";")[0].strip() for l in lines[head:]]
    n4 = sum(1 for c in body if c.startswith("global_store_dwordx4"))
    n3 = sum(1 for c in body if c.startswith("global_store_dwordx3"))
    n2 = sum(1 for c in body if c.startswith("global_store_dwordx2"))
    problems = []
    indexed = once = False
    if "emit_kernel" in name:
        # template arguments in order: FAST, INDEXED, ONCE
        args = re.findall(r"Lb([01])E", name)
        indexed = len(args) >= 2 and args[1] == "1"
        once = len(args) >= 3 and args[2] == "1"
    if indexed:
        if n3 < 3:
            problems.append("%s: %d global_store_dwordx3 (the source counts 2 vertex stores + 1 index store per batch)" % (name[:60], n3))
        if n4 or n2:
            problems.append("%s: unexpected store widths in the indexed kernel (dwordx4 %d, dwordx2 %d)" % (name[:60], n4, n2))
    else:
        # per inlined call: three passes (stream_batch76, half a batch at a time) or, in the vertex-once kernel, five (stream_batch76_full)
        ok = n4 >= 3 and (any((n4 - 5 * k) >= 0 and (n4 - 5 * k) % 3 == 0 for k in range(1, n4 // 5 + 1)) if once else n4 % 3 == 0)
        if not ok:
            problems.append("%s: %d global_store_dwordx4 (the source counts three -- vertex-once: five -- unrolled passes of stream_out_range per call)" % (name[:60], n4))
        if n3 or n2:
            problems.append("%s: unexpected store widths in the soup kernel (dwordx3 %d, dwordx2 %d)" % (name[:60], n3, n2))
    return problems


def compiler_waits_in_loop(name, lines):
    """From the counted wait at the head of the main loop (the last wait ladder in the text) to the wave's final wait, hipcc itself must
    not emit a single `s_waitcnt vmcnt`: one such wait in the loop drains the stores the asynchronous prefetch exists to leave in flight."""
    ladders, in_asm, start = [], False, 0
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm, start = True, i
        elif "#ASMEND" in l:
            in_asm = False
        elif in_asm and "s_waitcnt vmcnt(48)" in l:
            ladders.append(start)
    if not ladders:
        return ["%s: no counted wait found" % name[:60]]
    problems, in_asm = [], False
    for i in range(ladders[-1], len(lines)):
        l = lines[i]
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif not in_asm and "s_waitcnt" in l and "vmcnt" in l:
            problems.append("%s: line %d: the compiler waits on vmcnt inside the main loop: %s" % (name[:60], i, l.strip()))
    return problems


def sgpr_hazards(name, lines):
    """A VALU-written SGPR (v_readlane / v_readfirstlane / v_cmp ...) needs five wait states before a VMEM instruction reads it;
    hipcc pads its own instructions, never the inside of an asm string.  For every vector-memory instruction inside an asm
    statement: no VALU write of one of its SGPR operands within the five preceding wait states (s_nop N counts N + 1)."""
    problems = []
    code = []   # (text, in_asm)
    in_asm = False
    for l in lines:
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        c = l.split(";")[0].strip()
        if not c or c.endswith(":") or c.startswith("."):
            continue
        code.append((c, in_asm))

    def sregs(text):
        r = set()
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
            r.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bs(\d+)\b", text):
            r.add(int(m.group(1)))
        return r

    for i, (c, a) in enumerate(code):
        if not a or not re.match(r"(global_|buffer_)", c):
            continue
        need = sregs(c)
        states, j = 0, i - 1
        while j >= 0 and states < 5:
            cj = code[j][0]
            op = cj.split()[0]
            if op == "s_nop":
                states += int(cj.split()[1]) + 1
            else:
                if op.startswith("v_") and sregs(cj.split(",")[0]) & need:
                    problems.append("%s: %s reads an SGPR written %d wait states earlier by: %s" % (name[:60], c, states, cj))
                states += 1
            j -= 1
    return problems


def m0_outside_asm(text):
    """density.hip's sign words go into a register pair by v_writelane with the lane select in m0 (two SGPR operands in one VOP3 break gfx9's
    constant-bus rule), named as a clobber although hipcc calls it reserved: legal only while NOTHING the compiler emits in those kernels reads
    or writes m0.  Returns the compiler-emitted lines that mention m0, and the number of asm lines that do."""
    inside, bad, n_asm = False, [], 0
    for line in text.splitlines():
        code = line.split(";")[0] if not line.strip().startswith(";;#") else line
        if ";;#ASMSTART" in line:
            inside = True
        elif ";;#ASMEND" in line:
            inside = False
        elif re.search(r"\bm0\b", code):
            if inside:
                n_asm += 1
            else:
                bad.append(line.strip())
    return bad, n_asm


def audit_density():
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "density.s")
        compile_asm(path, os.path.join(vt_build.CSRC, "density.hip"))
        text = open(path).read()
    bad, n_asm = m0_outside_asm(text)
    spills = [m.group(0) for m in re.finditer(r"\.vgpr_spill_count:\s+(\d+)", text) if int(m.group(1))]
    return bad, n_asm, spills


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    text = ""
    with tempfile.TemporaryDirectory() as d:
        for i, src in enumerate(SRCS):
            path = (keep + (".%d" % i if i else "")) if keep else os.path.join(d, "unit%d.s" % i)
            compile_asm(path, src)
            text += open(path).read() + "\n"
    bad = []
    n_async = 0
    for name, lines in kernels(text).items():
        dst, problems = audit_kernel(name, lines)
        if dst is None:
            continue
        n_async += 1
        bad += problems
        print("%s: %d asm loads, %d findings" % (name[:70], dst, len(problems)))
    for m in re.finditer(r"\.name:\s+(\S+)|\.vgpr_spill_count:\s+(\d+)|\.private_segment_fixed_size:\s+(\d+)", text):
        if m.group(2) and int(m.group(2)):
            bad.append("VGPR spills: %s" % m.group(0))
        if m.group(3) and int(m.group(3)):
            bad.append("scratch in use: %s" % m.group(0))
    if n_async == 0:
        bad.append("no kernel with asm loads found")
    m0_lines, n_m0, dens_spills = audit_density()
    print("density.hip: %d asm lines use m0, %d compiler-emitted lines mention it, %d kernels spill" % (n_m0, len(m0_lines), len(dens_spills)))
    bad += ["density.hip: the compiler touches m0: %s" % l for l in m0_lines]
    bad += ["density.hip: VGPR spills (scratch reloads wait on vmcnt(0): behind every store of the walk): %s" % x for x in dens_spills]
    if n_m0 == 0:
        bad.append("density.hip: no asm statement with m0 found (the audit lost its subject)")
    for b in bad:
        print("FINDING:", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
