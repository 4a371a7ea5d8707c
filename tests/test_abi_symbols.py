"""CPU-side checks of the drop-in boundary: libvtmc.so builds for gfx950, loads, exports every
symbol include/vtmc.h declares, keeps the 76-byte wire record, and fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "vtmc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vtmc_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    import volumetricterrain_amd._lib as L
    assert sorted(L.SYMBOLS) == declared_functions()


def test_library_exports_every_declared_symbol():
    import volumetricterrain_amd as vt
    lib = vt.load()
    assert vt.library_path() and os.path.exists(vt.library_path())
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.vtmc_version().decode().startswith("vtmc ") and "gfx950" in lib.vtmc_version().decode()


def test_header_is_plain_c_and_a_c_program_links_the_library(tmp_path):
    """The boundary is a C ABI: include/vtmc.h compiles as pedantic C99 (and as C++11), and a C program linked against libvtmc.so calls
    through it -- the entry points that need no GPU: the version string, the record size, a null context answered with a status code."""
    import shutil
    import subprocess
    import volumetricterrain_amd as vt
    if not shutil.which("gcc"):
        pytest.skip("gcc not installed")
    vt.load()
    src = tmp_path / "c_host.c"
    src.write_text("""
#include "vtmc.h"
#include <stdio.h>
#include <string.h>
int main(void)
{
    vtmc_triangle t;
    memset(&t, 0, sizeof t);
    if (sizeof t != 76 || (char *)&t.block - (char *)&t != 72) return 2;      /* CSTriangle, VoxelTerrain.cs:23-37 */
    if (strncmp(vtmc_version(), "vtmc ", 5) != 0) return 3;
    if (vtmc_extract_blocks(NULL, NULL, 0, NULL) != VTMC_ERR_INVALID_ARG) return 4;   /* a status code, never a crash */
    if (vtmc_destroy(NULL) != VTMC_OK) return 5;
    puts(vtmc_version());
    return 0;
}
""")
    inc = os.path.join(ROOT, "include")
    lib = vt.library_path()
    exe = tmp_path / "c_host"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(exe), lib,
                    "-Wl,-rpath," + os.path.dirname(lib)], check=True, capture_output=True, text=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-x", "c++", "-fsyntax-only", str(src)],
                   check=True, capture_output=True, text=True)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "gfx950" in p.stdout, (p.returncode, p.stdout, p.stderr[-500:])


def test_code_object_is_gfx950_only():
    import subprocess
    import volumetricterrain_amd as vt
    vt.load()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", vt.library_path()],
                         capture_output=True, text=True).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs


def test_wire_record_is_76_bytes():
    import volumetricterrain_amd as vt
    assert vt.TRI_DTYPE.itemsize == 76            # CSTriangle.stride, VoxelTerrain.cs:36
    assert vt.TRI_DTYPE.fields["block"][1] == 72  # six float3 then the int32 block id
    text = open(os.path.join(ROOT, "include", "vtmc.h")).read()
    assert "float position0[3];" in text and "int32_t block;" in text


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product path must fail loudly, never compute on the host."""
    import torch
    import volumetricterrain_amd as vt
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(vt.VtmcError) as e:
        vt.Extractor(0)
    assert e.value.code == -4 and "no CPU path" in str(e.value)
    lib = vt.load()
    assert lib.vtmc_extract_blocks(None, None, 0, None) == -1   # null context is an error, not a crash
    assert lib.vtmc_destroy(None) == 0


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under volumetricterrain_amd/ or include/ may
    reference it (bench.py's cpu_baseline leg and __graft_entry__.smoke() are the only other users)."""
    bad = []
    for base in ("volumetricterrain_amd", "include", "host"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".c")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"\bimport oracle\b|from oracle\b|mc_oracle|libvtmc_oracle|vto_", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
