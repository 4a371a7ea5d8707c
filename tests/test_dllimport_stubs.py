"""INTEGRATION.md holds the only artefact of SURVEY.md section 8f-3 (the C# host side): the [DllImport] stubs a maintainer of the
reference would add to VoxelTerrain.cs.  No C# toolchain exists in the image, so nothing compiles them -- this test keeps them honest
instead: every stub is parsed out of the markdown and checked against the prototype of the same name in include/vtmc.h (it must
exist; same number of parameters; every parameter and the return value of a blittable type that marshals to the C type)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_prototypes():
    text = open(os.path.join(ROOT, "include", "vtmc.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int32_t|const char \*|void)\s*(vtmc_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, params)
    return protos


def cs_stubs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stubs = {}
    for m in re.finditer(r"\[DllImport\([^\]]*\)\]\s*public static extern\s+(\w+)\s+(vtmc_\w+)\s*\((.*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [a.strip() for a in re.split(r",(?![^\[]*\])", args)] if args else []   # commas inside float[,,] do not split
        stubs[name] = (ret, params)
    return stubs


def c_kind(decl):
    """('ptr', base) | ('val', base) of a C parameter declaration"""
    d = re.sub(r"\b(const|struct)\b", "", decl).strip()
    d = re.sub(r"\[[^\]]*\]", "*", d)           # uint8_t id[128] is a pointer parameter
    stars = d.count("*")
    base = d.replace("*", " ").split()[0]
    return ("ptr" if stars else "val", base, stars)


CS_VALUE = {"int": "int32_t", "uint": "uint32_t", "long": "int64_t", "ulong": "uint64_t", "float": "float"}
CS_ARRAY = {"float": {"float", "void"}, "int": {"int32_t", "void"}, "uint": {"uint32_t", "void"}, "byte": {"uint8_t", "void"},
            "VtmcTriangle": {"vtmc_triangle"}, "VtmcModifier": {"vtmc_modifier"}, "VtmcVertex": {"vtmc_vertex"}}
CS_STRUCT = {"VtmcVolumeBatch": "vtmc_volume_batch", "VtmcChunkView": "vtmc_chunk_view", "VtmcDensityParams": "vtmc_density_params"}


def matches(cs, c):
    kind, base, stars = c_kind(c)
    cs = re.sub(r"\[(In|Out|In, Out)\]\s*", "", cs).strip()
    words = cs.split()
    mod = words[0] if words[0] in ("out", "ref", "in") else None
    typ = words[1] if mod else words[0]
    if typ == "string":
        return kind == "ptr" and base == "char" and not mod
    if typ == "IntPtr":
        return kind == "ptr" and (stars == 2 if mod else stars >= 1)       # out IntPtr <-> T **
    m = re.match(r"(\w+)\[[,]*\]$", typ)
    if m:                                                                     # blittable array: pinned, passed as a pointer to its first element
        return kind == "ptr" and stars == 1 and base in CS_ARRAY.get(m.group(1), set()) and not mod
    if typ in CS_STRUCT:
        return kind == "ptr" and stars == 1 and base == CS_STRUCT[typ] and mod in ("ref", "out", "in")
    if typ in CS_VALUE:
        if mod:                                                               # out int <-> int32_t *
            return kind == "ptr" and stars == 1 and base == CS_VALUE[typ]
        return kind == "val" and base == CS_VALUE[typ]
    return False


def test_every_dllimport_stub_matches_its_c_prototype():
    protos, stubs = c_prototypes(), cs_stubs()
    assert len(stubs) >= 15, "INTEGRATION.md lost its [DllImport] stubs"
    problems = []
    for name, (ret, params) in stubs.items():
        if name not in protos:
            problems.append("%s: not declared in include/vtmc.h" % name)
            continue
        c_ret, c_params = protos[name]
        if not ((ret == "int" and c_ret == "int32_t") or (ret == "IntPtr" and c_ret == "const char *") or (ret == "void" and c_ret == "void")):
            problems.append("%s: returns %s, the C function returns %s" % (name, ret, c_ret))
        if len(params) != len(c_params):
            problems.append("%s: %d parameters, the C function has %d (%s)" % (name, len(params), len(c_params), ", ".join(c_params)))
            continue
        for cs, c in zip(params, c_params):
            if not matches(cs, c):
                problems.append("%s: '%s' does not marshal to '%s'" % (name, cs, c))
    assert not problems, "\n".join(problems)


def test_the_core_sequence_of_batchupdate_is_covered():
    """VoxelTerrain.cs:151-156 / :228-244 / :365-427: create, destroy, extract, read back -- the calls BatchUpdate shrinks to."""
    stubs = cs_stubs()
    for name in ("vtmc_create", "vtmc_destroy", "vtmc_last_error", "vtmc_extract_blocks", "vtmc_extract_grid", "vtmc_read_triangles"):
        assert name in stubs, name
