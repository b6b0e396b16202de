"""-m "not gpu": the C-ABI library loads without a GPU and exports every symbol that
include/odinn_hip.h declares; no compute is called.  Compute entry points must fail loudly
(no CPU fallback) when no HIP device is visible."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "odinn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(odinn_[A-Za-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(odinn):
    lib = odinn._lib.lib()
    names = _header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/odinn_hip.h but not exported"
    assert sorted(odinn._lib.SIGNATURES) == names, "ctypes signature table out of sync with the header"


def test_struct_layouts_match_the_header(odinn):
    L = odinn._lib
    assert ctypes.sizeof(L.Phys) == 9 * 8
    assert ctypes.sizeof(L.GlacierDesc) == 8 + 16 + 72 + 16
    assert ctypes.sizeof(L.MlpDesc) == 4 * (1 + 9 + 8 + 1) + 4 + 32 + 8 + 16
    assert L.MlpDesc.pre_lo.offset == 80 and L.MlpDesc.post_lo.offset == 120
    assert ctypes.sizeof(L.SolverOpts) == 64 and L.SolverOpts.cfl.offset == 56 and ctypes.sizeof(L.SolveStats) == 40
    assert ctypes.sizeof(L.AdjointOpts) == 40 and L.AdjointOpts.maxiters.offset == 32
    assert ctypes.sizeof(L.Schedule) == 80 and L.Schedule.adj_theta_fused.offset == 56
    hdr = open(os.path.join(ROOT, "include", "odinn_hip.h")).read()
    body = re.search(r"typedef struct odinn_schedule \{(.*?)\} odinn_schedule;", hdr, flags=re.S).group(1)
    fields = re.findall(r"int32_t\s+([a-z_]+)\s*;", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert tuple(fields) == L.SCHEDULE_FIELDS, "odinn_schedule fields out of sync with the ctypes mirror"


def test_no_silent_cpu_fallback(odinn):
    if odinn.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(odinn.OdinnError, match="no CPU fallback"):
        odinn.GlacierBatch([(16, 16)], [50.0])


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "odinn.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} references the oracle"


def test_julia_shim_binds_only_declared_symbols():
    """julia/OdinnHIP.jl (the reference-side ccall binding, not executable here) and INTEGRATION.md
    refer only to entry points that include/odinn_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "odinn_hip.h")).read()
    declared = set(re.findall(r"\b(odinn_[A-Za-z0-9_]+)\s*\(", hdr))
    for rel in ("julia/OdinnHIP.jl", "INTEGRATION.md"):
        txt = open(os.path.join(ROOT, rel)).read()
        used = set(re.findall(r"\(:(odinn_[A-Za-z0-9_]+),\s*lib\)", txt))
        assert used, rel
        assert used <= declared, (rel, sorted(used - declared))


def _split_top(s):
    """split on commas that are not nested inside (), {} or []"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _header_prototypes():
    src = open(os.path.join(ROOT, "include", "odinn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for ret, name, args in re.findall(r"\b(int64_t|int|const char\*)\s+(odinn_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        a = [x for x in _split_top(" ".join(args.split())) if x and x != "void"]
        protos[name] = (ret, a)
    return protos


def _c_class(arg):
    """pointer / int / double class of one C parameter"""
    if "*" in arg:
        return "ptr"
    if re.match(r"(const\s+)?double\b", arg):
        return "double"
    return "int"


def _jl_class(t):
    t = t.strip()
    if t.startswith("Ptr{") or t in ("Cstring",):
        return "ptr"
    if t in ("Cdouble", "Float64"):
        return "double"
    assert t in ("Cint", "Int32", "Int64", "Clonglong"), f"unexpected Julia argument type {t}"
    return "int"


def test_julia_shim_ccalls_match_the_header_prototypes():
    """Every ccall of julia/OdinnHIP.jl: return type, argument COUNT and the pointer / integer / double class of each
    argument against the prototype in include/odinn_hip.h (the shim cannot be executed here: no Julia in the image)."""
    protos = _header_prototypes()
    assert len(protos) >= 40
    txt = open(os.path.join(ROOT, "julia", "OdinnHIP.jl")).read()
    calls = re.findall(r"ccall\(\(:(odinn_[A-Za-z0-9_]+),\s*lib\),\s*([A-Za-z0-9_{}]+),\s*\(([^)]*(?:\{[^}]*\}[^)]*)*)\)", txt, flags=re.S)
    assert len(calls) >= 12
    seen = set()
    for name, ret, types in calls:
        assert name in protos, name
        cret, cargs = protos[name]
        assert {"int": "Cint", "int64_t": "Int64", "const char*": "Cstring"}[cret] == ret, (name, ret)
        jt = [x for x in _split_top(" ".join(types.split())) if x]
        assert len(jt) == len(cargs), f"{name}: ccall passes {len(jt)} arguments, the header declares {len(cargs)}"
        for k, (a, b) in enumerate(zip(jt, cargs)):
            assert _jl_class(a) == _c_class(b), f"{name}: argument {k} is {a} in the shim, `{b}` in the header"
        seen.add(name)
    # the seams a maintainer needs are all bound
    for need in ("odinn_batch_create", "odinn_sia2d_dhdt", "odinn_sia2d_vjp_H", "odinn_sia2d_vjp_theta", "odinn_loss_grad",
                 "odinn_loss_grad_continuous", "odinn_batch_loss_grad", "odinn_comm_init_rank", "odinn_comm_get_unique_id"):
        assert need in seen, need
    # struct mirrors: same field count as the C structs
    for jl, cname, n in (("SolverOpts", "odinn_solver_opts", 9), ("AdjointOpts", "odinn_adjoint_opts", 6), ("Phys", "odinn_phys", 9),
                         ("Schedule", "odinn_schedule", 16)):
        m = re.search(r"struct %s;(.*?)end" % jl, txt, flags=re.S)
        assert m and len(re.findall(r"::", m.group(1))) == n, jl
    assert "mean_temp(" in txt and re.search(r"^mean_temp\(", txt, flags=re.M), "mean_temp must be defined in the shim"
