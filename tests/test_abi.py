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
    for need in MUST_BIND:
        assert need in seen, f"julia/OdinnHIP.jl does not ccall {need}"
    # every other exported entry point is bound too, except the measurement / probe ones
    unbound = set(protos) - seen
    assert unbound == set(SHIM_WHITELIST), ("entry points neither bound by the shim nor whitelisted (or whitelisted but bound): "
                                            + str(sorted(unbound ^ set(SHIM_WHITELIST))))
    assert "mean_temp(" in txt and re.search(r"^mean_temp\(", txt, flags=re.M), "mean_temp must be defined in the shim"
    # the law, the mass balance and the VJP method reach the library when a Batch / HIPAdjoint is built
    ctor = re.search(r"function Batch\(simulation;.*?\nend\n", txt, flags=re.S).group(0)
    for call in ("set_law!(", "set_mass_balance!(", "set_grad_interpolation!(", "set_glacier_stops!("):
        assert call in ctor, f"Batch(simulation) does not call {call[:-1]}"
    assert ctor.index("set_law!(") < ctor.index("set_grad_interpolation!("), "odinn_set_law resets the interpolation mode: set it after"
    assert re.search(r"function HIPAdjoint\(.*?set_vjp_method!\(", txt, flags=re.S), "HIPAdjoint must hand adj.method.VJP_method to the library"


# The whole NN_theta path must be bound on the reference side (a shim that cannot carry the law runs the
# constant-A law and odinn_loss_grad rejects the 83-entry theta)
MUST_BIND = (
    "odinn_batch_create", "odinn_batch_destroy", "odinn_set_fields", "odinn_set_reference",
    "odinn_set_law", "odinn_set_theta", "odinn_set_T_field", "odinn_set_grad_interpolation",   # Laws.jl:97-183,240-273,323-386
    "odinn_set_mass_balance",                                                                     # inversion_utils.jl:498-517
    "odinn_solve", "odinn_get_snapshot",                                                          # run!(Prediction), :472-572
    "odinn_set_vjp_method",                                                                       # adj.method.VJP_method
    "odinn_sia2d_dhdt", "odinn_sia2d_vjp_H", "odinn_sia2d_vjp_theta", "odinn_mb_vjp_H",
    "odinn_surface_V", "odinn_surface_V_vjp_H", "odinn_surface_V_vjp_theta",
    "odinn_loss", "odinn_loss_grad", "odinn_loss_grad_continuous", "odinn_batch_loss_grad",
    "odinn_comm_get_unique_id", "odinn_comm_init_rank", "odinn_comm_destroy",
    "odinn_set_glacier_stops", "odinn_set_schedule", "odinn_set_A", "odinn_set_A_field",
    "odinn_get_lambda0", "odinn_get_grad_parts", "odinn_get_grad_field",
)
# exported for bench.py / probes / device-pointer callers only: a Julia host has no use for them
SHIM_WHITELIST = (
    "odinn_time_kernel", "odinn_bench_prepare", "odinn_bench_enqueue", "odinn_batch_cells",   # measurement (HIP events in the library)
    "odinn_bench_kernel_events", "odinn_bench_kernel_ms",
    "odinn_batch_sync", "odinn_device_name",                                                     # probes
    "odinn_comm_allreduce_sum_dev",                                                              # takes a DEVICE pointer + hipStream_t
)


def _c_layout():
    """{struct: (size, {field: (offset, size)})} as the C compiler lays the header's structs out (csrc/abi_layout, built by
    `make` / __graft_entry__.build())."""
    import subprocess
    exe = os.path.join(ROOT, "odinn.jl_amd", "csrc", "abi_layout")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe), "abi_layout"], check=True, capture_output=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    lay = {}
    for ln in out.splitlines():
        w = ln.split()
        if w[0] == "S":
            lay[w[1]] = (int(w[2]), {})
        else:
            lay[w[1]][1][w[2]] = (int(w[3]), int(w[4]))
    return lay


def _header_struct_fields():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "odinn_hip.h")).read(), flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\} (odinn_\w+);", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            for nm in decl.split(" ", 1)[1].split(","):
                fields.append(re.sub(r"\[.*", "", nm.strip()))
        out[name] = fields
    return out


_CT = {"odinn_phys": "Phys", "odinn_glacier_desc": "GlacierDesc", "odinn_mlp_desc": "MlpDesc", "odinn_solver_opts": "SolverOpts",
       "odinn_solve_stats": "SolveStats", "odinn_adjoint_opts": "AdjointOpts", "odinn_schedule": "Schedule"}


def test_struct_offsets_match_the_compiled_header(odinn):
    """Every field of every ABI struct: offset and size of the ctypes mirror == offsetof / sizeof of the compiled header."""
    lay = _c_layout()
    hdr = _header_struct_fields()
    assert set(lay) == set(hdr) == set(_CT), (sorted(lay), sorted(hdr))
    for cname, (size, fields) in lay.items():
        assert list(fields) == hdr[cname], f"{cname}: csrc/abi_layout.c lists {list(fields)}, the header declares {hdr[cname]}"
        ct = getattr(odinn._lib, _CT[cname])
        assert ctypes.sizeof(ct) == size, cname
        assert [f[0] for f in ct._fields_] == hdr[cname], cname
        for f, (off, sz) in fields.items():
            d = getattr(ct, f)
            assert (d.offset, d.size) == (off, sz), f"{cname}.{f}: ctypes ({d.offset}, {d.size}) vs C ({off}, {sz})"


_JL_SIZE = {"Float64": 8, "Int64": 8, "Int32": 4, "UInt8": 1}


def _jl_struct_layout(txt, name, known):
    """C layout (Julia lays isbits structs out like C) of `struct name; a::T; ... end` in the shim: [(field, offset, size)], size, align"""
    m = re.search(r"^struct %s;(.*?)end\b" % name, txt, flags=re.S | re.M)
    assert m, f"struct {name} not found in the shim"
    off, align, out = 0, 1, []
    for fld, ty in re.findall(r"(\w+)::((?:NTuple\{[^}]*\})|\w+)", m.group(1)):
        nt = re.match(r"NTuple\{\s*(\d+)\s*,\s*(\w+)\s*\}", ty)
        if nt:
            a = _JL_SIZE[nt.group(2)]
            sz = a * int(nt.group(1))
        elif ty in _JL_SIZE:
            a = sz = _JL_SIZE[ty]
        else:
            sz, a = known[ty]
        off = (off + a - 1) // a * a
        out.append((fld, off, sz))
        off += sz
        align = max(align, a)
    return out, (off + align - 1) // align * align, align


def test_julia_struct_mirrors_match_the_compiled_header():
    """The struct mirrors of julia/OdinnHIP.jl, laid out by C's rules from their field types: same field ORDER, offsets and
    sizes as the compiled header (field counts alone would miss a swapped pair or a wrong integer width)."""
    lay = _c_layout()
    txt = open(os.path.join(ROOT, "julia", "OdinnHIP.jl")).read()
    known = {}
    for jl, cname in (("Phys", "odinn_phys"), ("GlacierDesc", "odinn_glacier_desc"), ("MlpDesc", "odinn_mlp_desc"),
                      ("SolverOpts", "odinn_solver_opts"), ("SolveStats", "odinn_solve_stats"),
                      ("AdjointOpts", "odinn_adjoint_opts"), ("Schedule", "odinn_schedule")):
        fields, size, align = _jl_struct_layout(txt, jl, known)
        known[jl] = (size, align)
        csize, cfields = lay[cname]
        assert size == csize, f"{jl}: {size} bytes in the shim, {csize} in C"
        assert [(f, o, s) for f, o, s in fields] == [(f, o, s) for f, (o, s) in cfields.items()], (jl, fields, cfields)


def test_every_environment_variable_the_library_reads_is_documented_in_the_header():
    """The knob surface: at most 15 variables, each listed with its purpose in include/odinn_hip.h; the kernel-schedule overrides go
    through ONE of them (ODINN_SCHEDULE="field=value,...")."""
    import glob, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for f in glob.glob(os.path.join(root, "odinn.jl_amd", "csrc", "*")):
        if f.endswith((".hip", ".hpp", ".inc")):
            read |= set(re.findall(r'getenv\("(ODINN_[A-Z0-9_]+)"', open(f).read()))
    hdr = open(os.path.join(root, "include", "odinn_hip.h")).read()
    listed = set(re.findall(r"^ \*   (ODINN_[A-Z0-9_]+)", hdr, flags=re.M))
    assert read == listed, (sorted(read - listed), sorted(listed - read))
    assert len(read) <= 15
