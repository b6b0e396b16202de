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
