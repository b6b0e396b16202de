"""CPU: no kernel of the built library has a vector instruction ahead of an EXEC restore (tools/exec_lint.py).

ROCm 7.2's backend placed a live-range-split VGPR copy between the label of a join block and the `s_or_b64 exec` that ends a divergent
branch; the lanes that skipped the branch kept a stale register and k_rk_fused_strip<SC, YT> took a nondeterministic number of steps
(fuzz seed 24379; profiles/r06/sc_yt_rootcause.md).  The defect is visible in the machine code, so it is checked there: on every code
object of the in-tree libodinn_hip.so (what `make` also does before it installs the library), and the lint itself on a listing with and
without the pattern."""
import importlib.util, os, textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("exec_lint", os.path.join(ROOT, "tools", "exec_lint.py"))
exec_lint = importlib.util.module_from_spec(spec); spec.loader.exec_module(exec_lint)

BAD = textwrap.dedent("""\
    _Z6kernelv:
    \ts_and_saveexec_b64 s[0:1], vcc
    \ts_cbranch_execz .LBB0_2
    ; %bb.1:
    \tglobal_store_dword v24, v25, s[6:7]
    .LBB0_2:
    \tv_mov_b64_e32 v[46:47], v[34:35]
    \ts_mov_b64 s[52:53], s[44:45]
    \ts_or_b64 exec, exec, s[0:1]
    \tv_cvt_i32_f64_e32 v24, v[92:93]
    \ts_endpgm
    """)
GOOD = BAD.replace("\tv_mov_b64_e32 v[46:47], v[34:35]\n\ts_mov_b64 s[52:53], s[44:45]\n\ts_or_b64 exec, exec, s[0:1]\n",
                   "\ts_or_b64 exec, exec, s[0:1]\n\tv_mov_b64_e32 v[46:47], v[34:35]\n")
# a THEN block placed out of line ends with the restore as well: vector code in it is intended
THEN = textwrap.dedent("""\
    _Z6kernelv:
    \ts_and_saveexec_b64 s[4:5], s[82:83]
    \ts_cbranch_execnz .LBB0_3
    .LBB0_2:
    \ts_endpgm
    .LBB0_3:
    \tglobal_load_dwordx2 v[10:11], v10, s[76:77]
    \ts_or_b64 exec, exec, s[4:5]
    \ts_branch .LBB0_2
    """)


def test_lint_flags_the_miscompiled_join_block(tmp_path):
    for name, text, n in (("bad.s", BAD, 1), ("good.s", GOOD, 0), ("then.s", THEN, 0)):
        p = tmp_path / name
        p.write_text(text)
        found = exec_lint.lint_listing(str(p))
        assert len(found) == n, (name, found)
    assert "v_mov_b64_e32 v[46:47], v[34:35]" in exec_lint.lint_listing(str(tmp_path / "bad.s"))[0][3]


def test_built_library_has_no_vector_code_ahead_of_an_exec_restore():
    lib = os.environ.get("ODINN_LIB") or os.path.join(ROOT, "odinn.jl_amd", "csrc", "libodinn_hip.so")
    assert os.path.exists(lib), "build the library first (__graft_entry__.build())"
    found, nk = exec_lint.lint_library(lib)
    assert nk > 1000, nk  # every kernel of every translation unit was disassembled
    assert not found, found
