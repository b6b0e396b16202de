"""ISA lint for a miscompile of ROCm 7.2's AMDGPU backend (`profiles/r06/sc_yt_rootcause.md`).

After a divergent `if` the join block restores EXEC with `s_or_b64 exec, exec, s[a:b]`.  Vector instructions the backend placed in
that block BEFORE the restore -- a VGPR copy of the register allocator's live-range splitting, a spill store -- run under the branch's
restricted mask and do nothing in the lanes that skipped the branch: those lanes go on with a stale register / scratch slot.
(k_rk_fused_strip<SC, YT>: `v_mov_b64 v[46:47], v[34:35]` -- the bed elevation of a strip's row 5 -- ahead of the restore behind the
table-overflow flag's branch: fuzz seed 24379.)

    python tools/exec_lint.py --lib libodinn_hip.so      every gfx950 code object embedded in a built library (llvm-objdump; seconds)
    python tools/exec_lint.py file.s [...]               listings from hipcc -S --cuda-device-only

Reports every vector instruction between the entry of a join block (the target of an `s_cbranch_execz`) and the `s_or_b64 exec` of
that block; exit status 1 if there is any."""
import re, struct, subprocess, sys, os, tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
VEC = re.compile(r"^(v_|ds_|global_|scratch_|buffer_|flat_)")
LANE = ("v_readlane", "v_writelane", "v_readfirstlane")   # lane accesses ignore EXEC
STOP = ("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_and_saveexec", "s_andn2_saveexec", "s_or_saveexec", "s_mov_b64 exec",
        "s_andn2_b64 exec", "s_and_b64 exec", "s_xor_b64 exec", "s_setpc", "s_swappc")


def demangle(names):
    try:
        return subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        return list(names)


def lint_listing(path):
    """hipcc -S listing: labels are explicit."""
    found = []
    text = open(path, errors="replace").read().split("\n")
    joins = set(re.findall(r"s_cbranch_execz (\.LBB\d+_\d+)", "\n".join(text)))
    kernel = None; pending = []; in_head = False
    for n, line in enumerate(text, 1):
        m = re.match(r"^(_Z\w+):", line)
        if m: kernel = m.group(1)
        s = line.split(";")[0].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            in_head = m.group(1) in joins; pending = []; continue
        if not in_head or not s: continue
        if re.match(r"s_or_b64 exec, exec, s\[", s):
            found += [(path, kernel, p[0], p[1]) for p in pending]
            in_head = False; pending = []
        elif VEC.match(s):
            if not s.startswith(LANE): pending.append((n, s))
        elif s.startswith(STOP):
            in_head = False; pending = []
    return found


def code_objects(lib):
    """The gfx950 code objects of every clang offload bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    while True:
        i = data.find(magic, pos)
        if i < 0: return
        o = i + len(magic)
        nb = struct.unpack_from("<Q", data, o)[0]; o += 8
        for _ in range(nb):
            off, size, tl = struct.unpack_from("<QQQ", data, o); o += 24
            triple = data[o:o + tl].decode(); o += tl
            if "gfx950" in triple and size:
                yield data[i + off:i + off + size]
        pos = i + len(magic)


def lint_library(lib):
    found = []; nk = 0
    for k, blob in enumerate(code_objects(lib)):
        with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
            f.write(blob); f.flush()
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        kernel = None; ins = []   # (address, text) of the current function
        def flush():
            nonlocal ins
            if not ins: return
            index = {a: j for j, (a, _) in enumerate(ins)}
            targets = set(); joins = set()
            for a, t in ins:
                m = re.match(r"s_c?branch\w* (-?\d+)$", t)
                if m:
                    tgt = a + 4 + 4 * int(m.group(1))
                    targets.add(tgt)
                    if t.startswith("s_cbranch_execz"): joins.add(tgt)
            for tgt in sorted(joins):
                j = index.get(tgt)
                pending = []
                while j is not None and j < len(ins):
                    a, t = ins[j]
                    if a != tgt and a in targets: break      # the next block
                    if re.match(r"s_or_b64 exec, exec, s\[", t):
                        found.extend(("%s[code object %d]" % (os.path.basename(lib), k), kernel, p[0], p[1]) for p in pending); break
                    if VEC.match(t):
                        if not t.startswith(LANE): pending.append(("0x%x" % a, t))
                    elif t.startswith(STOP): break
                    j += 1
            ins = []
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                flush(); kernel = m.group(1); nk += 1; continue
            m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+):", line)
            if m and kernel:
                ins.append((int(m.group(2), 16), re.sub(r"\s+", " ", m.group(1)).strip()))
        flush()
    return found, nk


if __name__ == "__main__":
    args = sys.argv[1:]
    found = []
    if args and args[0] == "--lib":
        for lib in args[1:]:
            f, nk = lint_library(lib)
            found += f
            print("%s: %d functions scanned" % (lib, nk))
    else:
        for p in args: found += lint_listing(p)
    for (where, _, n, t), name in zip(found, demangle([f[1] or "?" for f in found])):
        print("%s:%s: %s   in %s" % (where, n, t, re.sub(r"[(]odinn::Pools.*", "", name)[:140]))
    print("%d vector instruction(s) ahead of an EXEC restore" % len(found))
    sys.exit(1 if found else 0)
