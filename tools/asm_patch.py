"""Edit ONE kernel of a gfx950 assembly listing: python tools/asm_patch.py in.s out.s KERNEL_PREFIX MODE
MODE: dpp (s_nop 7 before every DPP move), lane (s_nop 7 around v_readlane / v_writelane), scratch (s_waitcnt vmcnt(0) + s_nop around
every scratch access), all.  Used to test hazard hypotheses on a miscompiled kernel without recompiling it."""
import sys
src, dst, prefix, mode = sys.argv[1:5]
out = []; inside = False; n = 0
for line in open(src, errors="replace"):
    if line.startswith(prefix) and line.rstrip().split(":")[0].startswith(prefix) and ":" in line:
        inside = True
    if inside and line.startswith(".Lfunc_end"):
        inside = False
    if inside:
        s = line.strip()
        if mode in ("dpp", "all") and "_dpp" in s.split(" ")[0]:
            out.append("\ts_nop 7\n"); n += 1
        if mode in ("lane", "all") and (s.startswith("v_readlane") or s.startswith("v_writelane")):
            out.append("\ts_nop 7\n"); out.append(line); out.append("\ts_nop 7\n"); n += 1; continue
        if mode in ("scratch", "all") and s.startswith("scratch_"):
            out.append("\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n"); out.append(line); out.append("\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n"); n += 1; continue
    out.append(line)
open(dst, "w").writelines(out)
print(mode, "patched", n, "sites")
