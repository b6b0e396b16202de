"""ISA-level probe of one kernel of a gfx950 listing, without recompiling it (register allocation and schedule stay what they are):
copy a 64-bit VGPR pair into AGPRs right after a given line of the kernel and store those instead of the kernel's own result.
    python tools/asm_dump.py in.s out.s KERNEL_PREFIX "LINE:vLO ..." "STORELINE:vLO ..."
LINE / STORELINE are 1-based lines relative to the kernel's label; row m's probe (m-th item) is stored by the m-th store.
Bumps .amdhsa_next_free_vgpr of the kernel to 160 (AGPRs a0..a31)."""
import sys, re
src, dst, prefix, probes, stores = sys.argv[1:6]
L = open(src, errors="replace").read().split("\n")
start = [i for i, l in enumerate(L) if l.startswith(prefix) and ":" in l][0]
ins = {}
for m, it in enumerate(probes.split()):
    ln, r = it.split(":"); ln = int(ln); r = int(r[1:])
    ins.setdefault(start + ln - 1, []).append(("after", "\ts_nop 7\n\ts_nop 7\n\tv_accvgpr_write_b32 a%d, v%d\n\tv_accvgpr_write_b32 a%d, v%d" % (2 * m, r, 2 * m + 1, r + 1)))
for m, it in enumerate(stores.split()):
    ln, r = it.split(":"); ln = int(ln); r = int(r[1:])
    ins.setdefault(start + ln - 1, []).append(("before", "\tv_accvgpr_read_b32 v%d, a%d\n\tv_accvgpr_read_b32 v%d, a%d\n\ts_nop 1" % (r, 2 * m, r + 1, 2 * m + 1)))
out = []
for i, l in enumerate(L):
    for how, txt in ins.get(i, []):
        if how == "before": out.append(txt)
    out.append(l)
    for how, txt in ins.get(i, []):
        if how == "after": out.append(txt)
# the kernel's descriptor: first .amdhsa_next_free_vgpr after an .amdhsa_kernel line naming it
k = [i for i, l in enumerate(out) if l.strip().startswith(".amdhsa_kernel " + prefix)][0]
for i in range(k, k + 80):
    if ".amdhsa_next_free_vgpr" in out[i]:
        out[i] = "\t\t.amdhsa_next_free_vgpr 160"; break
open(dst, "w").write("\n".join(out))
