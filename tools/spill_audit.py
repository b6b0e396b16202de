"""Per kernel of gfx950 assembly listings (hipcc -S --cuda-device-only): VGPR spills (scratch bytes) and SGPR spills into VGPR lanes
(v_writelane / v_readlane counts).  Kernels with BOTH are listed first: the combination miscompiled k_rk_fused_strip<SC, YT>.
    python tools/spill_audit.py file.s [...]"""
import re, sys, subprocess
rows = []
for path in sys.argv[1:]:
    name = None; wl = rl = 0
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):\s", line)
        if m and name is None:
            name = m.group(1); wl = rl = 0; continue
        if name:
            if "v_writelane_b32" in line: wl += 1
            elif "v_readlane_b32" in line: rl += 1
            m = re.match(r"^; ScratchSize: (\d+)", line)
            if m:
                rows.append((int(m.group(1)), wl, rl, name, path)); name = None
rows.sort(key=lambda r: (-(r[0] > 0 and r[1] > 0), -r[0], -r[1]))
names = subprocess.run(["c++filt"], input="\n".join(r[3] for r in rows), capture_output=True, text=True).stdout.splitlines()
for (sc, wl, rl, _, path), n in zip(rows, names):
    if sc or wl:
        print(f"scratch {sc:5d} B  writelane {wl:4d} readlane {rl:4d}  {re.sub(r'[(]odinn::Pools.*', '', n)[:150]}")
