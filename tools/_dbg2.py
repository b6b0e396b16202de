import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, math
import _odinn_import
gpu = _odinn_import.load()
import test_gpu_fuzz as F
from oracle import sia2d_oracle as O
seed = int(sys.argv[1]); g = int(sys.argv[2])
c = F._draw(gpu, 400000 + seed)
gl, law = c["gls"][g], c["laws"][g]
kind = c["kind"]
b = gpu.GlacierBatch([c["shapes"][g]], [c["dxs"][g]], [c["dys"][g]], phys=[gpu.PhysicalParameters(**c["phs"][g].__dict__)], A=[c["As"][g]], T=[c["Ts"][g]])
b.set_fields(0, gl.H0, gl.B)
if kind != O.LAW_CONST_A:
    b.set_law(kind, c["gm"], c["th"])
ph = c["phs"][g]
f = lambda H: O.sia2d_rhs(H, gl.B, gl.dx, gl.dy, ph, law, c["th"] if kind != O.LAW_CONST_A else None) if False else None
import inspect
print(inspect.signature(O.forward)); print(inspect.signature(O.sia2d_rhs))
_rms = O._rms_scaled
def _rms_tr(err, u0, u1, a, r):
    v = _rms(err, u0, u1, a, r); print("[oracle step] EEst", v); return v
O._rms_scaled = _rms_tr
snaps, so, _ = O.forward(gl, law, O.SimConfig(tstops=c["own"][g], reltol=1e-8), c["th"] if kind != O.LAW_CONST_A else None)
O._rms_scaled = _rms
os.environ["ODINN_TRACE_STEPS"]="40"; os.environ["ODINN_STEP_SC"]="0"
st = b.solve(c["own"][g], reltol=1e-8)
os.environ.pop("ODINN_TRACE_STEPS")
for j in range(len(c["own"][g])):
    d = b.snapshot(0, j)
    print("snap", j, "rel_l2 dev vs oracle", np.linalg.norm(d - snaps[j]) / max(np.linalg.norm(snaps[j]), 1e-300), "max H", snaps[j].max(), d.max(), "nan", np.isnan(d).any())
dH = b.dhdt(0, gl.H0)
print("oracle steps", so.naccept, so.nreject, "device", st[0].naccept, st[0].nreject)
# stricter oracle for comparison
for rt in (1e-8, 1e-10):
    s2, so2, _ = O.forward(gl, law, O.SimConfig(tstops=c["own"][g], reltol=rt), c["th"] if kind != O.LAW_CONST_A else None)
    print("oracle reltol", rt, so2.naccept, so2.nreject, "final vs device", np.linalg.norm(b.snapshot(0, 4) - s2[-1]) / np.linalg.norm(s2[-1]))
