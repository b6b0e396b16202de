"""Writes the INPUTS of the committed golden cases (tests/golden/*.npz, generator tests/golden/make_golden.py) as raw
little-endian Float64 files + one meta.txt per case under oracle/julia/io/<case>/, for dump_reference.jl to read
(Julia needs no package to read raw doubles).  The outputs of the Julia run go to tests/golden/reference_dump/.

    python oracle/julia/export_inputs.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG  # noqa: E402

OUT = os.path.join(ROOT, "oracle", "julia", "io")


def main():
    for c in MG.CASES:
        H, B, lam, dx, ph, law = MG.case_inputs(c)
        d = os.path.join(OUT, c)
        os.makedirs(d, exist_ok=True)
        for name, a in (("H", H), ("B", B), ("lam", lam)):
            np.asfortranarray(a, dtype="<f8").ravel(order="F").tofile(os.path.join(d, name + ".f64"))
        th = np.zeros(0) if law.theta is None else np.asarray(law.theta, dtype="<f8")
        th.tofile(os.path.join(d, "theta.f64"))
        kind = {0: "constA", 1: "nnA_scalar", 2: "nnA_gridded", 3: "nnY", 4: "nnU"}[law.kind]
        with open(os.path.join(d, "meta.txt"), "w") as f:
            f.write(f"nx {H.shape[0]}\nny {H.shape[1]}\ndx {dx!r}\ndy {dx!r}\nlaw {kind}\n")
            f.write(f"A {float(law.A) if np.ndim(law.A) == 0 else 0.0!r}\nT {float(law.T) if np.ndim(law.T) == 0 else 0.0!r}\n")
            f.write(f"rho {ph.rho!r}\ng {ph.g!r}\neta0 {ph.eta0!r}\nn {ph.n!r}\np {ph.p!r}\nq {ph.q!r}\nC {ph.C!r}\n")
            f.write(f"minA {ph.minA!r}\nmaxA {ph.maxA!r}\nP {th.size}\n")
    export_solve_case()
    export_mask_cases()
    print("inputs written to", OUT)


def _w(d, name, a):
    np.asfortranarray(a, dtype="<f8").ravel(order="F").tofile(os.path.join(d, name + ".f64"))


def export_solve_case():
    """Inputs of the whole-path dump (forward solve, loss, both gradients) = the committed golden solve case
    tests/golden/solve_valley_nnA.npz: a 48 x 40 valley, A = NN_theta(T = -2), k = 5 thickness snapshots, reltol 1e-8."""
    c = MG.solve_case()
    ph = MG.O.Phys()
    d = os.path.join(OUT, "solve_valley_nnA")
    os.makedirs(d, exist_ok=True)
    _w(d, "H0", c["H0"]); _w(d, "B", c["B"])
    np.asarray(c["ts"], dtype="<f8").tofile(os.path.join(d, "ts.f64"))
    np.asarray(c["th0"], dtype="<f8").tofile(os.path.join(d, "theta.f64"))
    for j in range(c["ref"].shape[0]):
        _w(d, f"ref_{j}", c["ref"][j])
    nx, ny = c["H0"].shape
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write(f"nx {nx}\nny {ny}\ndx 50.0\ndy 50.0\nk {len(c['ts'])}\nT -2.0\nA 2.21e-18\nC 0.0\n")
        f.write(f"rho {ph.rho!r}\ng {ph.g!r}\neta0 {ph.eta0!r}\nn {ph.n!r}\nminA {ph.minA!r}\nmaxA {ph.maxA!r}\n")
        f.write(f"P {c['th0'].size}\nreltol 1e-8\nn_quadrature {MG.SOLVE_NQ}\ndistance 3\n")


def export_mask_cases():
    """Inputs of the small out-of-tree pieces the oracle had to define itself: is_in_glacier(H, d) (Sleipnir) on two fields,
    the mass-balance mask / clip (apply_MB_mask!, Huginn) on a state + an MB field, create_interpolation's input Hbar."""
    d = os.path.join(OUT, "pieces")
    os.makedirs(d, exist_ok=True)
    p = MG.pieces_inputs()
    for k, a in p.items():
        _w(d, k, a)
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write(f"nx {p['H_a'].shape[0]}\nny {p['H_a'].shape[1]}\nnxb {p['H_b'].shape[0]}\nnyb {p['H_b'].shape[1]}\n")
        f.write(f"distance 3\nn_interp_half {MG.PIECES_NHALF}\n")


if __name__ == "__main__":
    main()
