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
    print("inputs written to", OUT)


if __name__ == "__main__":
    main()
