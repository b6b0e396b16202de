"""4 (or argv[1]) alpine glaciers of BASELINE configs[3]: wall-clock of a discrete and a continuous gradient evaluation and the
reverse step counts (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import _odinn_import
import bench

odinn = _odinn_import.load()
Ga = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes4 = [(96, 80), (128, 112), (160, 128), (192, 160)]
shapes = [shapes4[k % 4] for k in range(Ga)]
ph = odinn.PhysicalParameters()
ts = [2010.0 + k / 12.0 for k in range(25)]
mlp = odinn.MLPSpec([1, 3, 10, 3, 1], [odinn.ACT_SOFTPLUS] * 3 + [odinn.ACT_SIGMOID], None, odinn.POST_AFFINE, ph.minA, ph.maxA)
ba = odinn.GlacierBatch(shapes, [50.0] * Ga, T=[-9.0 + 0.5 * (k % 7) for k in range(Ga)])
cache = {s: bench.alpine(*s) for s in shapes4}
for k, s in enumerate(shapes):
    ba.set_fields(k, *cache[s])
ba.set_law(odinn.LAW_NN_A_SCALAR, mlp, odinn.NeuralNetwork(odinn.Parameters(), seed=42).theta)
ba.solve(ts, reltol=1e-8)
refs = [[ba.snapshot(k, j) for j in range(len(ts))] for k in range(4)]
for k in range(Ga):
    ba.set_reference(k, ts, refs[k % 4], 3)
th0 = odinn.NeuralNetwork(odinn.Parameters(), seed=1234).theta
ba.loss_grad(ts, theta=th0, reltol=1e-8)
t0 = time.perf_counter()
for _ in range(5):
    ba.loss_grad(ts, theta=th0, reltol=1e-8)
td = (time.perf_counter() - t0) / 5
ba.loss_grad_continuous(ts, theta=th0, reltol=1e-8)
t0 = time.perf_counter()
for _ in range(5):
    ba.loss_grad_continuous(ts, theta=th0, reltol=1e-8)
tc = (time.perf_counter() - t0) / 5
sr = ba.last_stats_rev
print(f"G {Ga}: discrete {td * 1e3:.3f} ms, continuous {tc * 1e3:.3f} ms, reverse steps (max over glaciers) "
      f"{max(s.naccept + s.nreject for s in sr)}, us per reverse step {tc * 1e6 / max(s.naccept + s.nreject for s in sr):.1f}")
ba.close()
