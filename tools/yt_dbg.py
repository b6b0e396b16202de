import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import _odinn_import
odinn = _odinn_import.load()
from bench import make_glacier
n, G = 512, 2
gl = [make_glacier(n, j) for j in range(G)]
b = odinn.GlacierBatch([(n, n)] * G, [100.0] * G, A=[g[2] for g in gl], T=[-5.0] * G)
for j, (H0, B, A) in enumerate(gl):
    b.set_fields(j, H0, B)
P = odinn.Parameters()
model = odinn.SIA2Dmodel(P, Y=odinn.LawY(odinn.NeuralNetwork(P, architecture=odinn.build_default_NN(2), seed=666), P))
law = model.law
b.set_law(law.kind, law.mlp, law.nn.theta, law.n_H, law.n_gradS)
print("before", b.law_table())
ts = [2010.0 + j / 12.0 for j in range(4)]
st = b.solve(ts, reltol=1e-8)
print("after", b.law_table(), [(s.naccept, s.nreject) for s in st], [g[0].max() for g in gl])
