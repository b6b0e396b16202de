"""grad-eval/s on BASELINE configs[3]-like inputs: G alpine glaciers (sizes cycling through the
4 README stand-ins), default A(T) MLP, tspan 2 yr monthly (k=25), reltol 1e-8."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import _odinn_import
odinn = _odinn_import.load()


from _inputs import synthetic_alpine

shapes4 = [(96, 80), (128, 112), (160, 128), (192, 160)]
ph = odinn.PhysicalParameters()
for G in [int(a) for a in (sys.argv[1:] or ["4", "64", "512"])]:
    shapes = [shapes4[k % 4] for k in range(G)]
    Ts = [-9.0 + 0.5 * (k % 7) for k in range(G)]
    nn = odinn.NeuralNetwork(odinn.Parameters(), seed=42)
    mlp = odinn.MLPSpec(nn.widths, nn.acts, None, odinn.POST_AFFINE, ph.minA, ph.maxA)
    b = odinn.GlacierBatch(shapes, [50.0] * G, T=Ts)
    cache = {}
    for k, (nx, ny) in enumerate(shapes):
        if (nx, ny) not in cache:
            cache[(nx, ny)] = synthetic_alpine(nx, ny)
        b.set_fields(k, *cache[(nx, ny)])
    ts = [2010.0 + j / 12.0 for j in range(25)]
    b.set_law(odinn.LAW_NN_A_SCALAR, mlp, nn.theta)
    t0 = time.perf_counter(); st = b.solve(ts, reltol=1e-8); t_fwd0 = time.perf_counter() - t0
    for k in range(G):
        b.set_reference(k, ts, [b.snapshot(k, j) for j in range(len(ts))] if k < 4 else [b.snapshot(k % 4, j) for j in range(len(ts))], 3)
    th0 = odinn.NeuralNetwork(odinn.Parameters(), seed=1234).theta
    b.loss_grad(ts, theta=th0, reltol=1e-8)
    t0 = time.perf_counter(); n = 3
    for _ in range(n):
        L, g = b.loss_grad(ts, theta=th0, reltol=1e-8)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        b.solve(ts, reltol=1e-8)
    dtf = (time.perf_counter() - t0) / n
    steps = max(s.naccept + s.nreject for s in b.last_stats)
    # the reference's default gradient: continuous adjoint, 200 quadrature nodes, reltol = abstol = 1e-8
    b.loss_grad_continuous(ts, theta=th0, reltol=1e-8)
    t0 = time.perf_counter()
    for _ in range(n):
        Lc, gc = b.loss_grad_continuous(ts, theta=th0, reltol=1e-8)
    dtc = (time.perf_counter() - t0) / n
    rsteps = max(s.naccept + s.nreject for s in b.last_stats_rev)
    cosang = float(np.dot(g, gc) / (np.linalg.norm(g) * np.linalg.norm(gc)))
    print(json.dumps({"G": G, "cells": b.cells, "grad_eval_ms": dt * 1e3, "forward_ms": dtf * 1e3, "grad_evals_per_s": G / dt,
                      "max_steps": steps, "us_per_step": dtf * 1e6 / steps,
                      "continuous_grad_eval_ms": dtc * 1e3, "continuous_grad_evals_per_s": G / dtc,
                      "continuous_rev_steps": rsteps, "continuous_us_per_rev_step": (dtc - dtf) * 1e6 / rsteps,
                      "cos_discrete_vs_continuous": cosang, "norm_ratio": float(np.linalg.norm(gc) / np.linalg.norm(g))}))
    b.close()
