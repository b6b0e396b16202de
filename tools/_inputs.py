"""Shared synthetic inputs of the tools (never import the oracle here)."""
import numpy as np


def synthetic_alpine(nx, ny, dx=50.0, hmax=110.0, slope=0.08):
    """Gentle valley glacier (same formula as the test inputs; tools never import the oracle)."""
    x = (np.arange(nx) * dx)[:, None]
    y = (np.arange(ny) * dx)[None, :]
    yc = ny * dx / 2
    B = 2200.0 - slope * x + 300.0 * ((y - yc) / yc) ** 2
    ell = ((x - 0.45 * nx * dx) / (0.38 * nx * dx)) ** 2 + ((y - yc) / (0.30 * ny * dx)) ** 2
    return np.asfortranarray(np.maximum(0.0, hmax * (1.0 - ell))), np.asfortranarray(B + 0.0 * ell)

