import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def odinn():
    import _odinn_import

    return _odinn_import.load()


@pytest.fixture(scope="session")
def gpu(odinn):
    if odinn.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (the product path has no CPU fallback)")
    return odinn


def stats_err_arrays(a, b):
    """ratio, angle, relerr exactly as the reference's test/test_utils.jl:78-83."""
    a = np.asarray(a, float).ravel()
    b = np.asarray(b, float).ravel()
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    return na / nb - 1.0, float(a @ b) / (na * nb) - 1.0, np.linalg.norm(a - b) / na


def rel_l2(a, b):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    d = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (d if d > 0 else 1.0)


def sched_env(monkeypatch, **fields):
    """Set / clear fields of ODINN_SCHEDULE (the library's one schedule override variable: "field=value,...") for this test.
    sched_env(monkeypatch, ADJ_FUSED="0", ADJ_ROWS=4) merges into what is set; a value of None removes the field."""
    cur = {}
    for item in os.environ.get("ODINN_SCHEDULE", "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            cur[k.strip()] = v.strip()
    for k, v in fields.items():
        if v is None:
            cur.pop(k.lower(), None)
        else:
            cur[k.lower()] = str(v)
    if cur:
        monkeypatch.setenv("ODINN_SCHEDULE", ",".join(f"{k}={v}" for k, v in cur.items()))
    else:
        monkeypatch.delenv("ODINN_SCHEDULE", raising=False)
