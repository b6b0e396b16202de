"""GPU: no entry point of the C ABI may crash the host process on arguments a binding can get wrong -- "every function returns
int status ... no exceptions cross the boundary" (SURVEY 8(b)).  Two sweeps over every symbol of include/odinn_hip.h (taken
from the ctypes signature table): (1) a NULL batch handle with every other argument zero / NULL, (2) a VALID batch with every
pointer argument NULL and every count / index zero, (3) the same with every count / index -1.  Each call runs in a child process, so that a segmentation fault is a test failure
and not the end of the test session; a call must come back with a status (0 where NULL is a documented "nothing" -- e.g.
odinn_set_glacier_stops(n = 0) clears the table) and leave odinn_last_error() readable."""
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes as C, sys
sys.path.insert(0, "/root/repo") if False else None
import _odinn_import
odinn = _odinn_import.load()
L = odinn._lib
lib = L.lib()
valid = sys.argv[1] in ("valid", "negative")
ival = -1 if sys.argv[1] == "negative" else 0
import numpy as np
b = None
if valid:
    b = odinn.GlacierBatch([(24, 20), (16, 18)], [50.0, 40.0])
    for g, (nx, ny) in enumerate([(24, 20), (16, 18)]):
        H = np.asfortranarray(np.maximum(0.0, 80.0 - 0.5 * ((np.arange(nx)[:, None] - nx / 2) ** 2 + (np.arange(ny)[None, :] - ny / 2) ** 2)))
        b.set_fields(g, H, np.asfortranarray(1000.0 + 0.0 * H))
skip = {"odinn_last_error", "odinn_batch_destroy", "odinn_comm_destroy", "odinn_batch_create", "odinn_device_count", "odinn_device_name"}
done = []
for name, (res, args) in L.SIGNATURES.items():
    if name in skip or res is not C.c_int and res is not C.c_int64:
        continue
    vals = []
    for k, a in enumerate(args):
        if k == 0 and a is L._vp and valid and not name.startswith("odinn_comm"):
            vals.append(b._h)
        elif a in (C.c_int, C.c_int64, C.c_longlong):
            vals.append(ival)
        elif a is C.c_double:
            vals.append(0.0)
        else:
            vals.append(None)
    rc = getattr(lib, name)(*vals)
    msg = lib.odinn_last_error()
    assert msg is None or isinstance(msg, bytes)
    done.append((name, int(rc)))
    print(name, int(rc), flush=True)
if not valid:
    bad = [n for n, rc in done if rc == 0 and not n.startswith("odinn_comm") and n != "odinn_batch_cells"]  # (a count: 0 cells)
    assert not bad, ("accepted a NULL batch", bad)
print("SWEEP-OK", len(done))
'''


@pytest.mark.parametrize("mode", ["null", "valid", "negative"])
def test_abi_calls_with_null_and_zero_arguments_return_a_status(gpu, mode):
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", CHILD, mode], cwd=root, capture_output=True, text=True, timeout=300)
    last = [l for l in r.stdout.splitlines() if l.strip()][-1:] or [""]
    assert r.returncode == 0 and last[0].startswith("SWEEP-OK"), (mode, r.returncode, r.stdout[-600:], r.stderr[-1200:])
    assert int(last[0].split()[1]) >= 45
