import importlib.util
import os


def load_make_golden():
    p = os.path.join(os.path.dirname(__file__), "golden", "make_golden.py")
    spec = importlib.util.spec_from_file_location("make_golden", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
