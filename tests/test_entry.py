"""CPU: the driver's build check must pass from the test suite too (it compiles incrementally, so this is cheap)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graft_entry_build_succeeds_without_a_gpu():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    entry = importlib.import_module("__graft_entry__")
    entry.build()  # make (no-op when up to date) + every exported symbol resolves + ABI version matches include/hfpf.h
    assert callable(entry.smoke)
