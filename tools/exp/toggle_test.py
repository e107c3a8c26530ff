#!/usr/bin/env python3
"""Run one test function with module-level switches of mirror_amd.functional turned off, one at a time."""
import importlib, sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
from mirror_amd import functional as Fn
mod, fn = sys.argv[1], sys.argv[2]
m = importlib.import_module(mod)
for sw in [None] + sys.argv[3:]:
    if sw:
        setattr(Fn, sw, False)
    try:
        getattr(m, fn)()
        print(f"{sw}: PASS", flush=True)
    except AssertionError as e:
        print(f"{sw}: FAIL {str(e)[:200]!r}", flush=True)
    if sw:
        setattr(Fn, sw, True)
