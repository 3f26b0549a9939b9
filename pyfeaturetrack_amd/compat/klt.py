"""Top-level module name 'klt', as scripts written against the reference import it
(`from klt import *`).  It *is* pyfeaturetrack_amd.klt: the name is aliased, not copied, so
module-level switches such as KLT_verbose act on the real module.

Put this directory on PYTHONPATH (together with the repository root) to run such a script --
e.g. the reference's own example1.py -- unchanged on the MI355X backend.
"""
import sys
import time as _time

import pyfeaturetrack_amd.klt as _m

if not hasattr(_time, "clock"):
    # removed in Python 3.8; scripts of the reference's era time themselves with it (example1.py:51, :56) and import this module first
    _time.clock = _time.perf_counter

sys.modules[__name__] = _m
