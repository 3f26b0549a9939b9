"""Top-level module name 'goodFeaturesUtils', as scripts written against the reference import it
(`from goodFeaturesUtils import *`).  It *is* pyfeaturetrack_amd.goodFeaturesUtils: the name is aliased, not copied, so
module-level switches such as KLT_verbose act on the real module.

Put this directory on PYTHONPATH (together with the repository root) to run such a script --
e.g. the reference's own example1.py -- unchanged on the MI355X backend.
"""
import sys

import pyfeaturetrack_amd.goodFeaturesUtils as _m

sys.modules[__name__] = _m
