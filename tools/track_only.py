"""rocprofv3 target: the tracker alone on a long feature list (default 20000), a few launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.backend import Context
from pyfeaturetrack_amd.klt import KLT_TrackingContext
from pyfeaturetrack_amd.params import params_from_tc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tc = KLT_TrackingContext(); tc.nPyramidLevels, tc.subsampling = 3, 4; tc.KLTUpdateTCBorder()
cx = Context(0); cx.set_params(params_from_tc(tc))
f0, f1 = synth.synth_pair(1920, 1080, seed=1)
cx.upload(0, f0); cx.upload(1, f1); cx.build_pyramids(0); cx.build_pyramids(1)
fl, _ = cx.select(0, 5000, use_pyramid=True)
big = np.tile(fl, (n + 4999) // 5000)[:n].copy()
cx.featbuf_upload(0, big); cx.featbuf_upload(1, big)
for _ in range(12): cx.track_async(0, 1, 0, 1, n)
cx.sync(); cx.close()
