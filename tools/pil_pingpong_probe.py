"""KLTTrackFeatures ping-pong on numpy frames and on Pillow images, alternating in one process: medians and spread per pass
(is the PIL path ever slower than the numpy path, and if so in whole passes or in single calls?)."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib.common import NFEAT, WIDTH, HEIGHT, cfg2_context, synth      # noqa: E402
from pyfeaturetrack_amd import selectGoodFeatures as sgf, trackFeatures as trk   # noqa: E402
from PIL import Image      # noqa: E402

if os.environ.get("KLT_API_FIGURES_CPUS"):
    os.sched_setaffinity(0, {int(c) for c in os.environ["KLT_API_FIGURES_CPUS"].split(",")})
sgf.KLT_verbose = trk.KLT_verbose = 0
fd = os.dup(1)
os.dup2(2, 1)
tc = cfg2_context()
a0, a1 = synth.synth_pair(WIDTH, HEIGHT, seed=1)
kinds = {"numpy": (a0, a1), "pil_owned": tuple(Image.frombytes("L", (WIDTH, HEIGHT), a.tobytes()) for a in (a0, a1)),
         "pil_mapped": (Image.fromarray(a0), Image.fromarray(a1))}
out = {k: [] for k in kinds}
for rep in range(8):
    for name, (f0, f1) in kinds.items():
        trk.KLTForgetFrames(tc)
        fl = sgf.KLTSelectGoodFeatures(tc, f0, NFEAT)
        ts = []
        for k in range(60):
            x, y = (f0, f1) if k % 2 == 0 else (f1, f0)
            t = time.perf_counter()
            trk.KLTTrackFeatures(tc, x, y, fl)
            ts.append((time.perf_counter() - t) * 1e3)
        ts = ts[4:]
        out[name].append({"median": round(statistics.median(ts), 4), "min": round(min(ts), 4), "p90": round(sorted(ts)[int(0.9 * len(ts))], 4), "max": round(max(ts), 4)})
os.write(fd, (json.dumps(out) + "\n").encode())
