"""Time line of the tracker's wavefronts at cfg-2 (first 256 features): builds a private copy of the library with
KLT_TRACK_CLOCKS (every mark waits for outstanding memory operations first) and prints ticks of 10 ns between the marks.
Run on the GPU box:  python tools/track_clocks.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "pyfeaturetrack_amd", "csrc")
dbg = os.path.join(ROOT, "gpurun_out", "libkltgpu_clk.so")
os.makedirs(os.path.dirname(dbg), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                "-I" + os.path.join(ROOT, "include"), "-Wno-cuda-compat", "-DKLT_TRACK_CLOCKS", "-c", os.path.join(src, "track_kernels.hip"),
                "-o", "/tmp/track_clk.o"], check=True)
objs = [os.path.join(src, f) for f in ("api_context.o", "api_frames.o", "api_featbuf.o", "api_select.o", "api_track.o", "api_comm.o", "api_compat.o", "host_pool.o", "comm.o", "conv_kernels.o", "pyramid_kernels.o", "select_kernels.o", "sat_pipeline.o", "affine_kernels.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", dbg, "/tmp/track_clk.o"] + objs + ["-ldl", "-lpthread"], check=True)
os.environ["KLT_GPU_LIB"] = dbg

import numpy as np                                          # noqa: E402
from pyfeaturetrack_amd import synth                       # noqa: E402
from pyfeaturetrack_amd.backend import Context              # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext      # noqa: E402
from pyfeaturetrack_amd.params import params_from_tc        # noqa: E402

tc = KLT_TrackingContext(); tc.nPyramidLevels, tc.subsampling = 3, 4; tc.KLTUpdateTCBorder()
ctx = Context(0); ctx.set_params(params_from_tc(tc))
f0, f1 = synth.synth_pair(1920, 1080, seed=1)
ctx.upload(0, f0); ctx.upload(1, f1); ctx.build_pyramids(0); ctx.build_pyramids(1)
fl, _ = ctx.select(0, 5000, use_pyramid=True)
ctx.featbuf_upload(0, fl); ctx.featbuf_upload(1, fl)
for _ in range(3):
    ctx.track_async(0, 1, 0, 1, 5000)
ctx.sync()
lib = C.CDLL(dbg)
buf = (C.c_longlong * (256 * 32))()
assert lib.klt_debug_track_clocks(buf) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(256, 32)
out = ctx.featbuf_download(1, 5000)
t0 = t[:, 0].min()
names = ["start"] + ["L%d %s" % (l, n) for l in (2, 1, 0) for n in ("tmpl", "it0 smp", "it0 slv", "it1 smp", "it1 slv", "it2 smp", "it2 slv", "loop end", "residue")]
print("feature: start | per level: template, [sampled, solved] x iterations, loop end, residue   (ticks of 10 ns since the previous mark)")
for f in list(range(0, 24)) + [255]:
    row = t[f]
    marks = [(i, row[i]) for i in range(28) if row[i] >= t0 and row[i] - t0 < 10**7]
    marks.sort(key=lambda m: m[1])
    s = "%3d: %5d |" % (f, row[0] - t0)
    prev = row[0]
    for i, v in marks[1:]:
        s += " %s +%d" % (names[i].split(" ", 1)[1] if i % 9 != 1 else "|" + names[i], v - prev)
        prev = v
    s += "  || total %d  aux %x" % (prev - row[0], int(out["aux"][f]) & 0xfff)
    print(s)
tot = []
for f in range(256):
    row = t[f]; v = [x for x in row[:28] if x >= t0 and x - t0 < 10**7]
    tot.append((max(v) - row[0], row[0] - t0))
tot = np.array(tot)
print("wavefront life: mean %.0f  min %d  max %d ticks; start spread %d..%d" % (tot[:, 0].mean(), tot[:, 0].min(), tot[:, 0].max(), tot[:, 1].min(), tot[:, 1].max()))
