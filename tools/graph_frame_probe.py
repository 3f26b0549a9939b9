#!/usr/bin/env python3
"""Does a captured HIP graph take the host out of a sequence's frame?  

One frame of the replace-every-frame sequence -- build + score preparation of frame k+1 on the build stream, next to the replacement
selection of frame k and followed by the tracker k -> k+1 -- is (a) enqueued eagerly, frame after frame over a ring of three slots,
exactly as bench.py --config cfg5 / KLTTrackSequence do, and (b) captured ONCE into a HIP graph (klt_debug_graph, an experiment hook that
is not part of the ABI) and replayed.  The replay repeats the same frame (its list is restored by a 320 KB device copy before every
launch, which the eager loop does too), so its kernels do the work of a real frame; what differs is how the launches reach the GPU.

    python tools/graph_frame_probe.py [4k|1080p] [reps]
"""
import ctypes as C
import json
import os
import sys
import time

import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the capture hook (klt_debug_graph) is not in the product library: a private copy of it, api_compat.hip compiled with -DKLT_GRAPH_PROBE
_src = os.path.join(ROOT, "pyfeaturetrack_amd", "csrc")
_lib = os.path.join(ROOT, "gpurun_out", "libkltgpu_graph.so")
os.makedirs(os.path.dirname(_lib), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                "-I" + os.path.join(ROOT, "include"), "-Wno-cuda-compat", "-Wno-unused-result", "-DKLT_GRAPH_PROBE", "-c", os.path.join(_src, "api_compat.hip"),
                "-o", "/tmp/klt_api_graph.o"], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", _lib, "/tmp/klt_api_graph.o"] +
               [os.path.join(_src, f) for f in ("api_context.o", "api_frames.o", "api_featbuf.o", "api_select.o", "api_track.o", "api_comm.o", "host_pool.o", "conv_kernels.o", "pyramid_kernels.o", "select_kernels.o", "sat_pipeline.o", "track_kernels.o",
                                                "affine_kernels.o", "comm.o")] + ["-ldl", "-lpthread"], check=True)
os.environ["KLT_GPU_LIB"] = _lib
from pyfeaturetrack_amd import synth                                     # noqa: E402
from pyfeaturetrack_amd.backend import Context                           # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext                   # noqa: E402


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "4k"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    w, h, n = (3840, 2160, 20000) if size == "4k" else (1920, 1080, 5000)
    tc = KLT_TrackingContext()
    tc.nPyramidLevels, tc.subsampling = 3, 4
    tc.KLTUpdateTCBorder()
    tc.max_residue = 10.0
    ctx = Context(0)
    ctx.configure(tc)
    lib = ctx._lib
    lib.klt_debug_graph.restype = C.c_int
    lib.klt_debug_graph.argtypes = [C.c_void_p, C.c_int]
    phases = synth.sequence_phases(w, h, 4, workers=8)
    frames = list(synth.periodic_sequence(w, h, 4, 9, phases=phases))
    S = [0, 1, 2]
    for k in range(3):
        ctx.upload(S[k], frames[k])
    FB0, FB1, SAVED = 10, 11, 12
    ctx.set_option(15, 1)                                       # KLT_OPT_BUILD_STREAM

    def stage(slot):
        ctx.build_pyramids(slot, sync=False)
        ctx.select_prepare(slot)

    def eager_sequence(nframes):
        """the loop of bench.py --config cfg5 over frames that cycle through the ring (frame content k % 3: the list jumps back every third
        frame, which costs a few more lost features than a real sequence has)"""
        redone = 0
        ctx.build_pyramids(S[0], sync=False)
        ctx.select_async(S[0], 1, True, FB0, n)
        stage(S[1])
        ctx.track_async(S[0], S[1], FB0, FB1, n)
        for k in range(1, nframes):
            cur, nxt = S[k % 3], S[(k + 1) % 3]
            fb_cur, fb_nxt = (FB0, FB1)[k % 2], (FB0, FB1)[(k + 1) % 2]
            ctx.select_begin(cur, 2, True, fb_cur, n)
            stage(nxt)
            ctx.track_async(cur, nxt, fb_cur, fb_nxt, n)
            if ctx.select_finish():
                redone += 1
                ctx.track_async(cur, nxt, fb_cur, fb_nxt, n)
        ctx.sync()
        return redone

    eager_sequence(12)                                          # sizes every buffer, settles the number of passes per look
    t = time.perf_counter()
    redone = eager_sequence(reps)
    eager_ms = (time.perf_counter() - t) / (reps - 1) * 1e3

    # the fixed frame: list after tracking 0 -> 1 (with its lost features) saved; per repetition: restore it, stage frame 2, replace on
    # frame 1, track 1 -> 2
    ctx.build_pyramids(S[0], sync=False)
    ctx.select_async(S[0], 1, True, FB0, n)
    ctx.build_pyramids(S[1], sync=False)
    ctx.track_async(S[0], S[1], FB0, SAVED, n)
    saved = ctx.featbuf_download(SAVED, n)
    ctx.featbuf_upload(FB1, saved)
    ctx.featbuf_alloc(FB0, n)
    lost = int((saved["val"] < 0).sum())
    dev = lambda fb: C.c_void_p(ctx.featbuf_devptr(fb))          # noqa: E731

    def frame_eager(prepare_cur):
        if prepare_cur:
            ctx.select_prepare(S[1])                            # (the eager repetition has to re-prepare the scores its selection consumes)
        ctx.featbuf_upload(FB1, saved)
        stage(S[2])
        ctx.select_begin(S[1], 2, True, FB1, n)
        ctx.track_async(S[1], S[2], FB1, FB0, n)
        ctx.select_finish()

    frame_eager(True)
    ctx.sync()
    want = ctx.featbuf_download(FB0, n).tobytes()
    t = time.perf_counter()
    for _ in range(reps):
        frame_eager(True)
    ctx.sync()
    fixed_eager_ms = (time.perf_counter() - t) / reps * 1e3

    # capture: the build first (the build stream forks at the graph's root), then the selection, then the tracker
    ctx.select_prepare(S[1])
    ctx.sync()
    rc = lib.klt_debug_graph(ctx._h, 0)
    assert rc == 0, lib.klt_last_error(ctx._h)
    try:
        stage(S[2])
        ctx.select_begin(S[1], 2, True, FB1, n)
        ctx.track_async(S[1], S[2], FB1, FB0, n)
    finally:
        nodes = lib.klt_debug_graph(ctx._h, 1)
    assert nodes > 0, lib.klt_last_error(ctx._h)
    rows = np.frombuffer(saved.tobytes(), np.uint8)
    pinned = ctx.pinned_array((rows.size,), np.uint8)
    pinned[:] = rows

    def frame_graph():
        ctx._check(lib.klt_featbuf_upload_async(ctx._h, FB1, pinned.ctypes.data, n))
        ctx._check(lib.klt_debug_graph(ctx._h, 2))

    frame_graph()
    ctx.sync()
    same = ctx.featbuf_download(FB0, n).tobytes() == want
    host = 0.0
    t = time.perf_counter()
    for _ in range(reps):
        a = time.perf_counter()
        frame_graph()
        host += time.perf_counter() - a
    ctx.sync()
    graph_ms = (time.perf_counter() - t) / reps * 1e3
    # the same with the host waiting for every frame (a look per frame)
    t = time.perf_counter()
    for _ in range(reps):
        frame_graph()
        ctx.sync()
    graph_sync_ms = (time.perf_counter() - t) / reps * 1e3
    lib.klt_debug_graph(ctx._h, 3)
    out = {"size": size, "features": n, "lost_in_the_fixed_frame": lost, "graph_nodes": nodes, "records_equal": bool(same),
           "eager_sequence_ms_per_frame": eager_ms, "eager_sequence_trackers_repeated": redone,
           "eager_fixed_frame_ms": fixed_eager_ms, "eager_fixed_frame_note": "includes the re-preparation of the current frame's scores (3 launches the sequence does not have)",
           "graph_replay_ms_per_frame": graph_ms, "graph_host_ms_per_launch": host / reps * 1e3,
           "graph_replay_synchronised_ms_per_frame": graph_sync_ms}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
