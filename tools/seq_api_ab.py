"""KLTTrackSequence (the reference-shaped sequence call) with its uploads on one / two copy streams, alternating in one process."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyfeaturetrack_amd import synth, trackSequence          # noqa: E402
from pyfeaturetrack_amd.klt import KLT_TrackingContext       # noqa: E402


def clip(w, h, seed, nframes):
    base = synth.synth_base(w, h, seed)
    distinct = [synth.synth_frame(w, h, seed, k, base=base) for k in range(16)]
    order = list(range(16)) + list(range(14, 0, -1))
    return [distinct[order[k % 30]] for k in range(nframes)]


def main():
    fd = os.dup(1)
    os.dup2(2, 1)
    out = {}
    for name, (w, h, n, seed) in (("4k", (3840, 2160, 20000, 4)), ("1080p", (1920, 1080, 5000, 1))):
        tc = KLT_TrackingContext()
        tc.nPyramidLevels, tc.subsampling = 3, 4
        tc.KLTUpdateTCBorder()
        tc.max_residue = 10.0
        frames = clip(w, h, seed, 256)
        trackSequence.KLTTrackSequence(tc, frames, n)
        arrangements = {"1_stream_1_helper": (1, 1), "1_stream_2_helpers": (1, 2), "2_streams_1_helper": (2, 1), "2_streams_2_helpers": (2, 2)}
        runs = {a: [] for a in arrangements}
        for _ in range(5):
            for a, (streams, helpers) in arrangements.items():
                trackSequence.SEQUENCE_COPY_STREAMS, trackSequence.STAGER_WORKERS = streams, helpers
                t = time.perf_counter()
                trackSequence.KLTTrackSequence(tc, frames, n)
                runs[a].append((time.perf_counter() - t) * 1e3 / 255)
        out[name] = {a: {"median": statistics.median(v), "runs": v} for a, v in runs.items()}
    os.write(fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
