#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, with --kernel-trace only)
into HBM bytes per launch for each kernel family, as MI355X_MICROARCH.md prescribes:

    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced
streaming read.  The factor 2 is re-calibrated here on a kernel of this code base with a known byte
count: sat_cols_kernel reads exactly 3*ncols*nrows*4 bytes with one dword per lane (and writes the
same amount, which WRITE_SIZE reports exactly).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json
"""
import collections
import csv
import glob
import json
import os
import sys

FAMILY = [("smooth_grad_rb<unsigned char, true", "smooth_grad_l0"), ("smooth_grad_rb<float, true", "smooth_grad_l0"),
          ("smooth_grad_rb<float, false", "gradients"), ("smooth_grad_kernel", "smooth_grad_l0"),
          ("pyr_reduce", "pyramid_reduce"), ("pyr_vreduce", "pyramid_reduce"), ("track_kernel", "track"), ("sat_rows", "sat_rows"), ("sat_cols", "sat_cols"),
          ("eigen_kernel", "eigen_keys"), ("eigen_hist_kernel", "eigen_keys"), ("cols_eigen_pipe", "eigen_keys"), ("nms_kernel", "nms"), ("mis_init", "min_distance_init"),
          ("mis_round", "min_distance_pass"), ("mis_compact", "min_distance_compact"), ("mis_rank", "min_distance_rank"),
          ("mis_place", "min_distance_place"), ("mis_prepare", "min_distance_prepare"), ("mask_hist", "mask_hist"),
          ("seed_fill", "seed_fill"), ("affine_kernel", "affine_check")]


def family_of(kernel):
    for pat, fam in FAMILY:
        if pat in kernel:
            return fam
    return None


def read(dirname, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                fam = family_of(r["Kernel_Name"])
                if fam:
                    acc[fam].append(float(r["Counter_Value"]))
    return acc


# everything a per-launch counter depends on: the kernels, the shared load / store helpers (klt_internal.h) and the launch geometry
# (the files of the C ABI that enqueue those kernels: pyramid build, selection, tracker)
KERNEL_SOURCES = ["pyfeaturetrack_amd/csrc/pyramid_kernels.hip", "pyfeaturetrack_amd/csrc/track_kernels.hip",
                  "pyfeaturetrack_amd/csrc/select_kernels.hip", "pyfeaturetrack_amd/csrc/sat_pipeline.hip",
                  "pyfeaturetrack_amd/csrc/klt_internal.h", "pyfeaturetrack_amd/csrc/klt_context.h",
                  "pyfeaturetrack_amd/csrc/api_frames.hip", "pyfeaturetrack_amd/csrc/api_select.hip", "pyfeaturetrack_amd/csrc/api_track.hip"]


def provenance(what):
    """`_meta` record bench.py checks before it quotes a committed counter: where the numbers come from and the hash of the
    kernel sources they were collected for (a counter of an older kernel is dropped, not quoted)."""
    import hashlib
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shas = {}
    for rel in KERNEL_SOURCES:
        try:
            shas[rel] = hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest()[:16]
        except OSError:
            pass
    return {"source": "%s, builder gpurun (rocprofv3 --pmc, kernel-trace only), tag %s, %s" %
                      (what, os.environ.get("KLT_PROFILE_TAG", "?"), time.strftime("%Y-%m-%d")),
            "kernel_source_sha16": shas,
            # frame pairs that share a launch of the cfg-2 passes (bench.py --batch): per-launch counters only describe launches of that size
            "cfg2_pairs_per_launch": int(os.environ.get("KLT_PROFILE_BATCH", "2"))}


def main():
    fetch = read(sys.argv[1], "FETCH_SIZE")
    write = read(sys.argv[2], "WRITE_SIZE")
    out = {"_unit": "bytes per launch (largest launch of the family)", "_formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "_meta": provenance("FETCH_SIZE / WRITE_SIZE passes of bench.py --inflight 1")}
    if "sat_cols" in fetch and "sat_cols" in write:
        out["_calibration"] = {"kernel": "SAT column pass (reads == writes == 3*ncols*nrows*4 bytes)",
                               "WRITE_SIZE_KiB": max(write["sat_cols"]), "FETCH_SIZE_KiB": max(fetch["sat_cols"]),
                               "fetch_over_known_read": max(fetch["sat_cols"]) / max(write["sat_cols"])}
    for fam in sorted(set(fetch) | set(write)):
        f = max(fetch.get(fam, [0.0]))
        w = max(write.get(fam, [0.0]))
        out[fam] = (2.0 * f + w) * 1024.0
        out["_" + fam + "_raw_KiB"] = {"FETCH_SIZE": f, "WRITE_SIZE": w, "launches": len(fetch.get(fam, []))}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
