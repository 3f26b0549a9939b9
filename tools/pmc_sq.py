#!/usr/bin/env python3
"""Per-launch SQ counters of each kernel family from rocprofv3 --pmc passes (kernel-trace only):

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/pmc_sq1 -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT ... -d gpurun_out/pmc_sq2 -- ...
    python tools/pmc_sq.py profiles/sq_counters.json gpurun_out/pmc_sq1 gpurun_out/pmc_sq2

Counters are summed over the device per launch; the value reported for a family is that of its largest launch.
bench.py reads SQ_INSTS_VALU of the dominant kernel for `roofline.issue_bound` (wavefront-instructions / 1024 SIMDs x the
calibrated 4.5 clocks per instruction of tools/mb/valu_rate.hip and fp64_mix.hip)."""
import collections
import csv
import glob
import json
import os
import sys

from pmc_traffic import family_of, provenance


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = collections.defaultdict(float)
            fam_of = {}
            for r in csv.DictReader(open(f)):
                fam = family_of(r["Kernel_Name"])
                if fam:
                    key = (r["Dispatch_Id"], r["Counter_Name"])
                    per_dispatch[key] += float(r["Counter_Value"])
                    fam_of[r["Dispatch_Id"]] = fam
            for (disp, counter), v in per_dispatch.items():
                acc[fam_of[disp]][counter].append(v)
    out = {"_unit": "counter value per launch, summed over the device (largest launch of the family)",
           "_meta": provenance("SQ counter passes of bench.py --inflight 1")}
    for fam in sorted(acc):
        out[fam] = {c: max(v) for c, v in sorted(acc[fam].items())}
        out[fam]["_launches"] = max(len(v) for v in acc[fam].values())
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
