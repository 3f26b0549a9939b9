"""The other four BASELINE configs next to the headline: after the cfg-2 line's contexts are closed, `bench.py --gpus 1`
runs `bench.py --config cfg1 | cfg3 | cfg4 | cfg5` at the BASELINE counts (cfg-4: 256 pairs on the one GPU; cfg-5: the 512-frame clip), each
in a fresh child process that opens the GPU itself (started as a child and waited for -- never an exec: this process has initialised the GPU),
and folds a compact record of each child's line into `extra.configs`.  A child that fails -- its records differ from the oracle's, a
fraction of peak above 1, a crash, a timeout -- leaves an `error` record and makes the parent exit non-zero AFTER it has printed its line."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, arguments, timeout in s).  Steps: cfg-1 / cfg-3 as the driver's own default call; cfg-4 ten passes over all 256 pairs per region;
# cfg-5 one pass over the 512-frame clip per region.  Every config times >= 5 regions and >= 6 s of GPU work, as the headline does; the
# four children take 4 + 18 + 27 + 24 s at 3 s of timed work each (gpurun, round 6), the whole default run ~2 min.
SWEEP = (
    ("cfg1", ["--config", "cfg1", "--steps", "20", "--warmup", "5"], 240),
    ("cfg3", ["--config", "cfg3", "--steps", "20", "--warmup", "5"], 300),
    ("cfg4", ["--config", "cfg4", "--pairs", "256", "--steps", "10", "--warmup", "2"], 420),
    ("cfg5", ["--config", "cfg5", "--frames", "512", "--steps", "511"], 600),
)
RANK_VARS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "KLT_FORCE_DIST", "KLT_RANKS_SHARE_DEVICE", "KLT_RDZV_FILE", "KLT_SPAWNED",
             "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE")


def compact(line, wall_s):
    """what extra.configs keeps of a config's own line (the full line: `python bench.py --config <name>`)"""
    roof, cpu = line.get("roofline") or {}, line.get("cpu_baseline") or {}
    rec = {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line["steps"],
           "step_is": (line.get("config") or {}).get("workload", "").split(";")[0][:160],
           "roofline": {"kernel": roof.get("kernel"), "frac": roof.get("frac"), "step_frac": roof.get("step_frac"),
                        "launch_us": roof.get("launch_us"), "bound": roof.get("bound"), "peak": roof.get("peak"), "unit": roof.get("unit")},
           "cpu_baseline": {"value": cpu.get("value"), "cores": cpu.get("cores"), "kind": cpu.get("kind"), "unit": cpu.get("unit"),
                            "all_cores": {k: (cpu.get("all_cores") or {}).get(k) for k in ("value", "cores")} if cpu.get("all_cores") else None},
           "parity_checked": line.get("parity_checked"), "parity_cases": line.get("parity_cases"), "max_abs_dx": line.get("max_abs_dx"),
           "timed_regions": ((line.get("extra") or {}).get("region_ms_per_step") or {}).get("regions"),
           "timed_s_total": ((line.get("extra") or {}).get("region_ms_per_step") or {}).get("timed_s_total"),
           "wall_s": wall_s}
    if (line.get("config") or {}).get("pairs_per_step"):
        rec["pairs_per_step"] = line["config"]["pairs_per_step"]
    if (line.get("config") or {}).get("frames"):
        rec["frames"] = line["config"]["frames"]
    return rec


def run_one(name, argv, timeout, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in RANK_VARS}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + list(argv) + list(extra_args)
    t = time.perf_counter()
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s" % timeout, "wall_s": time.perf_counter() - t, "cmd": " ".join(cmd[1:])}
    except OSError as e:
        return {"error": "%s: %s" % (type(e).__name__, e), "wall_s": time.perf_counter() - t, "cmd": " ".join(cmd[1:])}
    wall = time.perf_counter() - t
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    rec = None
    if lines:
        try:
            rec = compact(json.loads(lines[-1]), wall)
        except (ValueError, KeyError) as e:
            rec = {"error": "unreadable line (%s: %s)" % (type(e).__name__, e), "wall_s": wall}
    if r.returncode != 0 or rec is None:
        # (a child whose records differ from the oracle's prints its line -- with parity_checked false -- and then exits non-zero)
        rec = dict(rec or {}, error="bench.py %s exited with %d: %s" % (" ".join(argv), r.returncode, r.stderr.strip()[-400:]), wall_s=wall)
    elif "error" not in rec and rec.get("parity_checked") is not True:
        rec["error"] = "the config's records were not checked against the oracle (parity_checked = %r)" % rec.get("parity_checked")
    rec["cmd"] = "python bench.py " + " ".join(cmd[2:])
    return rec


def config_sweep(extra_args=()):
    """{name: compact record} for cfg-1, cfg-3, cfg-4, cfg-5, one child process after the other (one GPU: they must not run side by side),
    and the list of configs that failed"""
    out, failed = {}, []
    for name, argv, timeout in SWEEP:
        out[name] = run_one(name, argv, timeout, extra_args)
        if "error" in out[name]:
            failed.append(name)
    out["note"] = ("the other BASELINE configs at their BASELINE counts, each `python bench.py --config <name>` in a child process of its own, "
                   "started after the headline's contexts were closed (one GPU: one after the other); every record's roofline.frac is the "
                   "config's dominant kernel against 8 TB/s, step_frac its whole step; parity_checked = what the config timed equals the CPU "
                   "oracle's records (cfg-4: ALL 256 pairs; cfg-5: the first frames; cfg-3's affine check is UNPINNED -- the reference does not "
                   "define it); a failing config makes this run exit non-zero; the full lines: the `cmd` of each record")
    return out, failed
