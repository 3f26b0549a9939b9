"""bench.py on the GPU box (SURVEY 8 d): the driver's command line prints one clean record, figures above the roof are refused, every
config line carries checker / roofline / CPU baseline, dispatch timestamps never drop a launch silently.  The line's plumbing without a
GPU: test_bench_line.py.  (Folded by component from the round-3 / 4 files in round 6: the tests are unchanged.)"""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import REPO
from helpers import _api_modules, _records, default_cache, default_lists, make_tc, params_from_tc
from pyfeaturetrack_amd import synth
from pyfeaturetrack_amd.params import affine_params_from_tc

pytestmark = pytest.mark.gpu


def run_bench(extra_args, timeout=1500, **env_kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "KLT_RDZV_FILE")}
    env.update(env_kw)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_the_drivers_command_line_prints_a_clean_record():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- exactly what the driver runs at round end: no figure above its roof,
    iteration counters that describe the launches they are divided by, 72 distinct pairs all checked against the oracle, a timed
    phase long enough to be seen from outside the process, and a value that does not depend on K."""
    line = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    roof = line["roofline"]
    assert line["parity_checked"] is True and line["parity_cases"] == 72 and line["max_abs_dx"] <= 1e-3
    assert line["config"]["resident_pairs"] == 72 and line["config"]["pairs_per_step"] == 72 and line["config"]["contexts"] == 3
    assert 0 < roof["frac"] < 1 and 0 < roof["frac_moved"] < roof["frac"] and 0 < roof["step_frac"] < 1
    assert roof["kernel"] == "smooth_grad_l0" and roof["launch_us_source"] == "dispatch timestamps"
    for name, k in roof["kernels"].items():
        assert 0 <= k["frac"] < 1 and k["GBps"] < roof["peak"], name
    for l, it in enumerate(roof["newton_iterations_per_level"]):
        assert 5000 <= it <= 25000, "level %d: %.0f Newton iterations per pair" % (l, it)
    assert abs(roof["step_algorithmic_bytes"] / roof["step_algorithmic_bytes_formula"] - 1) < 1e-6
    assert line["extra"]["region_ms_per_step"]["timed_s_total"] >= 2.0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] == 1
    # (round 6) ... and the other four BASELINE configs at their BASELINE counts, each run by a child process of that command and folded
    # into the line (benchlib/sweep.py): driver-timed, checked against the oracle, with their own roofline and CPU baseline
    configs = line["extra"]["configs"]
    for name, steps, cases in (("cfg1", 20, 2), ("cfg3", 20, 5), ("cfg4", 10, 256), ("cfg5", 511, 5)):
        rec = configs[name]
        assert "error" not in rec and rec["parity_checked"] is True and rec["parity_cases"] == cases and rec["steps"] == steps, (name, rec)
        assert rec["value"] > 0 and 0 < rec["roofline"]["frac"] < 1 and 0 < rec["roofline"]["step_frac"] < 1 and rec["cpu_baseline"]["value"] > 0, (name, rec)
        assert rec["timed_s_total"] >= 2.0 and rec["wall_s"] < 300, (name, rec)
    assert configs["cfg4"]["pairs_per_step"] == 256 and configs["cfg5"]["frames"] == 512
    # ... and the reference-shaped API on the reference's own image type (Pillow, mode "L") next to numpy frames
    x = line["extra"]
    assert "api_error" not in x and x["api_ms_per_KLTTrackFeatures_pil"] < 1.5 * x["api_ms_per_KLTTrackFeatures"], (x.get("api_error"), x.get("api_pil_note"))
    assert x["sequence_from_host"]["4k"]["arrangement"] == "one_copy_stream"
    # the same figures from a run with five times the steps per region
    long = run_bench(["--gpus", "1", "--steps", "100", "--warmup", "5", "--repeats", "8", "--no-cpu-baseline", "--no-extras"])
    assert abs(long["roofline"]["step_algorithmic_bytes"] / roof["step_algorithmic_bytes"] - 1) < 0.05
    assert abs(long["value"] / line["value"] - 1) < 0.08, (long["value"], line["value"])     # (seen: 0.3-1.7 %)


def test_bench_refuses_figures_above_the_roof():
    import bench
    ok = {"roofline": {"peak": 8000.0, "frac": 0.4, "step_frac": 0.2, "kernels": {"track": {"GBps": 900.0, "frac": 0.11}}}}
    assert bench.check_fractions(ok) == []
    for path, bad in (("step_frac", {"roofline": {"peak": 8000.0, "frac": 0.4, "step_frac": 2.6}}),
                      ("GBps", {"roofline": {"peak": 8000.0, "kernels": {"track": {"GBps": 72833.0, "frac": 0.5}}}}),
                      ("frac_moved", {"roofline": {"frac_moved": 1.01}})):
        assert any(path in m for m in bench.check_fractions(bad)), path
    fd = os.open(os.devnull, os.O_WRONLY)
    try:
        with pytest.raises(SystemExit):
            bench.emit(fd, {"roofline": {"peak": 8000.0, "frac": 0.4, "step_frac": 2.6}})
    finally:
        os.close(fd)


@pytest.mark.parametrize("cfg,extra", [("cfg1", []), ("cfg3", ["--steps", "10", "--repeats", "5"])])
def test_every_config_line_carries_checker_roofline_and_baseline(cfg, extra):
    line = run_bench(["--config", cfg] + extra)
    assert line["parity_checked"] is True, line
    assert 0 < line["roofline"]["frac"] < 1 and 0 < line["roofline"]["step_frac"] < 1
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    if cfg == "cfg3":
        k = line["roofline"]["kernels"]
        assert k["affine_check"]["timed_by"] == "dispatch timestamps" and k["track"]["timed_by"] == "dispatch timestamps"
        assert line["roofline"]["affine_iterations_per_checked_feature"] >= 1


def test_dispatch_timestamps_never_drop_a_launch_silently():
    """klt_timing_enable(ctx, 2) times single-launch families by their dispatch's own timestamps, which only launches that
    go through klt_launch carry.  On a frame whose width is not a multiple of 4 the summed-area passes take the barrier-coupled
    fallback kernels: every scope is either measured or counted in "<family>!unstamped" -- never lost."""
    from helpers import synth251_frames
    from pyfeaturetrack_amd.backend import Context
    c = Context(0)
    try:
        for frame, tc in ((synth251_frames()[0], make_tc(levels=2, ss=2)), (synth.synth_pair(640, 480, 2)[0], make_tc(levels=2, ss=4))):
            c.configure(tc)
            c.upload(0, frame)
            c.build_pyramids(0)
            for mode in (1, 2):
                c.timing_enable(mode)
                c.select(0, 40, use_pyramid=True)
                t = {k["name"]: k["launches"] for k in c.timing_read()}
                c.timing_enable(0)
                for fam in ("sat_rows", "sat_cols"):
                    assert t.get(fam, 0) + t.get(fam + "!unstamped", 0) == 1, (frame.shape, mode, t)
                assert mode == 2 or not any(k.endswith("!unstamped") for k in t)
    finally:
        c.close()
