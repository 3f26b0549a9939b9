"""CPU-side checks of bench.py's line plumbing: the config sweep (benchlib/sweep.py) that folds cfg-1 / 3 / 4 / 5 into the default line."""
import json
import subprocess
import types

import pytest

from benchlib import sweep


def _child_line(parity=True, frac=0.4):
    return {"metric": "features tracked/sec", "value": 1.0e7, "unit": "features/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.1,
            "config": {"workload": "cfg-9: something; more text", "pairs_per_step": 256},
            "roofline": {"bound": "hbm", "kernel": "track", "frac": frac, "step_frac": 0.3, "launch_us": 40.0, "peak": 8000.0, "unit": "GB/s",
                         "kernels": {"track": {"frac": frac}}},
            "cpu_baseline": {"value": 3.0e4, "cores": 1, "kind": "port", "unit": "features/s", "all_cores": {"value": 1.0e5, "cores": 16, "ms_per_step": 1.0}},
            "parity_checked": parity, "parity_cases": 5, "max_abs_dx": 0.0,
            "extra": {"region_ms_per_step": {"regions": 30, "timed_s_total": 3.1}}}


def _fake_run(stdout, rc=0, stderr=""):
    def run(cmd, **kw):
        assert "--gpus" in cmd and "--config" in cmd and kw["capture_output"] and kw["timeout"] > 0
        for v in sweep.RANK_VARS:                         # a child is a one-GPU run of its own, whatever launched the parent
            assert v not in kw["env"]
        return types.SimpleNamespace(returncode=rc, stdout=stdout, stderr=stderr)
    return run


def test_sweep_folds_compact_records(monkeypatch):
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sweep.subprocess, "run", _fake_run("RCCL chatter\n" + json.dumps(_child_line()) + "\n"))
    out, failed = sweep.config_sweep()
    assert failed == [] and sorted(k for k in out if k != "note") == ["cfg1", "cfg3", "cfg4", "cfg5"]
    rec = out["cfg4"]
    assert rec["value"] == 1.0e7 and rec["ms_per_step"] == 0.1 and rec["steps"] == 20 and rec["parity_checked"] is True and rec["parity_cases"] == 5
    assert rec["roofline"] == {"kernel": "track", "frac": 0.4, "step_frac": 0.3, "launch_us": 40.0, "bound": "hbm", "peak": 8000.0, "unit": "GB/s"}
    assert rec["cpu_baseline"]["value"] == 3.0e4 and rec["cpu_baseline"]["cores"] == 1 and rec["cpu_baseline"]["all_cores"] == {"value": 1.0e5, "cores": 16}
    assert rec["wall_s"] >= 0 and rec["pairs_per_step"] == 256 and rec["step_is"] == "cfg-9: something"
    assert "--pairs 256" in rec["cmd"] and "--frames 512" in out["cfg5"]["cmd"]
    # the records travel inside the parent's line, whose fractions are checked before it is printed
    from benchlib.common import check_fractions
    assert check_fractions({"extra": {"configs": out}}) == []
    out["cfg3"]["roofline"]["frac"] = 1.2
    assert check_fractions({"extra": {"configs": out}})


@pytest.mark.parametrize("stdout, rc, what", [
    (json.dumps(_child_line(parity=False)) + "\n", 1, "exited with 1"),          # records differ from the oracle's: the child's own exit code
    (json.dumps(_child_line(parity=False)) + "\n", 0, "not checked"),            # oracle not built on the box: unchecked is a failure too
    ("", 139, "exited with 139"),                                                # crashed before its line
    ("{not json\n", 0, "unreadable"),
])
def test_sweep_reports_failures(monkeypatch, stdout, rc, what):
    monkeypatch.setattr(sweep.subprocess, "run", _fake_run(stdout, rc, "boom"))
    out, failed = sweep.config_sweep()
    assert failed == ["cfg1", "cfg3", "cfg4", "cfg5"]
    assert all(what in out[c]["error"] for c in failed)


def test_sweep_timeout_is_a_failure(monkeypatch):
    def run(cmd, **kw):
        raise subprocess.TimeoutExpired(cmd, kw["timeout"])
    monkeypatch.setattr(sweep.subprocess, "run", run)
    out, failed = sweep.config_sweep()
    assert len(failed) == 4 and "timed out" in out["cfg5"]["error"]
