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


# ---------------------------------------------------------------------------------------------- the N > 1 line validates itself
class _FakeCtx:
    """a context whose communicator has `size` ranks and whose all-reduce sees the vectors of `peers` (other ranks' contributions)"""

    def __init__(self, size, rank, peers=()):
        self.size, self.rank, self.peers = size, rank, list(peers)

    def comm_info(self):
        return self.size, self.rank

    def comm_allreduce_max(self, values):
        out = list(values)
        for p in self.peers:
            out = [max(a, b) for a, b in zip(out, p)]
        return out


def _ranks(monkeypatch, world, gpus, rank=0):
    from benchlib.common import Ranks
    monkeypatch.setenv("RANK", str(rank))
    monkeypatch.setenv("WORLD_SIZE", str(world))
    monkeypatch.delenv("KLT_FORCE_DIST", raising=False)
    monkeypatch.delenv("KLT_RANKS_SHARE_DEVICE", raising=False)
    return Ranks(types.SimpleNamespace(gpus=gpus))


def _peer(size, rank, ms):
    return [-float(size), float(size), ms, -ms] + [1.0 if r == rank else 0.0 for r in range(12)]


def test_rank_validation_reports_what_rccl_says_and_every_ranks_own_time(monkeypatch):
    from benchlib.common import Ranks
    r = _ranks(monkeypatch, 4, 4)
    r.ctxs = [_FakeCtx(4, 0, [_peer(4, 1, 2.5), _peer(4, 2, 2.0), _peer(4, 3, 2.2)]), _FakeCtx(4, 0), _FakeCtx(4, 0)]
    r.local_s = [0.040, 0.042, 0.041]                    # this rank's regions of 20 steps: median 2.05 ms per step
    v = r.validation(20)
    assert v["rccl_ranks"] == 4 and v["rccl_ranks_largest_communicator"] == 4 and v["rccl_rank_ids_seen"] == 4 and v["gpus_asked"] == 4
    assert abs(v["per_rank_ms_per_step"]["max"] - 2.5) < 1e-9 and abs(v["per_rank_ms_per_step"]["min"] - 2.0) < 1e-9
    assert "rccl_validation_failed" not in v
    Ranks.fail_on_validation(v)
    Ranks.fail_on_validation({})                         # one GPU, no communicator: nothing to validate


def test_rank_validation_fails_the_run_when_a_rank_saw_fewer_peers_than_asked(monkeypatch):
    from benchlib.common import Ranks
    # --gpus 8 under a launcher that started 4 ranks
    r = _ranks(monkeypatch, 4, 8)
    r.ctxs = [_FakeCtx(4, 0, [_peer(4, k, 2.0) for k in (1, 2, 3)])]
    v = r.validation(1)
    assert v["rccl_ranks"] == 4 and "--gpus 8" in v["rccl_validation_failed"]
    with pytest.raises(SystemExit):
        Ranks.fail_on_validation(v)
    # one rank's communicator is smaller than the others' (two launches sharing a rendezvous file)
    r = _ranks(monkeypatch, 4, 4)
    r.ctxs = [_FakeCtx(4, 0, [_peer(4, 1, 2.0), _peer(2, 1, 2.0), _peer(4, 3, 2.0)])]
    v = r.validation(1)
    assert v["rccl_ranks"] == 2 and v["rccl_ranks_largest_communicator"] == 4 and "rccl_validation_failed" in v
    # two processes answered as the same rank
    r = _ranks(monkeypatch, 4, 4)
    r.ctxs = [_FakeCtx(4, 0, [_peer(4, 1, 2.0), _peer(4, 1, 2.0), _peer(4, 3, 2.0)])]
    v = r.validation(1)
    assert v["rccl_rank_ids_seen"] == 3 and "different rank numbers" in v["rccl_validation_failed"]
