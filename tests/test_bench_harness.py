"""bench.py as a complete multi-rank command (VERDICT r2 "missing" item 1): `python bench.py --gpus N` with no torchrun
environment spawns its N ranks, rank 0 prints ONE JSON line with n_gpus = N, the headline (replica, weak scaling) and the
per-sample-latency mode (cfg + frame groups, strong scaling) measured in the same run.  There is no GPU here, so the ranks run
the tiny network on the CPU emulation of the C-ABI over gloo (`--emulate-kernels`: a self-test of the harness, never a
measurement — the line says so)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(gpus, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--backend", "gloo", "--config", "tiny",
           "--emulate-kernels", "--steps", "1", "--warmup", "1", *extra]
    pr = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, pr.stdout                      # ONE line, from rank 0 only
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_its_ranks_and_reports_both_scalings():
    d = _run(2)
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["config"]["parallelism"] == "replica x2"
    assert d["emulated_kernels"] is True and "SELF-TEST" in d["metric"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"] + 1e-9     # whole-job steps/s
    s = d["strong_scaling"]
    assert s["scaling"] == "strong" and s["ranks_per_sample"] == 2 and s["parallelism"] == "cfg x2"
    assert s["cfg_all_gathers_per_step"] == 1 and "2 ranks" in s["collective_backend"]
    # BASELINE config 4 literally: the VIEWS of one sample sharded over the 2 ranks, in the same line (round 5)
    v = d["strong_scaling_views"]
    assert "error" not in v, v
    assert v["scaling"] == "strong" and v["parallelism"] == "views x2" and v["exchange"]["neighbour_exchanges_per_step"] > 0


@pytest.mark.parametrize("mode,name", [("frames", "frames x2"), ("cfg", "cfg x2"), ("views", "views x2")])
def test_bench_explicit_sharded_modes(mode, name):
    d = _run(2, ("--parallelism", mode))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["parallelism"] == name
    assert "strong_scaling" not in d
    if mode == "frames":
        ex = d["config"]["exchange"]
        assert ex["frame_exchanges_per_step"] > 0 and ex["MB_sent_per_rank_and_step"] > 0
    if mode == "views":
        ex = d["config"]["exchange"]
        assert ex["neighbour_exchanges_per_step"] > 0 and ex["MB_sent_per_rank_and_step"] > 0


@pytest.mark.timeout(900)
def test_bench_gpus_8_is_the_drivers_scaling_command():
    """what the driver's 8-GPU run executes (here: gloo + emulated kernels, tiny network with 4 frames): replica x8 as the headline
    and ONE sample over 2 CFG halves x 4 frame groups under `strong_scaling`, in one line, unattended"""
    d = _run(8, ("--frames", "4"))
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["parallelism"] == "replica x8"
    s = d["strong_scaling"]
    assert "error" not in s, s
    assert s["parallelism"] == "cfg x2 . frames x4" and s["ranks_per_sample"] == 8 and s["samples_in_flight"] == 1
    assert s["exchange"]["frame_exchanges_per_step"] > 0 and s["per_sample_latency_ms"] > 0
    # SURVEY 8(e)'s 8-GPU grid as the second strong-scaling record: 2 CFG halves x 2 view groups x 2 frame groups
    v = d["strong_scaling_views"]
    assert "error" not in v, v
    assert v["parallelism"] == "cfg x2 . frames x2 . views x2" and v["ranks_per_sample"] == 8
    assert v["exchange"]["frame_exchanges_per_step"] > 0 and v["view_exchange"]["neighbour_exchanges_per_step"] > 0


def test_bench_yaml_exact_setup_runs_multi_rank():
    """BASELINE config 5's set-up (25-step schedule, last-frame `concat`, share-noise initial latent) through the multi-rank command:
    replicas as the headline, one sample over a CFG pair underneath"""
    d = _run(2, ("--yaml-exact",))
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "replica x2"
    assert d["strong_scaling"]["parallelism"] == "cfg x2" and "error" not in d["strong_scaling"]
