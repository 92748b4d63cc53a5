"""bench.py's N > 1 control flow, rehearsed on the one-GPU box: GAME_BENCH_REHEARSE=1 puts every rank on GPU 0 and carries
the collectives over gloo (staged through host memory), so what is checked is the flow - rank spawning, the frame / band /
index splits, the halo exchange between the two phases of a filtered frame, the position all-gather, the one JSON line -
not a rate. Real N-GPU runs use RCCL and refuse to fall back for the configs that exchange data."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args, rehearse=True, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    if rehearse:
        env["GAME_BENCH_REHEARSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-others", *args], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    return p, (json.loads(lines[-1]) if lines else None)


def test_strong_scaling_splits_one_job():
    p, one = _bench("--gpus", "1", "--config", "c2", "--frames", "40", "--batch", "10", "--steps", "2", "--warmup", "1", rehearse=False)
    assert p.returncode == 0, p.stderr[-2000:]
    p, two = _bench("--gpus", "2", "--config", "c2", "--frames", "40", "--batch", "10", "--steps", "2", "--warmup", "1", "--scaling", "strong")
    assert p.returncode == 0, p.stderr[-2000:]
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and "REHEARSAL" in two["data"]
    # the same 40-frame job either way; each rank renders half of it
    assert two["config"]["job_frames_per_step"] == one["config"]["job_frames_per_step"] == 40
    assert two["config"]["frames_per_step"] == 20 and two["config"]["parallelism"] == "frames2"
    p, weak = _bench("--gpus", "2", "--config", "c2", "--frames", "40", "--batch", "10", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0 and weak["scaling"] == "weak" and weak["config"]["job_frames_per_step"] == 80


def test_filtered_feedback_bands_trade_halo_rows():
    p, rec = _bench("--gpus", "2", "--config", "c4", "--blur", "2", "--frames", "5", "--steps", "1", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "bands2+halo2" and rec["control_backend"] == "gloo"
    assert rec["roofline"]["bytes_per_pixel"] == 16 and rec["value"] > 0
    p, rec = _bench("--gpus", "1", "--config", "c2", "--blur", "2", rehearse=False)
    assert p.returncode != 0 and "--blur applies to --config c4" in p.stderr


def test_particle_ranges_with_position_exchange():
    p, rec = _bench("--gpus", "2", "--config", "c5", "--particles", "65536", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert rec["config"]["parallelism"] == "index2" and rec["value"] > 0
