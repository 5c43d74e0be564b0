"""The example script runs end to end on a GPU box (small batch)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rollout_example_runs(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "rollout.py"), "256", "1", str(tmp_path)], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "env-steps/s" in out.stdout and "episode 0" in out.stdout
    from ris_vec_marl_amd.metrics import read_events
    files = [f for f in os.listdir(tmp_path) if f.startswith("events.out.tfevents.")]
    assert len(files) == 1
    tags = {t for (_, _, t, _) in read_events(os.path.join(str(tmp_path), files[0]))}
    assert {"delay/episode_mean", "reward/jain", "queue/mec_cycles"} <= tags
