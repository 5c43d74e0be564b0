"""The example script runs end to end on a GPU box (small batch)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rollout_example_runs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "rollout.py"), "256", "1"], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "env-steps/s" in out.stdout and "episode 0" in out.stdout
