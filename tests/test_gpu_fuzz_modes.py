"""A short run of the mode-interleaving soak (tools/fuzz_modes.py: four envs - host / host-async / device schedule and a
per-step-only one - through random sequences of fused rollouts, per-step pairs, one-launch steps, resets and
generator hand-backs; all four must stay bit-identical) as part of the GPU test run."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_mode_interleaving_soak_20s():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_modes.py"), "20"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
