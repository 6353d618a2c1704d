"""The native PCG64 schedule (ml2048_amd/csrc/hostrng.cpp) against numpy's Generator itself and
against the reference-derived golden schedule.  No GPU needed (host code only)."""
import os

import numpy as np
import pytest

from ml2048_amd.hostrng import HostRng


@pytest.mark.parametrize("seed", [0, 1, 12345, 20241008, 2**63 + 17])
def test_native_schedule_equals_numpy_draws(seed):
    a, b = HostRng(seed), HostRng(seed)
    for n in (1, 2, 7, 128, 300, 1500):
        sched, epochs, masks = a.draw_schedule(n, 0.8)
        s2, e2, m2 = b._draw_schedule_py(n, 0.8)
        assert np.array_equal(sched, s2), (seed, n)
        assert np.array_equal(epochs, e2) and np.array_equal(masks, m2)
        assert a.rand_step == b.rand_step and a.epoch == b.epoch
        assert np.array_equal(a.randperm, b.randperm) and np.array_equal(a.randfloat, b.randfloat)
        assert a.rand.bit_generator.state == b.rand.bit_generator.state
        # and the numpy generator continues identically after the native call (mixed use)
        assert a.draw_prepare() == b.draw_prepare()
        assert a.draw_step() == b.draw_step()
        assert a.rand.random() == b.rand.random()
    assert a.epoch > 100  # many table regenerations were crossed


def test_native_schedule_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "rng_schedule.npz"))
    for seed in (0, 12345, 20241008):
        rng = HostRng(seed)
        gs = g[f"s{seed}_sched"]
        sched, epochs, masks = rng.draw_schedule(gs.shape[0], 0.8)
        assert np.array_equal(sched[:, 1], gs[:, 1]) and np.array_equal(sched[:, 2], gs[:, 2])
        regen = np.diff(np.concatenate([[0], sched[:, 0]]))
        assert np.array_equal(regen, gs[:, 3])
        assert np.array_equal(epochs[0], g[f"s{seed}_perm0"])
        assert np.array_equal(rng.randperm, g[f"s{seed}_perm_last"])
        assert np.array_equal(rng.randfloat, g[f"s{seed}_flt_last"])


def test_rand_step_wraps_at_1024():
    """rand_step >= 1024 forces a regeneration (game_numba.py:622) even when u < 0.9."""
    a, b = HostRng(3), HostRng(3)
    a.rand_step = b.rand_step = 1020
    s1, e1, _ = a.draw_schedule(10, 0.8)
    s2, e2, _ = b._draw_schedule_py(10, 0.8)
    assert np.array_equal(s1, s2) and np.array_equal(e1, e2) and a.rand_step == b.rand_step


def test_two_prob_threshold_in_float64():
    """H2: float32 table value vs float64 two_prob compares in float64."""
    r = HostRng(5)
    x = np.float32(0.8)  # the smallest float32 >= 0.8 is > 0.8
    r.randfloat[:16] = x
    assert r.two_mask(0.8) == 0 and r.two_mask(float(x) + 1e-12) == 0xFFFF
    _, _, masks = r.draw_schedule(1, 0.8)
    assert masks[0] == 0


@pytest.mark.parametrize("threads,margin", [("1", "2500"), ("2", "0"), ("3", "40"), ("8", "0"), ("8", "2500"), ("16", "300")])
def test_speculative_scan_is_exact_for_any_thread_count_and_overlap(threads, margin):
    """The parallel row scan only guesses WHERE rows start; whatever the guesses and however little
    the threads' segments overlap (margin 0 => the lists rarely coalesce and the serial continuation
    takes over), the schedule must equal numpy's.  Run in a subprocess: the settings are read once."""
    import subprocess
    import sys

    code = (
        "import numpy as np\n"
        "from ml2048_amd.hostrng import HostRng\n"
        "a, b = HostRng(77), HostRng(77)\n"
        "for n in (128, 5, 700):\n"
        "    s1, e1, m1 = a.draw_schedule(n, 0.8); s2, e2, m2 = b._draw_schedule_py(n, 0.8)\n"
        "    assert np.array_equal(s1, s2) and np.array_equal(e1, e2) and np.array_equal(m1, m2)\n"
        "    assert a.rand.bit_generator.state == b.rand.bit_generator.state\n"
        "assert a.epoch > 60\n"
        "print('ok')\n"
    )
    env = dict(os.environ, ML2048_HOST_THREADS=threads, ML2048_HOSTRNG_MARGIN=margin, ML2048_HOSTRNG_SPEC="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
