"""HostRng.next_step_prepare / sync (the draws a step_prepare loop takes ahead, ml2048_amd/hostrng.py) against the plain
numpy-order draws of game_numba.py:622-626, :670 - same values, same regenerated tables and 2/4 masks, and after
sync() the same generator state, counter and tables, wherever the loop is interrupted.  Host code only."""
import numpy as np
import pytest

from ml2048_amd.hostrng import HostRng


@pytest.mark.parametrize("seed", [0, 1, 12345, 20241008])
def test_lookahead_equals_plain_draws(seed):
    a, b = HostRng(seed), HostRng(seed)
    rng = np.random.default_rng(seed + 1)
    assert a.draw_prepare() == b.draw_prepare()
    regens = 0
    for it in range(2500):
        row_s, row_p, e_s, e_p, blk = a.next_step_prepare(0.8)
        assert np.array_equal(blk[2][e_s], b.randperm) and blk[7][e_s] == b.two_mask(0.8), it   # the step's tables
        assert row_s == b.draw_step(), it
        regen, row_p2 = b.draw_prepare()
        assert (e_p != e_s, row_p) == (regen, row_p2), it
        regens += int(regen)
        assert np.array_equal(blk[2][e_p], b.randperm) and blk[7][e_p] == b.two_mask(0.8), it   # the prepare's
        if rng.random() < 0.01:  # another API takes over: exactly the consumed position, then plain draws
            a.sync()
            assert a.rand_step == b.rand_step and a.rand.bit_generator.state == b.rand.bit_generator.state
            assert np.array_equal(a.randperm, b.randperm) and np.array_equal(a.randfloat, b.randfloat)
            for _ in range(int(rng.integers(0, 3))):
                assert a.draw_step() == b.draw_step()
                assert a.draw_prepare() == b.draw_prepare()
    assert regens > 100
    a.sync()
    a.sync()  # idempotent
    assert a.rand.bit_generator.state == b.rand.bit_generator.state and a.rand_step == b.rand_step
    assert a.rand.random() == b.rand.random()


def test_lookahead_sync_right_after_the_first_call_and_at_block_edges():
    for n in (1, 2, HostRng.LOOKAHEAD - 1, HostRng.LOOKAHEAD, HostRng.LOOKAHEAD + 1, 2 * HostRng.LOOKAHEAD + 3):
        a, b = HostRng(9), HostRng(9)
        a.draw_prepare()
        b.draw_prepare()
        for _ in range(n):
            a.next_step_prepare(0.8)
            b.draw_step()
            b.draw_prepare()
        a.sync()
        assert a.rand.bit_generator.state == b.rand.bit_generator.state and a.rand_step == b.rand_step, n
        assert np.array_equal(a.randperm, b.randperm) and np.array_equal(a.randfloat, b.randfloat), n
