"""spawn="rocrand" is not bit-exact with numba (rocRAND cannot reproduce PCG64) - it has to be the same GAME:
`_spawn2` (src/ml2048/game_numba.py:172-212) places the new tile on a uniformly random EMPTY cell (a random permutation
row probed in order) and makes it a 2 with probability two_prob, else a 4.  These tests measure exactly that on
>= 1e6 spawns of the device generator, with the thresholds written out.

The pre-spawn board is recomputed on the host from the golden 18^4 line table (tests/golden/lines18.npz, produced by
the reference's own `_push_row`), so the spawned cell is "the one cell where the result differs from the pushed board".
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TWO_PROB = 0.8
SIGMAS = 5.0  # every threshold below is "within 5 standard deviations": false-alarm probability < 1e-6 per test


@pytest.fixture(scope="module")
def amd():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU box"
    import ml2048_amd

    ml2048_amd._lib.load()
    return ml2048_amd


def _moved(table, boards, actions):
    """Boards after the move alone (no spawn): every line through the reference's exhaustive push table."""
    b = boards.reshape(-1, 4, 4).astype(np.int64)
    out = np.empty(b.shape, np.uint8)
    for a in range(4):
        sel = actions == a
        x = b[sel]
        if a >= 2:
            x = x.transpose(0, 2, 1)   # up / down: columns are the lines (game_numba.py:107-122)
        if a % 2 == 1:
            x = x[:, :, ::-1]          # right / down: pushed toward the far end
        y = table[((x[..., 0] * 18 + x[..., 1]) * 18 + x[..., 2]) * 18 + x[..., 3]]
        if a % 2 == 1:
            y = y[:, :, ::-1]
        if a >= 2:
            y = y.transpose(0, 2, 1)
        out[sel] = y
    return out.reshape(-1, 16)


def _chi2_bound(dof):
    return dof + SIGMAS * np.sqrt(2.0 * dof)   # chi-square: mean dof, variance 2 dof


def test_step_spawn_is_uniform_over_empty_cells_and_two_with_prob_08(amd, golden_dir):
    table = np.load(os.path.join(golden_dir, "lines18.npz"))["result"]
    M = 65536
    env = amd.VecGame(M, "normal", spawn="rocrand", spawn_seed=0xD157, two_prob=TWO_PROB)
    env.reset(1)
    arng = np.random.default_rng(77)
    rank_counts = np.zeros((17, 16), np.int64)   # [number of empty cells before the spawn][rank of the chosen one]
    cell_two = np.zeros(16, np.int64)
    cell_all = np.zeros(16, np.int64)
    cols = np.arange(16)
    for t in range(26):
        env.prepare_torch()
        before = env.board.cpu().numpy()
        valid = env.valid_actions.cpu().numpy()
        a = arng.integers(0, 4, size=M)
        res = env.step(a)
        ok = res["invalid"] == 0
        assert np.array_equal(ok, valid[np.arange(M), a] != 0)
        after = res["state"][ok]
        moved = _moved(table, before[ok], a[ok])
        diff = after != moved
        assert np.all(diff.sum(axis=1) == 1), "exactly one cell differs from the pushed board: the spawned tile"
        cell = diff.argmax(axis=1)
        rows = np.arange(after.shape[0])
        assert np.all(moved[rows, cell] == 0), "the tile lands on an empty cell"
        val = after[rows, cell]
        assert np.all((val == 1) | (val == 2)), "a 2 or a 4 (exponent 1 or 2)"
        empty = moved == 0
        n_empty = empty.sum(axis=1)
        rank = (empty & (cols[None, :] < cell[:, None])).sum(axis=1)
        np.add.at(rank_counts, (n_empty, rank), 1)
        np.add.at(cell_all, cell, 1)
        np.add.at(cell_two, cell[val == 1], 1)
    n = int(cell_all.sum())
    assert n >= 1_000_000, n
    # (1) which empty cell: uniform over the n_empty candidates, for every n_empty that occurred often enough
    chi2 = dof = 0.0
    for k in range(2, 17):
        tot = rank_counts[k, :k].sum()
        if tot < 500 * k:
            continue
        exp = tot / k
        chi2 += float(((rank_counts[k, :k] - exp) ** 2 / exp).sum())
        dof += k - 1
    assert dof >= 60 and chi2 < _chi2_bound(dof), (chi2, dof)
    # (2) 2 vs 4: Binomial(n, two_prob)
    twos = int(cell_two.sum())
    assert abs(twos - TWO_PROB * n) < SIGMAS * np.sqrt(n * TWO_PROB * (1 - TWO_PROB)), (twos, n)
    # (3) ... independently of the cell (the table mode's cell-indexed randfloat, hazard H1, is a property of the
    #     reference's tables, not of the game's rules)
    z = (cell_two - TWO_PROB * cell_all) / np.sqrt(cell_all * TWO_PROB * (1 - TWO_PROB))
    assert np.all(np.abs(z) < SIGMAS + 0.5), z
    print(f"{n} spawns: chi2 {chi2:.1f} on {dof:.0f} dof (bound {_chi2_bound(dof):.1f}); P(2) = {twos / n:.5f}")


def test_deal_spawns_two_tiles_on_uniform_cells(amd):
    """The re-deal (game_numba.py:648-655, count=2 on an empty board): an unordered pair of distinct cells, every
    one of the 120 pairs equally likely; each tile a 2 with probability two_prob."""
    M, seeds = 65536, 8
    pair = np.zeros((16, 16), np.int64)
    twos = total = 0
    for s in range(seeds):
        env = amd.VecGame(M, "normal", spawn="rocrand", spawn_seed=1000 + s, two_prob=TWO_PROB)
        env.reset(0)
        env.prepare_torch()
        b = env.board.cpu().numpy()
        nz = b != 0
        assert np.all(nz.sum(axis=1) == 2) and np.all(b[nz] <= 2)
        cells = np.argsort(~nz, axis=1, kind="stable")[:, :2]   # the two occupied cells, ascending
        np.add.at(pair, (cells[:, 0], cells[:, 1]), 1)
        twos += int((b == 1).sum())
        total += 2 * M
    assert total >= 1_000_000
    iu = np.triu_indices(16, 1)
    obs = pair[iu]
    assert pair.sum() == obs.sum() == seeds * M
    exp = seeds * M / 120.0
    chi2 = float(((obs - exp) ** 2 / exp).sum())
    assert chi2 < _chi2_bound(119), chi2
    assert abs(twos - TWO_PROB * total) < SIGMAS * np.sqrt(total * TWO_PROB * (1 - TWO_PROB)), (twos, total)


def test_statistics_have_teeth(golden_dir):
    """The same estimators reject a spawn that prefers the first empty cell (no GPU involved)."""
    rng = np.random.default_rng(0)
    k, tot = 8, 200000
    biased = rng.choice(k, size=tot, p=np.array([1.1] + [1.0] * (k - 1)) / (k + 0.1))
    counts = np.bincount(biased, minlength=k)
    chi2 = float(((counts - tot / k) ** 2 / (tot / k)).sum())
    assert chi2 > _chi2_bound(k - 1)
