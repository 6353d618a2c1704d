"""Property tests (hypothesis) of the game rules on the CPU oracle and on the device arithmetic
compiled for the host - the invariants SURVEY.md 4/8a lists: valid mask <=> the move changes the
board, forward-movable <=> the push changes the line, spawn lands on an empty cell, tile-count
conservation, merged buckets vs reward, idempotence of a push."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

HERE = os.path.dirname(os.path.abspath(__file__))
cells = st.integers(min_value=0, max_value=17)
sparse_cells = st.one_of(st.just(0), st.just(0), st.integers(1, 4), cells)
boards = st.lists(sparse_cells, min_size=16, max_size=16)
lines = st.lists(cells, min_size=4, max_size=4)


@settings(max_examples=400, deadline=None)
@given(lines)
def test_push_line_properties(oracle, line):
    out, buckets = oracle.push_row(line)
    fwd, bwd = oracle.line_movable(*line)
    assert fwd == (out.tolist() != list(line))                       # movable <=> push changes the line
    again, b2 = oracle.push_row(out.tolist())
    nz = [v for v in out if v]
    assert out.tolist() == nz + [0] * (4 - len(nz))                  # result is left-packed
    assert sum(1 for v in line if v) - int(buckets.sum()) == len(nz)  # each merge removes one tile
    assert sum(2 ** v for v in line if v) == sum(2 ** int(v) for v in out if v)  # tile mass is conserved
    rev_out, _ = oracle.push_row(list(line)[::-1])
    assert bwd == (rev_out.tolist() != list(line)[::-1])             # backward = forward on the mirror


@settings(max_examples=300, deadline=None)
@given(boards)
def test_valid_mask_iff_move_changes_board(oracle, board):
    valid, any_ = oracle.compute_valid_actions(np.array(board, np.uint8))
    for a in range(4):
        moved, merged = oracle.step_board(board, a)
        assert bool(valid[a]) == (moved.tolist() != list(board)), (board, a)
        assert sum(1 for v in board if v) - int(merged.sum()) == sum(1 for v in moved if v)
    assert any_ == bool(valid.any())


@settings(max_examples=200, deadline=None)
@given(boards, st.integers(0, 1023), st.integers(0, 2**31 - 1))
def test_spawn_lands_on_an_empty_cell_in_permutation_order(oracle, board, row, seed):
    rng = oracle.HostRng(seed % 1000)
    b = np.array(board, np.uint8)
    before = b.copy()
    ok = oracle.lib().orc_spawn2(rng.randperm.ctypes.data, rng.randfloat.ctypes.data, row, b.ctypes.data, 0.8, 1)
    empties = np.flatnonzero(before == 0)
    if empties.size == 0:
        assert not ok and np.array_equal(b, before)
        return
    changed = np.flatnonzero(b != before)
    assert ok and changed.size == 1 and before[changed[0]] == 0 and b[changed[0]] in (1, 2)
    first = next(c for c in rng.randperm[row] if before[c] == 0)     # first empty cell in the row's order
    assert changed[0] == first
    assert b[first] == (1 if float(rng.randfloat[first]) < 0.8 else 2)  # H1: float table indexed by the CELL


@pytest.fixture(scope="module")
def H():
    lib = os.path.join(HERE, "csrc", "libhost_ops_harness.so")
    src = os.path.join(HERE, "csrc", "host_ops_harness.cpp")
    if not os.path.exists(lib):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", lib, src])
    L = C.CDLL(lib)
    L.h_move.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.h_valid.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


@settings(max_examples=300, deadline=None)
@given(boards, st.integers(0, 3))
def test_device_arithmetic_agrees_with_oracle_on_arbitrary_boards(oracle, H, board, action):
    b = np.array([board], np.uint8)
    a = np.array([action], np.uint8)
    out = np.zeros((1, 16), np.uint8)
    merged = np.zeros((1, 16), np.uint8)
    stats = np.zeros((1, 3), np.uint32)
    H.h_move(b.ctypes.data, 1, a.ctypes.data, out.ctypes.data, merged.ctypes.data, stats.ctypes.data)
    ob, om = oracle.step_board(board, action)
    assert np.array_equal(out[0], ob) and np.array_equal(merged[0], om[:16])
    valid = np.zeros((1, 4), np.uint8)
    emp = np.zeros(1, np.uint16)
    mx = np.zeros(1, np.uint8)
    H.h_valid(b.ctypes.data, 1, valid.ctypes.data, emp.ctypes.data, mx.ctypes.data)
    assert np.array_equal(valid[0], oracle.compute_valid_actions(b[0])[0])
    assert mx[0] == max(board)
