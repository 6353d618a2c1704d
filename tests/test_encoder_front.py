"""Front end of the reference's CNN encoder: ml2048_encoder_front == leaky_relu of the three
depthwise convolutions of CNNEncoder.forward (src/ml2048/policy/_network.py:91-116) applied to the one-hot
board, computed from the 16-byte board.  Floating point: every output is bias + at most 16 weights; the
kernel adds them in ascending cell order, torch's convolution in its own order, so the comparison is to
rtol 1e-5 / atol 1e-6 (float32), not bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "these tests need the GPU box"
    return torch


def _reference_front(torch, board, dw_full, dw_rows, dw_cols):
    """The first half of CNNEncoder.forward, with torch's own convolutions on the one-hot."""
    F = torch.nn.functional
    x = F.one_hot(board.long(), 16).float().permute(0, 2, 1)           # (N, 16 classes, 16 cells)   :91-95
    full = F.leaky_relu(dw_full(x)).flatten(1)                          # (N, 256)                    :98-100
    grid = x.reshape(-1, 16, 4, 4)                                      #                             :105
    rows = F.leaky_relu(dw_rows(grid)).reshape(-1, 256, 4)              # (N, 256, 4, 1)              :108-110
    cols = F.leaky_relu(dw_cols(grid)).reshape(-1, 256, 4)              # (N, 256, 1, 4)              :114-116
    return full, rows, cols


@pytest.mark.parametrize("N", [1, 5, 64, 1000, 4099])
def test_encoder_front_matches_torch_convolutions(torch_mod, N):
    torch = torch_mod
    from ml2048_amd import train_ops

    g = torch.Generator().manual_seed(N)
    dev = torch.device("cuda:0")
    dw_full = torch.nn.Conv1d(16, 256, 16, groups=16).to(dev)
    dw_rows = torch.nn.Conv2d(16, 256, (1, 4), groups=16).to(dev)
    dw_cols = torch.nn.Conv2d(16, 256, (4, 1), groups=16).to(dev)
    with torch.no_grad():
        for m in (dw_full, dw_rows, dw_cols):    # non-trivial biases (the reference initialises them to zero)
            m.bias.copy_(torch.randn(256, generator=g) * 0.1)
    board = torch.randint(0, 16, (N, 16), generator=g, dtype=torch.uint8)
    board[0] = 0                                   # all cells one class
    if N > 4:
        board[1] = torch.arange(16, dtype=torch.uint8)
        board[2] = 15
        board[3, ::2] = 0
    board = board.to(dev)
    with torch.no_grad():
        want = _reference_front(torch, board, dw_full, dw_rows, dw_cols)
        got = train_ops.encoder_front(board, dw_full, dw_rows, dw_cols)
    for name, a, b in zip(("full", "rows", "cols"), got, want):
        assert a.shape == b.shape, (name, a.shape, b.shape)
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (name, float((a - b).abs().max()))


def test_encoder_front_on_real_game_boards_and_int8_input(torch_mod):
    """Boards of a running game (mostly empty cells, small exponents), passed as the int8 the replay buffers
    hold (replay.py:10-20), through the example's encoder modules."""
    torch = torch_mod
    import ml2048_amd as amd
    from ml2048_amd import train_ops

    env = amd.VecGame(2048)
    env.reset(3)
    env.rollout(60, masked=True, act_seed=1)
    board = env.board.to(torch.int8)
    dev = board.device
    dw_full = torch.nn.Conv1d(16, 256, 16, groups=16).to(dev)
    dw_rows = torch.nn.Conv2d(16, 256, (1, 4), groups=16).to(dev)
    dw_cols = torch.nn.Conv2d(16, 256, (4, 1), groups=16).to(dev)
    with torch.no_grad():
        want = _reference_front(torch, board.clamp(0, 15), dw_full, dw_rows, dw_cols)
        got = train_ops.encoder_front(board, dw_full, dw_rows, dw_cols)
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
