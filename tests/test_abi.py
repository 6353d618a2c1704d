"""C-ABI checks that need no GPU: the library builds for gfx950, loads, exports every symbol
include/ml2048_hip.h declares, and validates arguments before touching the device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ml2048_amd import _lib, build

    build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ml2048_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ml2048_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    from ml2048_amd import _lib

    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ml2048_hip.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(_lib.EXPORTS) == declared


def test_version_and_sizes(lib):
    assert lib.ml2048_abi_version() == 1
    assert lib.ml2048_scratch_ints(1) == 4
    assert lib.ml2048_scratch_ints(65536) == 2 * 64 + 2
    assert lib.ml2048_rollout_flag_words(65536) == 1024
    assert lib.ml2048_rollout_flag_words(65) == 4
    assert lib.ml2048_rollout_scratch_ints(65536, 128) == 128 * 1024 + 128 + 2


def test_argument_validation_without_gpu(lib):
    from ml2048_amd import _lib

    st = _lib.State(*([0] * 11))
    assert lib.ml2048_reset(None, 0, C.byref(st)) == -1
    assert b"positive" in lib.ml2048_last_error()
    assert lib.ml2048_reset(None, 16, C.byref(st)) == -1
    assert b"NULL" in lib.ml2048_last_error()
    st2 = _lib.State(*([4096 + 8] + [4096] * 10))  # board not 16-byte aligned
    assert lib.ml2048_reset(None, 16, C.byref(st2)) == -2
    assert b"aligned" in lib.ml2048_last_error()
    with pytest.raises(ValueError):
        _lib.check(lib.ml2048_sample_actions(None, -5, 0, None, 0, 0, 1, None, None), "sample")
    # the later entry points reject bad arguments before any launch as well
    ok = _lib.State(*([4096] * 11))
    assert lib.ml2048_upload(None, None, None, 0, 0) == -1
    assert lib.ml2048_upload(None, 4096, 4104, 64, 0) == -2              # src not 16-byte aligned
    assert lib.ml2048_upload(None, 4096, 8192, 64, 80) == -1             # tables beyond the end
    assert lib.ml2048_upload(None, 4096, 8192, 64, 24) == -1             # table section not in 16-byte rows
    assert lib.ml2048_rollout_chain(None, 64, 0, C.byref(ok), 8, None, None, 1, None, 0, 0, 0, 0, None, None,
                                    None, None, 0, 0) == -1
    assert lib.ml2048_rollout_chain(None, 64, 0, C.byref(ok), 8, 4096, 4096, 1, 4096, 0, 0, 0, 0, 4096, 4096,
                                    8192, None, 64, 0) == -1              # next_dst missing
    assert lib.ml2048_episode_stats(None, 64, C.byref(ok), None, None, 0, 10, None, None, None) == -1
    assert lib.ml2048_episode_stats(None, 64, C.byref(ok), 4096, 4096, 0, -1, 4096, 4096, 4096) == -1
    assert lib.ml2048_episode_stats(None, 64, C.byref(ok), 4100, 4096, 0, 5, 4096, 4096, 4096) == -2   # age alignment
    assert lib.ml2048_encoder_front(None, 0, None, None, None, None, None, None, None, 0.01, None, None, None) == -1
    assert lib.ml2048_encoder_front(None, 8, 4100, 4096, 4096, 4096, 4096, 4096, 4096, 0.01, 4096, 4096, 4096) == -2
    assert lib.ml2048_timer_arm(None) == -1 and lib.ml2048_timer_elapsed_ms(None, None) == -1
    lib.ml2048_timer_destroy(None)
    assert lib.ml2048_sched_stream_take(None, None, 0, None, None, None, None, None, None, None, None) == -1
    assert lib.ml2048_sched_stream_take_slot(None, None, None, None, None, None, None, None, None, None) == -1
    # host-side plumbing of the pipelined rollout: NULL handles are rejected, nothing touches a device
    assert lib.ml2048_event_create(None) == -1 and lib.ml2048_event_record(None, None) == -1
    assert lib.ml2048_event_query(None) == -1 and lib.ml2048_event_synchronize(None) == -1
    assert lib.ml2048_stream_wait_event(None, None) == -1
    lib.ml2048_event_destroy(None)
    assert lib.ml2048_rollout_attach_event(None) == -1 and lib.ml2048_rollout_event_attached() == 0
    assert lib.ml2048_rollout_chain_ids(None, 64, 0, C.byref(ok), 8, 4096, 4096, 1, 4096, 0, 0, 0, 0, 4096, 4096, None, None,
                                        0, 0, None, 8192, 4096, 4096, 4096, None) == -1   # side-stream scan without a marker
    assert b"rollout_event" in lib.ml2048_last_error()
    # round 3: the line table and the late id scan reject bad arguments before any launch too
    assert lib.ml2048_row_lut_fill(None, None) == -2 and lib.ml2048_row_lut_fill(None, 4100) == -2
    assert b"16-byte aligned" in lib.ml2048_last_error()
    assert lib.ml2048_rollout_attach_row_lut(4104) == 0                  # (one-shot: consumed - and refused - by the next launch)
    assert lib.ml2048_rollout_chain(None, 64, 0, C.byref(ok), 8, 4096, 4096, 1, 4096, 0, 0, 0, 0, 4096, 4096,
                                    None, None, 0, 0) == -2 and b"row_lut" in lib.ml2048_last_error()
    assert lib.ml2048_rollout_ids_after(None, None, 64, C.byref(ok), 8, None, None, None, None, None, None) == -1
    assert lib.ml2048_rollout_ids_after(None, None, -1, C.byref(ok), 8, 4096, 4096, 4096, 4096, 4096, None) == -1
    assert lib.ml2048_sched_stream_slot_bytes(0) == 0
    assert lib.ml2048_sched_stream_slot_bytes(128) == 1536 + 528 + 129 * 1024 * 16
    assert not lib.ml2048_sched_stream_create_pinned(None, None, None, 0, 128, 0.8, 4, None, 0)
    # the one-launch step: one argument struct, checked field by field before any launch
    assert lib.ml2048_step_prepare(None, None) == -1
    A = _lib.StepPrepareArgs()
    A.M, A.st = 64, ok
    assert lib.ml2048_step_prepare(None, C.byref(A)) == -1 and b"actions" in lib.ml2048_last_error()
    A.actions, A.action_dtype, A.reward_kind = 4096, 0, 7
    assert lib.ml2048_step_prepare(None, C.byref(A)) == -1 and b"reward_kind" in lib.ml2048_last_error()
    A.reward_kind, A.randperm_step = 1, 4096
    assert lib.ml2048_step_prepare(None, C.byref(A)) == -1 and b"randperm_prepare" in lib.ml2048_last_error()
    A.randperm_prepare, A.two_mask_step = 4096, 1 << 16
    assert lib.ml2048_step_prepare(None, C.byref(A)) == -1 and b"two_mask" in lib.ml2048_last_error()
    A.two_mask_step = 0xFFFF
    assert lib.ml2048_step_prepare(None, C.byref(A)) == -1 and b"scratch" in lib.ml2048_last_error()
    A.game_count = A.n_reset = 4096
    A.scratch = 4100
    assert lib.ml2048_step_prepare(None, C.byref(A)) == -2                # scratch alignment
    A.scratch = 4096
    A.onehot = 4104
    assert lib.ml2048_step_prepare(None, C.byref(A)) == -2                # onehot alignment
    A.onehot, A.next_logp = None, 4096
    assert lib.ml2048_step_prepare(None, C.byref(A)) == -1 and b"next_logp" in lib.ml2048_last_error()
    assert lib.ml2048_step_prepare_scratch_bytes() >= 1024 * 8 + 4


def test_python_api_fails_loudly_without_gpu():
    import torch

    import ml2048_amd

    with pytest.raises(ValueError):
        ml2048_amd.VecGame(0)
    if not torch.cuda.is_available():
        with pytest.raises(ml2048_amd._lib.Ml2048Error):
            ml2048_amd.VecGame(8)


def test_missing_library_is_an_error(tmp_path):
    from ml2048_amd import _lib

    with pytest.raises(_lib.Ml2048Error):
        _lib.load(str(tmp_path / "nope.so"))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ml2048_amd/ may reference it."""
    pkg = os.path.join(ROOT, "ml2048_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn), encoding="utf-8", errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), fn
                assert "ml2048_oracle" not in text and "libml2048_oracle" not in text, fn


def test_reward_sentinels_match_known_answer(golden_dir):
    import json

    import ml2048_amd as m

    ka = json.load(open(os.path.join(golden_dir, "known_answer.json")))
    s, p, mg = ka["state"], ka["prev_state"], ka["merged"]
    assert m.reward_fn_normal(s, p, mg) == 6144.0
    assert m.reward_fn_rank(s, p, mg) == 42.0
    assert m.reward_fn_maxcell(s, p, mg) == 2052.0
    assert m.reward_fn_improved(s, p, mg) == ka["reward_fn_improved"]
    assert m.ITEM_VALUES.tolist() == [0] + [2**k for k in range(1, 18)]
    assert (m.STEP_LEFT, m.STEP_RIGHT, m.STEP_UP, m.STEP_DOWN) == (0, 1, 2, 3)
    assert m.RECORD_DTYPE.itemsize == 64 and m.RECORD_DTYPE.fields["board"][1] == 16


def test_host_rng_matches_golden(golden_dir):
    """The product's host RNG schedule (not the oracle's copy) against the reference fixtures."""
    import numpy as np

    from ml2048_amd.hostrng import HostRng

    g = np.load(os.path.join(golden_dir, "rng_schedule.npz"))
    for seed in (0, 12345, 20241008):
        rng = HostRng(seed)
        assert np.array_equal(rng.randperm, g[f"s{seed}_perm0"])
        assert np.array_equal(rng.randfloat, g[f"s{seed}_flt0"])
        sched = g[f"s{seed}_sched"]
        for k in range(sched.shape[0]):
            regen, base_p = rng.draw_prepare()
            assert int(regen) == sched[k, 3] and rng.rand_step == sched[k, 0] and base_p == sched[k, 1]
            assert rng.draw_step() == sched[k, 2]
        assert np.array_equal(rng.randperm, g[f"s{seed}_perm_last"])
        assert np.array_equal(rng.randfloat, g[f"s{seed}_flt_last"])
        m = rng.two_mask(0.8)
        assert m == sum(1 << c for c in range(16) if float(rng.randfloat[c]) < 0.8)
