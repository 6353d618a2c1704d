"""The asynchronous schedule stream (native producer thread) against the synchronous native
schedule: same schedules in the same order, exact after-states.  Host code only (no GPU)."""
import ctypes as C

import numpy as np

from ml2048_amd import _lib
from ml2048_amd.hostrng import HostRng


def _state_of(rng):
    st = rng.rand.bit_generator.state
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return _lib.Pcg64State(s >> 64, s & m, inc >> 64, inc & m, int(st["has_uint32"]), int(st["uinteger"]))


def test_stream_delivers_the_same_schedules_in_order():
    L = _lib.load()
    N = 64
    a, ref = HostRng(2024), HostRng(2024)
    c = _state_of(a)
    h = L.ml2048_sched_stream_create(C.byref(c), a.randperm.ctypes.data, a.randfloat.ctypes.data, a.rand_step, N, 0.8, 3)
    assert h
    try:
        for k in range(25):
            p_s, p_e, p_m, p_rp, p_rf = (C.c_void_p() for _ in range(5))
            ne, rs = C.c_int32(0), C.c_int32(0)
            after = _lib.Pcg64State()
            assert L.ml2048_sched_stream_next(h, C.byref(p_s), C.byref(p_e), C.byref(p_m), C.byref(ne), C.byref(after),
                                              C.byref(rs), C.byref(p_rp), C.byref(p_rf)) == 0
            s2, e2, m2 = ref.draw_schedule(N, 0.8)
            E = ne.value
            assert E == e2.shape[0]
            sched = np.frombuffer((C.c_uint8 * (N * 12)).from_address(p_s.value), np.int32).reshape(N, 3)
            epochs = np.frombuffer((C.c_uint8 * (E * 16384)).from_address(p_e.value), np.uint8).reshape(E, 1024, 16)
            masks = np.frombuffer((C.c_uint8 * (E * 4)).from_address(p_m.value), np.uint32)
            perm_after = np.frombuffer((C.c_uint8 * 16384).from_address(p_rp.value), np.uint8).reshape(1024, 16)
            flt_after = np.frombuffer((C.c_uint8 * 4096).from_address(p_rf.value), np.float32)
            assert np.array_equal(sched, s2) and np.array_equal(epochs, e2) and np.array_equal(masks, m2), k
            assert rs.value == ref.rand_step
            assert np.array_equal(perm_after, ref.randperm) and np.array_equal(flt_after, ref.randfloat)
            st = ref.rand.bit_generator.state
            assert ((int(after.state_hi) << 64) | int(after.state_lo)) == st["state"]["state"]
            assert int(after.has_uint32) == st["has_uint32"] and int(after.uinteger) == st["uinteger"]
            assert L.ml2048_sched_stream_release(h) == 0
    finally:
        L.ml2048_sched_stream_destroy(h)


def test_stream_create_validates_and_destroy_is_idempotent_on_null():
    L = _lib.load()
    assert not L.ml2048_sched_stream_create(None, None, None, 0, 8, 0.8, 2)
    L.ml2048_sched_stream_destroy(None)
    assert L.ml2048_sched_stream_release(None) == -1


def _check_slot(buf, N, E, o_mask, o_tab, nbytes, ref_draw):
    s2, e2, m2 = ref_draw
    assert E == e2.shape[0] and o_mask == (N * 12 + 15) // 16 * 16 and o_tab == o_mask + ((N + 1) * 4 + 15) // 16 * 16
    assert nbytes == o_tab + E * 16384
    assert np.array_equal(buf[:N * 12].view(np.int32).reshape(N, 3), s2)
    assert not buf[N * 12:o_mask].any()                       # padding is defined (it is copied to the device)
    assert np.array_equal(buf[o_mask:o_mask + E * 4].view(np.uint32), m2)
    assert np.array_equal(buf[o_tab:nbytes].reshape(E, 1024, 16), e2)


def test_take_slot_hands_out_caller_memory_in_place_and_holds_several():
    """Zero-copy hand-over (VecGame's pinned slots): the producer writes [schedule | masks | tables] straight into
    caller memory; several schedules are held at once and released oldest first; a held slot is never rewritten."""
    L = _lib.load()
    N, depth = 48, 5
    sb = int(L.ml2048_sched_stream_slot_bytes(N))
    mem = np.zeros((depth, sb + 16), np.uint8)
    base = [(m.ctypes.data + 15) // 16 * 16 for m in mem]
    views = [m[b - m.ctypes.data:b - m.ctypes.data + sb] for m, b in zip(mem, base)]
    a, ref = HostRng(7), HostRng(7)
    c = _state_of(a)
    ptrs = (C.c_void_p * depth)(*base)
    assert not L.ml2048_sched_stream_create_pinned(C.byref(c), a.randperm.ctypes.data, a.randfloat.ctypes.data,
                                                   a.rand_step, N, 0.8, depth, ptrs, sb - 1)      # slots too small
    h = L.ml2048_sched_stream_create_pinned(C.byref(c), a.randperm.ctypes.data, a.randfloat.ctypes.data, a.rand_step,
                                            N, 0.8, depth, ptrs, sb)
    assert h
    held = []   # (slot index, snapshot of the slot, layout, draw of the reference)
    perm_after, flt_after = np.empty((1024, 16), np.uint8), np.empty((1024,), np.float32)
    try:
        for k in range(40):
            slot, ne, rs = C.c_int32(-1), C.c_int32(0), C.c_int32(0)
            o_m, o_t, nb = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            after = _lib.Pcg64State()
            rc = L.ml2048_sched_stream_take_slot(h, C.byref(slot), C.byref(ne), C.byref(o_m), C.byref(o_t), C.byref(nb),
                                                 C.byref(after), C.byref(rs), perm_after.ctypes.data, flt_after.ctypes.data)
            if len(held) == depth:
                assert rc == -3, "every slot is held: nothing to hand out"
                # ... and the held ones still hold what was handed out, the producer notwithstanding
                for i, lay, draw in held:
                    _check_slot(views[i], N, *lay, draw)
                for _ in range(3):
                    held.pop(0)
                    assert L.ml2048_sched_stream_release(h) == 0
                continue
            assert rc == 0
            draw = ref.draw_schedule(N, 0.8)
            lay = (ne.value, o_m.value, o_t.value, nb.value)
            _check_slot(views[slot.value], N, *lay, draw)
            assert rs.value == ref.rand_step
            assert np.array_equal(perm_after, ref.randperm) and np.array_equal(flt_after, ref.randfloat)
            st = ref.rand.bit_generator.state
            assert ((int(after.state_hi) << 64) | int(after.state_lo)) == st["state"]["state"]
            assert all(slot.value != i for i, _, _ in held), "a held slot was handed out again"
            held.append((slot.value, lay, draw))
    finally:
        L.ml2048_sched_stream_destroy(h)


def test_take_copies_the_slot_layout():
    L = _lib.load()
    N = 33
    a, ref = HostRng(99), HostRng(99)
    c = _state_of(a)
    h = L.ml2048_sched_stream_create(C.byref(c), a.randperm.ctypes.data, a.randfloat.ctypes.data, a.rand_step, N, 0.8, 2)
    assert h
    k, e = C.c_int32(0), C.c_int32(0)
    q = [C.c_int64(0) for _ in range(3)]
    assert L.ml2048_sched_stream_take_slot(h, C.byref(k), C.byref(e), C.byref(q[0]), C.byref(q[1]), C.byref(q[2]), None, None,
                                           None, None) == -3   # this stream owns its slots: nothing to hand out in place
    dst = np.zeros(int(L.ml2048_sched_stream_slot_bytes(N)) + 16, np.uint8)
    off = (-dst.ctypes.data) % 16
    try:
        for k in range(6):
            ne, rs = C.c_int32(0), C.c_int32(0)
            o_m, o_t, nb = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            after = _lib.Pcg64State()
            if k == 2:   # too small a destination: nothing consumed, the need reported
                assert L.ml2048_sched_stream_take(h, dst.ctypes.data + off, 64, C.byref(ne), C.byref(o_m), C.byref(o_t),
                                                  C.byref(nb), C.byref(after), C.byref(rs), None, None) == -3
                assert nb.value > 64
            assert L.ml2048_sched_stream_take(h, dst.ctypes.data + off, dst.size - off, C.byref(ne), C.byref(o_m),
                                              C.byref(o_t), C.byref(nb), C.byref(after), C.byref(rs), None, None) == 0
            _check_slot(dst[off:], N, ne.value, o_m.value, o_t.value, nb.value, ref.draw_schedule(N, 0.8))
            assert rs.value == ref.rand_step
    finally:
        L.ml2048_sched_stream_destroy(h)


def test_stream_with_the_scalar_generator_too():
    """The stream's PCG64 outputs come from an AVX-512 routine where the CPU has it (csrc/hostrng.cpp gen_outputs_avx512);
    ML2048_HOSTRNG_AVX512=0 selects the scalar four-lane form.  The choice is made once per process, so the scalar form is
    checked in a child process: the same stream-vs-synchronous comparison as above."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, ML2048_HOSTRNG_AVX512="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k", "delivers_the_same_schedules or after_states"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
