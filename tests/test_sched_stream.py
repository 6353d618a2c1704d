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
