"""Cost of stream markers between back-to-back rollout kernels (rocrand mode: no upload involved)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import ml2048_amd as amd
from ml2048_amd import _lib
hip = C.CDLL("libamdhip64.so")
env = amd.VecGame(65536, spawn="rocrand")
env.reset(1)
L = env._L
M, N, dev = 65536, 128, env.device
W = int(L.ml2048_rollout_flag_words(M))
flags = torch.empty((N, W), dtype=torch.int64, device=dev)
last = torch.empty((M,), dtype=torch.int32, device=dev)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
def launch(k):
    _lib.check(L.ml2048_rollout_fast(env._stream(False), M, 0, env._st(), N, 0.8, 1, k * N, 0, 0, 7, k * N, None, None,
                                     flags.data_ptr(), last.data_ptr()), "x")
def make_hip_event(flags_):
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), C.c_uint(flags_)) == 0
    return e
hipEventDisableTiming, hipEventReleaseToDevice = 0x2, 0x40000000
variants = {}
def v_none(k): launch(k)
ev_t = torch.cuda.Event()
def v_torch_record(k): launch(k); ev_t.record(main)
def v_torch_record_wait(k): launch(k); ev_t.record(main); side.wait_event(ev_t)
ev2 = torch.cuda.Event()
def v_torch_cross(k):  # what rollout() does: wait on an event of another stream, launch, record
    ev2.record(side); main.wait_event(ev2); launch(k); ev_t.record(main); side.wait_event(ev_t)
e_sys = make_hip_event(hipEventDisableTiming)
e_dev = make_hip_event(hipEventDisableTiming | hipEventReleaseToDevice)
e_sys2 = make_hip_event(hipEventDisableTiming)
e_dev2 = make_hip_event(hipEventDisableTiming | hipEventReleaseToDevice)
ms, ss = C.c_void_p(main.cuda_stream), C.c_void_p(side.cuda_stream)
def v_hip_sys_cross(k):
    hip.hipEventRecord(e_sys2, ss); hip.hipStreamWaitEvent(ms, e_sys2, 0); launch(k); hip.hipEventRecord(e_sys, ms); hip.hipStreamWaitEvent(ss, e_sys, 0)
def v_hip_dev_cross(k):
    hip.hipEventRecord(e_dev2, ss); hip.hipStreamWaitEvent(ms, e_dev2, 0); launch(k); hip.hipEventRecord(e_dev, ms); hip.hipStreamWaitEvent(ss, e_dev, 0)
for name, fn in (("none", v_none), ("torch record", v_torch_record), ("torch record+wait", v_torch_record_wait),
                 ("torch cross (as rollout)", v_torch_cross), ("hip sys-scope cross", v_hip_sys_cross),
                 ("hip device-scope cross", v_hip_dev_cross), ("none", v_none)):
    for k in range(10): fn(k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(100): fn(k)
    torch.cuda.synchronize()
    print("%-28s %.1f us per launch" % (name, (time.perf_counter() - t) / 100 * 1e6), flush=True)
