"""The pipelined host schedule stream under CPU oversubscription (a completion count tied to a slot instead
of a schedule index could strand the consumer after a badly timed preemption).  Host code only; the ThreadSanitizer
build runs on the CPU because GPU sanitizers are not available on the pool."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "stream_stress.cpp"), os.path.join(ROOT, "ml2048_amd", "csrc", "hostrng.cpp")]


def _build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-mssse3", "-pthread", *flags, *SRC, "-o", exe])
    return exe


def _run(exe, *args, timeout=240):
    env = dict(os.environ, ML2048_HOST_THREADS="16", TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1")
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and "stream stress ok" in r.stdout, (r.stdout[-800:], r.stderr[-4000:])


def test_stream_oversubscribed_matches_sync(tmp_path):
    """8+ workers, a stitcher, a jittery consumer and one busy thread per processor: 6000 short schedules, every one
    equal to the synchronous call's; must not hang."""
    exe = _build(tmp_path, "stream_stress", ["-O2", "-fopenmp"])
    _run(exe, 6000, 8)
    _run(exe, 300, 128)
    _run(exe, 3000, 8, os.cpu_count() or 8, "slots")   # the zero-copy consumer: several schedules held in caller memory
    _run(exe, 300, 128, os.cpu_count() or 8, "slots")


def test_stream_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "stream_stress_tsan", ["-O1", "-g", "-fsanitize=thread"])
    _run(exe, 600, 8, 4)
    _run(exe, 600, 8, 4, "slots")
