"""Oracle (C) vs device arithmetic (board_ops.hpp compiled for the host) under ASan + UBSan.
GPU AddressSanitizer is not available on the pool, so the sanitizers run on the CPU builds."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_oracle_and_device_arithmetic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_main")
    obj = str(tmp_path / "oracle.o")
    flags = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
    subprocess.check_call(["gcc", "-std=c11", "-ffp-contract=off", *flags, "-c",
                           os.path.join(HERE, "..", "oracle", "ml2048_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-std=c++17", "-Wno-unknown-pragmas", *flags,
                           os.path.join(HERE, "csrc", "sanitize_main.cpp"), obj, "-o", exe, "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "sanitize ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
