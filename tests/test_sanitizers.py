"""Host-side hygiene: the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer, and the
C++ shim header compiled with -fsanitize (its run needs the GPU library, so only the oracle is executed here)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORA = os.path.join(ROOT, "oracle")


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    srcs = [os.path.join(ORA, f) for f in sorted(os.listdir(ORA)) if f.endswith(".c")]
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-ffp-contract=off", "-I" + ORA, os.path.join(ROOT, "tests", "oracle_san_test.c")]
    cmd += srcs + ["-o", exe, "-lm"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 and "cannot find" in r.stdout and "san" in r.stdout:
        pytest.skip("sanitizer runtime not installed: " + r.stdout[-300:])
    assert r.returncode == 0, r.stdout[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitizer run ok" in r.stdout, r.stdout[-4000:]


def test_shim_header_compiles_with_sanitizers_and_warnings(tmp_path):
    """The shim is a header: instantiate it (tests/shim_test.cpp) with -fsanitize=address,undefined -Wall -Wextra;
    linking needs liborbgpu.so, running needs the device, so this stops at the object file."""
    obj = str(tmp_path / "shim_test.o")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-Wall", "-Wextra", "-Werror",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "orb_slam2_map_amd", "shim"), "-c",
           os.path.join(ROOT, "tests", "shim_test.cpp"), "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
