"""Builds and runs the C++ host-mirror test (include/spalinalg.hpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_mirror")


def build():
    src = os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp")
    lib = os.path.join(ROOT, "spalinalg_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src,
                           "-o", EXE, "-L", lib, "-lspal_hip", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_mirror_host():
    build()
    out = subprocess.run([EXE, "host"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_gpu():
    build()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_host_code_under_asan_ubsan():
    """spal_host.cpp (validation, partition) and spal_synth.cpp (generators) compiled with g++
    -fsanitize=address,undefined and driven with edge-case / fuzzed inputs."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_sanitize")
    subprocess.check_call(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
         "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
         "-I", os.path.join(ROOT, "spal_synth"), os.path.join(ROOT, "spal_synth", "spal_synth.cpp"),
         os.path.join(ROOT, "spalinalg_amd", "csrc", "spal_host.cpp"),
         os.path.join(ROOT, "tests", "cpp", "test_host_sanitize.cpp"), "-o", exe, "-pthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr


def test_oracle_under_asan_ubsan():
    exe = os.path.join(ROOT, "tests", "cpp", "test_oracle_sanitize")
    subprocess.check_call(
        ["gcc", "-std=c11", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
         "-fno-omit-frame-pointer", os.path.join(ROOT, "oracle", "spal_oracle.c"),
         os.path.join(ROOT, "tests", "cpp", "test_oracle_sanitize.c"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True,
                         env=dict(os.environ, UBSAN_OPTIONS="halt_on_error=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr


C_EXE = os.path.join(ROOT, "tests", "c", "abi_demo")


def build_c():
    """include/spal.h is a C header: a C99 translation unit, -pedantic -Werror."""
    src = os.path.join(ROOT, "tests", "c", "abi_demo.c")
    lib = os.path.join(ROOT, "spalinalg_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I",
                           os.path.join(ROOT, "include"), src, "-o", C_EXE, "-L", lib, "-lspal_hip",
                           f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])


def test_plain_c_caller_host():
    build_c()
    out = subprocess.run([C_EXE], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "spalinalg_amd" in out.stdout and "devices:" in out.stdout


@pytest.mark.gpu
def test_plain_c_caller_gpu():
    build_c()
    out = subprocess.run([C_EXE, "gpu"], capture_output=True, text=True)
    assert out.returncode == 0 and "gpu ok" in out.stdout, out.stdout + out.stderr
