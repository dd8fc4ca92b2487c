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
