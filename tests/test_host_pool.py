"""The host worker pool of the library (csrc/host/pool.h: lock-free task claims, polling workers) under a stress program:
20 000 jobs of 2 .. 62 tasks, armed and cold, must run every task exactly once; the same program under ThreadSanitizer
must report nothing.  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "host_pool_stress.cpp")
INC = os.path.join(ROOT, "simpleworks_amd", "csrc")


def _build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-pthread", "-I", INC, SRC, "-o", exe] + flags)
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_every_task_runs_exactly_once(tmp_path):
    out = subprocess.run([_build(tmp_path, "pool", ["-O2"])], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK "), out.stdout + out.stderr


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_no_data_race_under_thread_sanitizer(tmp_path):
    try:
        exe = _build(tmp_path, "pool_tsan", ["-O1", "-g", "-fsanitize=thread"])
    except subprocess.CalledProcessError:
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "WARNING: ThreadSanitizer" not in out.stderr and out.stdout.startswith("OK "), out.stdout + out.stderr[-3000:]
