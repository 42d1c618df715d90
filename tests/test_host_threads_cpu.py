"""The library's host-side thread helpers (csrc/host_threads.hpp: the pool of the MSM tails with its early wake-up, the uploader thread of
host-fed MSMs) under ThreadSanitizer, without a GPU: tests/abi/host_threads_stress.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_threads_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "host_threads_stress")
    src = os.path.join(ROOT, "tests", "abi", "host_threads_stress.cpp")
    inc = os.path.join(ROOT, "zkp-implementation_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I", inc, src, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env={**os.environ, "TSAN_OPTIONS": "halt_on_error=0:second_deadlock_stack=1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 failures" in r.stdout
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
