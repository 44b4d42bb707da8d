"""Builds the page manager + fake backend under ThreadSanitizer and runs a mixed API workload against the live mapper
thread (tests/native/tsan_driver.cpp).  Any data race makes TSan exit with code 66."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _skip_only_if_the_sanitizer_is_missing(stderr: str, what: str):
    """a toolchain without the sanitizer runtime skips; a compile or link error in OUR sources fails (round 6: a missing stub once hid
    both tests behind a skip)"""
    if "undefined reference" in stderr or "error:" in stderr.replace("ld returned 1 exit status", ""):
        if "tsan" not in stderr and "asan" not in stderr and "ubsan" not in stderr and "sanitizer" not in stderr.lower():
            pytest.fail(what + " build failed in the package's own sources:\n" + stderr[-1500:])
    pytest.skip(what + " build unavailable: " + stderr[-300:])


def test_mapper_thread_is_race_free_under_tsan():
    src = [os.path.join(ROOT, "vattention_amd/csrc/page_manager.cpp"), os.path.join(ROOT, "vattention_amd/csrc/capi.cpp"),
           os.path.join(ROOT, "tests/native/fake_backend.cpp"), os.path.join(ROOT, "tests/native/tsan_driver.cpp")]
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "tsan_driver")
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", *src, "-o", exe], capture_output=True, text=True)
        if r.returncode != 0:
            _skip_only_if_the_sanitizer_is_missing(r.stderr, "ThreadSanitizer")
        env = dict(os.environ, TSAN_OPTIONS="exitcode=66 halt_on_error=0")
        run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        if "FATAL: ThreadSanitizer" in run.stderr and "unexpected memory mapping" in run.stderr:
            pytest.skip("ThreadSanitizer cannot run in this container (ASLR layout)")
        assert "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-3000:]
        assert run.returncode == 0, (run.returncode, run.stdout[-500:], run.stderr[-1500:])
        assert "violations 0" in run.stdout


def test_page_manager_is_clean_under_asan_and_ubsan():
    """The same workload under AddressSanitizer + UndefinedBehaviorSanitizer + the leak checker: no out-of-bounds / use-after-free in the
    plan-then-execute containers, no signed overflow or bad shift in the u64 arithmetic (num_free_kvblocks wraps on purpose, as unsigned),
    nothing left allocated after cleanup."""
    src = [os.path.join(ROOT, "vattention_amd/csrc/page_manager.cpp"), os.path.join(ROOT, "vattention_amd/csrc/capi.cpp"),
           os.path.join(ROOT, "tests/native/fake_backend.cpp"), os.path.join(ROOT, "tests/native/tsan_driver.cpp")]
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "asan_driver")
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread", *src, "-o", exe],
                           capture_output=True, text=True)
        if r.returncode != 0:
            _skip_only_if_the_sanitizer_is_missing(r.stderr, "AddressSanitizer")
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1 exitcode=67", UBSAN_OPTIONS="print_stacktrace=1")
        run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        if "LeakSanitizer has encountered a fatal error" in run.stderr or "LeakSanitizer does not work under ptrace" in run.stderr:
            env["ASAN_OPTIONS"] = "detect_leaks=0 exitcode=67"       # (the leak checker needs ptrace; the rest still runs)
            run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
        assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error:" not in run.stderr and "ERROR: LeakSanitizer" not in run.stderr, run.stderr[-3000:]
        assert run.returncode == 0, (run.returncode, run.stdout[-500:], run.stderr[-1500:])
        assert "violations 0" in run.stdout
