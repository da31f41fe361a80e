"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds (never on the GPU): the oracle and the host builds
of the kernel sources, which run mpc_wave.hpp / mpc_ltv.hpp / mpc_preamble.hpp themselves with bounds-checked LDS
accesses (tests/host_wave_ctx.hpp) - see tests/san_run.py, which this test runs in a child process with libasan
preloaded (a sanitized shared library cannot be loaded into a plain python otherwise)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_cpu_builds_are_clean_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if asan is None:
        pytest.skip("libasan.so not found next to gcc")
    env = dict(os.environ, MPC_TEST_SANITIZE="1", LD_PRELOAD=asan + ((":" + ubsan) if ubsan else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san_run.py")], env=env, capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitized run ok" in r.stdout


def test_lds_bounds_check_of_the_host_context_fires(tmp_path):
    """The bounds check the sanitizer run relies on: an access one word beyond lds_doubles() aborts."""
    src = tmp_path / "oob.cpp"
    src.write_text('#include "%s"\nint main() { double w[4] = {0, 0, 0, 0}; HostCtx c{w, nullptr, 0, 1}; c.nwords = 4; '
                   'c.st(3, 1.0); if (c.ld(3) != 1.0) return 3; c.st(4, 1.0); return 0; }\n'
                   % os.path.join(ROOT, "tests", "host_wave_ctx.hpp"))
    exe = tmp_path / "oob"
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode != 0 and "outside the instance" in r.stderr
