"""Sanitizer + fuzz run of the host side (no GPU): `make -C ryg_rans_amd/csrc asan` builds tests/fuzz/fuzz_host.cpp with
AddressSanitizer + UndefinedBehaviorSanitizer from the library's pure-host sources (container.cpp, model.cpp) and the
oracle, and 10 000 iterations per seed of byte flips, truncations, forged (re-sealed) headers, random indexes, random and
garbage frequency tables and oracle round trips must pass without a sanitizer report."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "fuzz_host_asan")


def _build():
    if shutil.which("g++") is None:
        pytest.skip("no g++ on this box")
    srcs = [os.path.join(ROOT, "tests", "fuzz", "fuzz_host.cpp"), os.path.join(ROOT, "ryg_rans_amd", "csrc", "container.cpp"),
            os.path.join(ROOT, "ryg_rans_amd", "csrc", "model.cpp"), os.path.join(ROOT, "oracle", "rans_oracle.c")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(s) for s in srcs):
        return
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "ryg_rans_amd", "csrc"), "asan"], capture_output=True, text=True)
    if out.returncode != 0:
        if "sanitize" in out.stderr and ("cannot find" in out.stderr or "unrecognized" in out.stderr):
            pytest.skip("the sanitizer runtimes are not installed here")
        raise AssertionError(out.stderr[-3000:])


@pytest.mark.parametrize("seed", [1, 20260926])
def test_host_side_survives_fuzzing_under_asan_ubsan(seed):
    _build()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([EXE, "10000", str(seed)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    assert "no sanitizer report" in out.stdout and "10000 iterations" in out.stdout
    # the run did exercise every target: accepted AND rejected containers, built AND refused models, oracle round trips
    import re
    nums = [int(v) for v in re.findall(r"(\d+)", out.stdout.split("containers")[1])]
    assert all(v > 100 for v in nums[:5]), out.stdout
