"""Sanitizer leg on the CPU build (the reference's CI runs its tests under ASan and UBSan, .github/workflows/ci.yml:39-134;
GPU AddressSanitizer is not available on the pool).  tests/sanitize/san_driver.cpp links the HOST-side product code -- 1-D tables, the
cube-mesh partitioner, the native mesh / results files -- and the CPU oracle into one executable; built once with
-fsanitize=address,undefined and once with -fsanitize=thread (the oracle's threaded element loop adds into the result vector with atomic
adds, the reference's TBB + atomic_ref scheme), both with -fno-sanitize-recover: any report aborts the run."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["tests/sanitize/san_driver.cpp", "l3ster_amd/csrc/host/tables.cpp", "l3ster_amd/csrc/host/cube_mesh.cpp",
           "l3ster_amd/csrc/host/mesh_file.cpp", "l3ster_amd/csrc/host/native_io.cpp", "oracle/oracle.cpp"]
LEGS = {"asan_ubsan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], "tsan": ["-fsanitize=thread"]}


@pytest.fixture(scope="module")
def drivers(tmp_path_factory):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    out = tmp_path_factory.mktemp("sanitize")
    procs = {}
    for leg, flags in LEGS.items():  # (the two builds side by side: ~25 s)
        cmd = [cxx, "-std=c++20", "-O1", "-g", "-fno-omit-frame-pointer", *flags, "-Iinclude", "-Il3ster_amd/csrc", "-Ioracle", *SOURCES,
               "-o", str(out / f"san_driver_{leg}"), "-lpthread"]
        procs[leg] = subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    exes = {}
    for leg, p in procs.items():
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, f"{leg} build failed:\n{log[-3000:]}"
        exes[leg] = str(out / f"san_driver_{leg}")
    return exes, str(out)


@pytest.mark.parametrize("leg", list(LEGS))
def test_host_code_and_oracle_under_sanitizers(drivers, leg):
    exes, scratch = drivers
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exes[leg], scratch], capture_output=True, text=True, timeout=600, env=env)
    sys.stderr.write(r.stderr[-2000:])
    assert r.returncode == 0, f"{leg}: exit code {r.returncode}\n{r.stderr[-4000:]}"
    assert "sanitizer driver: ok" in r.stdout
    for word in ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: LeakSanitizer"):
        assert word not in r.stderr, r.stderr[-4000:]
