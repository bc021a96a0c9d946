"""The C++ host-side mirror of the reference interface (include/l3k/operator.hpp): compiles against l3k.h on CPU; on the
GPU box the example program runs (apply == K_e x through the C++ API, operator symmetry, exception on misuse)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "diffusion3d_mf.cpp")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _build(out):
    cmd = [HIPCC, "-std=c++20", "-O1", "--offload-arch=gfx950", f"-I{ROOT}/include", SRC, f"-L{ROOT}/l3ster_amd/lib", "-ll3k",
           f"-Wl,-rpath,{ROOT}/l3ster_amd/lib", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_cpp_shim_compiles_and_links(tmp_path):
    _build(str(tmp_path / "diffusion3d_mf"))


@pytest.mark.gpu
def test_cpp_shim_runs(tmp_path):
    exe = str(tmp_path / "diffusion3d_mf")
    _build(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
