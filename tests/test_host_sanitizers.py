"""AddressSanitizer + UBSan on the host-only part of the engine (params tables, factorisers, rand() fill, bf16
conversion): GPU sanitizers are not available on the pool, the CPU build is."""
import os
import shutil
import subprocess

import pytest
from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_helpers_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_san"
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-w",
           "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "lorastencil_amd", "csrc"),
           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "data", "host_sanitizer_main.cpp"),
           os.path.join(ROOT, "lorastencil_amd", "csrc", "weights.cpp"), "-o", str(exe)]
    subprocess.check_call(cmd)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "SAN_OK" in p.stdout, p.stdout + p.stderr
