"""Host-side sanitizer build ("race detection / sanitizers"; CPU only -- GPU sanitizer runs are refused
on the pool): `make asan` compiles the library's plain-C++ parts (contour tracer, command-list executor with HIP calls
replaced by csrc/host_stub/, crc32c) with AddressSanitizer + UndefinedBehaviorSanitizer and runs tools/asan_driver.cpp --
random / ragged / invalid arguments through the C ABI.  The SSTable reader is Python: its fuzz loop is in test_checkpoint.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan():
    cxx = os.environ.get("HOSTCXX", "/opt/rocm/lib/llvm/bin/clang++")
    if not (os.path.exists(cxx) or shutil.which(cxx)):
        pytest.skip("no host clang++ for the sanitizer build")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "dis-yolo_amd", "csrc"), "asan"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "asan_driver: ok" in out, out[-3000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out
