"""CPU sanitizers (AddressSanitizer + UndefinedBehaviorSanitizer, failing on the first report) over the code
that runs on the host: the C restatement of the engine (test infrastructure) and the host-only entry points
of libsfmi.  The reference has no such coverage; GPU sanitizers are not available on the pool."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_oracle_restatement_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    subprocess.check_call(["gcc", "-std=c99", "-ffp-contract=off", "-fno-builtin-sin", "-fno-builtin-cos"] + SAN +
                          ["-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "sanitize", "oracle_main.c"),
                           os.path.join(ROOT, "oracle", "sf_oracle.c"), "-lm", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout[-500:], r.stderr[-2000:])


def test_host_entry_points_are_clean_under_asan_ubsan(tmp_path):
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime.h"):
        pytest.skip("HIP headers not installed")
    exe = str(tmp_path / "host_san")
    csrc = os.path.join(ROOT, "spacefortress_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off"] + SAN +
                          ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                           os.path.join(ROOT, "tests", "sanitize", "host_main.cpp"), os.path.join(csrc, "sf_host.cpp"),
                           os.path.join(csrc, "sf_image.cpp"), os.path.join(csrc, "sf_cairo_host.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
