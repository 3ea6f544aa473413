"""CPU: the product's host-side C++ (field / curve header bn254.cuh, the 29-bit form fp29.cuh, transcripts and RNG
hostcrypto.hpp — the same headers the GPU kernels and the prover driver are built from) compiled for the host with
AddressSanitizer + UndefinedBehaviorSanitizer, every report fatal, and run through the native checks' full operand sets.
(GPU sanitizers are not available on the pool; the host halves of these headers are where an out-of-bounds index or a
shift past the type's width would be the same bug on both sides.)"""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1", "-std=c++17"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.mark.parametrize("src,expect", [
    ("host_field_check.cpp", ["Fr ok", "Fq ok", "G1 ok"]),
    ("fp29_check.cpp", ["Fq29 field ok", "Fr29 mixed radix ok", "weak reduction ok", "radix-4 blocks ok", "G1X29 ok"]),
    ("hostcrypto_kat.cpp", ["keccak256_empty"]),
])
def test_native_checks_are_clean_under_asan_and_ubsan(src, expect):
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "san_" + src.split(".")[0])
        subprocess.check_call(["g++", *SAN, "-o", exe, os.path.join(ROOT, "tests", "native", src)])
        p = subprocess.run([exe], text=True, capture_output=True, env=ENV, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr, p.stderr[-4000:]
    for e in expect:
        assert e in p.stdout
    assert "FAILED" not in p.stdout


@pytest.mark.parametrize("name,k", [("square", 4), ("lookup", 5), ("aadhaar_small", 7), ("aadhaar", 15)])
def test_cpp_mirror_configuration_is_clean_under_asan_and_ubsan(name, k):
    """include/amdzk_halo2.hpp (ConstraintSystem, Expression flattening, Assembly) building the reference's circuit
    configurations (tests/native/halo2_mirror_check.cpp `describe`: no GPU call) under the sanitizers. The HIP runtime that
    libamdzk.so pulls in is not ours to leak-check: leak detection is off here."""
    libdir = os.path.join(ROOT, "anon-aadhaar-halo2_amd")
    if not os.path.exists(os.path.join(libdir, "libamdzk.so")):
        pytest.skip("libamdzk.so not built")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "san_mirror")
        subprocess.check_call(["g++", *SAN, "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe,
                               os.path.join(ROOT, "tests", "native", "halo2_mirror_check.cpp"), "-L", libdir, "-lamdzk", "-Wl,-rpath," + libdir])
        env = dict(ENV, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
        p = subprocess.run([exe, "describe", name, str(k)], text=True, capture_output=True, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
    assert p.stdout.startswith("shape %d " % k)
