"""CPU: the product's field/curve header (anon-aadhaar-halo2_amd/csrc/bn254.cuh, the code the GPU
kernels are built from) compiled for the host and compared with the oracle on ~40k random and edge
operands per field, plus the XYZZ point formulas incl. P+P, P+(-P) and identity cases."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bn254_header_matches_oracle():
    src = os.path.join(ROOT, "tests", "native", "host_field_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "hfc")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    assert "Fr ok" in out and "Fq ok" in out and "G1 ok" in out
