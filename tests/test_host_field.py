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


def test_fp29_field_and_point_formulas_match_32bit_code():
    """The 9 x 29-bit / radix-2^261 base field and its XYZZ mixed addition (csrc/fp29.cuh, used by the
    level-1 MSM kernel) compiled for the host with the documented bounds as hard failures
    (FP29_CHECK_BOUNDS): radix round trips, products, sums, differences and packing on 20k operands, the
    scalar field's mixed-radix product (data in radix 2^256 times a radix-2^261 constant, as the NTT uses it),
    3000 random accumulation chains (identity, doubling, cancellation, negated points), 3000 random
    reduction trees of full additions / doublings against bn254.cuh, and the NTT's radix-4 blocks (lazy first-round sums)
    against the same blocks with normalised intermediates, at random and at the largest operands the bounds allow."""
    src = os.path.join(ROOT, "tests", "native", "fp29_check.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "fp29")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    assert "Fq29 field ok" in out and "Fr29 mixed radix ok" in out and "weak reduction ok" in out and "radix-4 blocks ok" in out and "G1X29 ok" in out and "FAILED" not in out


def test_fp29_constants_are_reproducible():
    """fp29.cuh's constant block is exactly what tools/gen_fp29_consts.py derives from the modulus
    (contract.sol:210-211), for both fields — digits of p, -p^-1 mod 2^29, the redundant multiples of p,
    2^261, 2^266, 2^256 mod p."""
    import sys
    gen = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "gen_fp29_consts.py")], text=True)
    hdr = open(os.path.join(ROOT, "anon-aadhaar-halo2_amd", "csrc", "fp29.cuh")).read()
    assert gen.strip() in hdr
