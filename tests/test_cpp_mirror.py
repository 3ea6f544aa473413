"""The C++ host-side mirror of the halo2_proofs interface (include/amdzk_halo2.hpp) — what a compiled
host such as the reference's Rust would use above the C ABI.

CPU: circuits configured in C++ (tests/native/halo2_mirror_check.cpp, mirroring
/root/reference/src/signal.rs:27-49, src/conditional_secrets.rs:81-187, src/timestamp.rs:58-68 and the
RSA-SHA256 column budget of src/lib.rs:263-274) flatten to exactly the C-ABI arrays the Python mirror
produces: query numbering, degree(), blinding_factors(), postfix expression words, constants,
permutation columns.
GPU: keygen + create_proof driven entirely from C++ (Assembly::copy, ParamsKZG::setup, ProvingKey,
create_proof) give the oracle prover's bytes for both transcripts."""
import os
import subprocess
import sys

import pytest

import circuits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
TAU = 0x1234567890ABCDEF1234567
SMALL = dict(num_advice=5, num_lookup_advice=2, lookup_bits=5, num_spread=2, spread_bits=3)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("hmc")
    out = str(d / "halo2_mirror_check")
    libdir = os.path.join(ROOT, "anon-aadhaar-halo2_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", out,
                           os.path.join(ROOT, "tests", "native", "halo2_mirror_check.cpp"), "-L", libdir, "-lamdzk", "-Wl,-rpath," + libdir])
    return out


@pytest.fixture(scope="module")
def plonk():
    import __graft_entry__ as g
    return g.load_package().plonk


def fixture_for(plonk, name):
    if name == "square":
        return circuits.square_circuit(plonk, 4, signal=9)
    if name == "lookup":
        return circuits.lookup_circuit(plonk, 5, seed=3)
    if name == "aadhaar_small":
        return circuits.full_aadhaar_shape(plonk, k=7, **SMALL)
    raise KeyError(name)


def python_description(plonk, cs, k):
    desc = cs.describe(k)
    cc, a = plonk.flatten_circuit(desc)
    flat = lambda arr: " ".join(str(int(v)) for v in arr.reshape(-1))
    line = lambda tag, body: tag + (" " + body if body else "")
    return "\n".join([
        "shape %d %d %d %d %d %d %d %d %d %d" % (k, cs.num_fixed, cs.num_advice, cs.num_instance, cs.blinding_factors(), cs.degree(),
                                               len(desc["gates"]), len(desc["lookups"]), cc.num_exprs, cs.minimum_rows()),
        line("aq", flat(a["aq"])), line("fq", flat(a["fq"])), line("iq", flat(a["iq"])), line("lookup_shape", flat(a["shape"])),
        line("expr_offsets", flat(a["off"])), line("expr_words", flat(a["words"])),
        line("constants", " ".join("%016x" % int(v) for v in a["consts"].reshape(-1))), line("perm", flat(a["perm"]))]) + "\n"


@pytest.mark.parametrize("name", ["square", "lookup", "aadhaar_small", "aadhaar"])
def test_cpp_constraint_system_flattens_like_python(exe, plonk, name):
    if name == "aadhaar":  # full column budget; only the configuration is needed
        cs, k = circuits.full_aadhaar_shape(plonk, k=9, lookup_bits=6, spread_bits=4).cs, 15
    else:
        c = fixture_for(plonk, name)
        cs, k = c.cs, c.k
    got = subprocess.check_output([exe, "describe", name, str(k)], text=True)
    assert got == python_description(plonk, cs, k)


def write_witness(c, path):
    with open(path, "w") as f:
        for tag, cols in (("F", c.fixed), ("A", c.advice)):
            for ci, col in enumerate(cols):
                for r, v in enumerate(col):
                    if v:
                        f.write("%s %d %d %x\n" % (tag, ci, r, v))
        for ci, col in enumerate(c.instances):
            f.write("I %d %s\n" % (ci, " ".join("%x" % v for v in col)))
        for cp in c.copies:
            f.write("C %d %d %d %d\n" % cp)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["square", "lookup", "aadhaar_small"])
def test_cpp_driven_proof_equals_oracle(exe, plonk, name, tmp_path):
    import plonk_ref as PR

    c = fixture_for(plonk, name)
    wit = str(tmp_path / "witness.txt")
    write_witness(c, wit)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=0xC0FFEE)
    for tr, mo in (("blake2b", "shplonk"), ("evm", "shplonk"), ("blake2b", "gwc")):
        fmt = tr + ("+gwc" if mo == "gwc" else "")
        out = subprocess.check_output([exe, "prove", name, str(c.k), wit, "17", fmt, "%x" % TAU, "%x" % 0xC0FFEE], text=True)
        proof = bytes.fromhex(out.split("proof ")[1].split()[0])
        assert proof == PR.create_proof(opk, c.instances, c.advice, seed=17, transcript=tr, multiopen=mo), (name, fmt)
        assert PR.verify_proof(opk, c.instances, proof, transcript=tr, multiopen=mo)
        assert "commitments %d %d" % (c.cs.num_fixed, len(c.cs.permutation_columns)) in out
        # upstream's slices through the C++ mirror: the circuit twice in one proof (key + workspace clone)
        two = bytes.fromhex(out.split("multi2 ")[1].split()[0])
        assert two == PR.create_proof_multi(opk, [c.instances] * 2, [c.advice] * 2, seed=17, transcript=tr, multiopen=mo), (name, fmt, "two instances")
        # WitnessStream (pinned staging, async upload, fence): same bytes as the resident path, seed by seed
        streamed = [bytes.fromhex(ln.split()[1]) for ln in out.splitlines() if ln.startswith("streamed ")]
        assert len(streamed) == 3 and streamed[0] == proof
        if tr == "blake2b" and mo == "shplonk":
            assert streamed[1] == PR.create_proof(opk, c.instances, c.advice, seed=18) and streamed[2] != streamed[1]
