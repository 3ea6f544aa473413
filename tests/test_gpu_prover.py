"""GPU parity of the whole create_proof: the proof bytes produced on the MI355X equal, byte for byte,
the bytes the pure-Python protocol oracle produces from the same circuit, witness, SRS and RNG seed
(small k), and proofs of the full RSA-SHA256-shaped circuit at the reference's k = 15 are accepted by
the oracle's verifier (vanishing identity + SHPLONK opening equation)."""
import os
import sys

import numpy as np
import pytest
import zkutil as zu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import circuits  # noqa: E402
import plonk_ref as PR  # noqa: E402

pytestmark = pytest.mark.gpu
TAU = 0x1234567890ABCDEF1234567


@pytest.fixture(scope="module")
def plonk(pkg):
    return __import__("anon_aadhaar_halo2_amd.halo2.plonk", fromlist=["x"])


_srs_cache = {}


def setup(ctx, pkg, plonk, oracle, c, transcript_repr_int=123456789, flags=None):
    if c.k not in _srs_cache:
        _srs_cache[c.k] = zu.test_srs(oracle, c.k, TAU)
    g, gl = _srs_cache[c.k]
    params = pkg.kzg.ParamsKZG(ctx, c.k, g=g, g_lagrange=gl)
    fixed = np.stack([zu.ints_to_fr(oracle, col) for col in c.fixed]) if c.fixed else np.zeros((0, c.n, 4), np.uint64)
    pk = plonk.ProvingKey(ctx, params, c.desc, fixed, c.assembly.mapping, zu.fr_from_int(transcript_repr_int), flags=flags)
    adv = np.stack([zu.ints_to_fr(oracle, col) for col in c.advice])
    d_adv = ctx.alloc(adv.nbytes).upload(adv)
    inst = [zu.ints_to_fr(oracle, col) if col else np.zeros((0, 4), np.uint64) for col in c.instances]
    return params, pk, d_adv, inst


@pytest.mark.parametrize("tables", ["range_first", "pair_first", "range_expr"])
def test_constant_tables_sorted_at_keygen(ctx, pkg, plonk, oracle, tables, monkeypatch):
    """A leading lookup whose table is one expression over fixed columns has its sorted table made at keygen
    (amdzk_pk::lk_const). The proof bytes are the oracle's with such a lookup in front, with it behind a theta-compressed
    one (so the cache does not apply), with a table expression rather than a plain column — and with the cache off."""
    c = circuits.lookup_circuit(plonk, 6, seed=21, tables=tables)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
    want = PR.create_proof(opk, c.instances, c.advice, seed=5)
    for cache in (True, False):
        if not cache:
            monkeypatch.setenv("AMDZK_NO_TABLE_CACHE", "1")
        for flags in (None, plonk.KEYGEN_SERIAL):
            params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c, flags=flags)
            assert plonk.create_proof(ctx, pk, inst, d_adv, seed=5) == want
            assert plonk.create_proof(ctx, pk, inst, d_adv, seed=5) == want
            d_adv.free(); pk.free(); params.free()


def vk_from_device(pk, c, transcript_repr_int=123456789):
    f, p = pk.commitments()
    return PR.VerifyingKey(c.desc, [zu.point_to_ints(x) for x in f], [zu.point_to_ints(x) for x in p], TAU, transcript_repr_int)


@pytest.mark.parametrize("name,k", [("square", 4), ("square", 6), ("lookup", 5), ("lookup", 7)])
def test_proof_bytes_equal_oracle(ctx, pkg, plonk, oracle, name, k):
    c = circuits.square_circuit(plonk, k, signal=5) if name == "square" else circuits.lookup_circuit(plonk, k, seed=k)
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    got = plonk.create_proof(ctx, pk, inst, d_adv, seed=99)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
    # keygen parity: VK commitments computed on the device equal the oracle's
    f, p = pk.commitments()
    assert [zu.point_to_ints(x) for x in f] == opk.fixed_commitments
    assert [zu.point_to_ints(x) for x in p] == opk.permutation_commitments
    want = PR.create_proof(opk, c.instances, c.advice, seed=99)
    assert got == want
    assert PR.verify_proof(vk_from_device(pk, c), c.instances, got)
    assert plonk.create_proof(ctx, pk, inst, d_adv, seed=99) == got  # workspace reuse is clean
    assert plonk.create_proof(ctx, pk, inst, d_adv, seed=100) != got
    d_adv.free(); pk.free(); params.free()


def test_unsatisfied_witness_does_not_verify(ctx, pkg, plonk, oracle):
    c = circuits.lookup_circuit(plonk, 6, seed=3)
    c.advice[2][0] = (c.advice[2][0] + 1) % zu.R  # break the first mul gate
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    proof = plonk.create_proof(ctx, pk, inst, d_adv, seed=5)
    with pytest.raises(AssertionError):
        PR.verify_proof(vk_from_device(pk, c), c.instances, proof)
    d_adv.free(); pk.free(); params.free()


def test_quotient_coset_modes(ctx, pkg, plonk, oracle, monkeypatch):
    """The quotient is computed on cs_degree - 1 cosets of the size-n subgroup; AMDZK_FULL_COSETS=1
    (read at keygen) uses all 2^(ek-k), which is upstream's own extended-domain computation. For a satisfying witness
    both give the oracle's bytes. For a witness that does NOT satisfy the circuit the numerator is not divisible by
    X^n - 1: the full mode still reproduces the oracle's (= upstream's) bytes, the default mode interpolates a
    different polynomial — and neither proof verifies."""
    small = dict(k=7, num_advice=5, num_lookup_advice=2, lookup_bits=5, num_spread=2, spread_bits=3)
    good = circuits.rsa_sha256_shape(plonk, **small)  # degree 4: 3 of the 4 cosets by default
    bad = circuits.rsa_sha256_shape(plonk, **small)
    assert good.desc["cs_degree"] == 4
    bad.advice[0][3] = (bad.advice[0][3] + 1) % zu.R  # break the first vertical gate's output cell
    with pytest.raises(AssertionError):
        circuits.check_satisfied(bad, rows=range(4))
    proofs = {}
    for mode in ("default", "full"):
        if mode == "full":
            monkeypatch.setenv("AMDZK_FULL_COSETS", "1")
        else:
            monkeypatch.delenv("AMDZK_FULL_COSETS", raising=False)
        for name, c in (("good", good), ("bad", bad)):
            params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
            proofs[(mode, name)] = plonk.create_proof(ctx, pk, inst, d_adv, seed=21)
            d_adv.free(); pk.free(); params.free()
    monkeypatch.delenv("AMDZK_FULL_COSETS", raising=False)
    opk = PR.keygen(good.desc, good.fixed, good.assembly.mapping, TAU, transcript_repr=123456789)
    want_good = PR.create_proof(opk, good.instances, good.advice, seed=21)
    want_bad = PR.create_proof(opk, bad.instances, bad.advice, seed=21)
    assert proofs[("default", "good")] == want_good and proofs[("full", "good")] == want_good
    assert proofs[("full", "bad")] == want_bad
    assert proofs[("default", "bad")] != want_bad
    for mode in ("default", "full"):
        with pytest.raises(AssertionError):
            PR.verify_proof(opk, bad.instances, proofs[(mode, "bad")])


def test_lanes_and_serial_mode_give_the_same_bytes(ctx, pkg, plonk, oracle):
    """create_proof spreads one proof over three streams (lanes: the coset transforms of a phase, the lookup products
    and the random polynomial run beside the commitments the transcript waits for). A key made with
    AMDZK_KEYGEN_SERIAL keeps everything on the caller's stream. Only the order of transcript writes is fixed by
    upstream, so the bytes must be the same — for SHPLONK and GWC, for circuits with lookups and permutations and for
    the reference's SquareCircuit (no lookup, one permutation set), with proofs back to back on one key (the lanes
    of proof i must be quiet before proof i + 1 reuses the workspace) and with the flag-selected full-coset mode."""
    shapes = [circuits.rsa_sha256_shape(plonk, k=7, num_advice=5, num_lookup_advice=2, lookup_bits=5, num_spread=2, spread_bits=3),
              circuits.lookup_circuit(plonk, 6, seed=9), circuits.square_circuit(plonk, 4, signal=5)]
    for c in shapes:
        got = {}
        for mode, flags in (("lanes", 0), ("serial", plonk.KEYGEN_SERIAL), ("serial_full", plonk.KEYGEN_SERIAL | plonk.KEYGEN_FULL_COSETS),
                            ("lanes_full", plonk.KEYGEN_FULL_COSETS)):
            params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c, flags=flags)
            got[mode] = [plonk.create_proof(ctx, pk, inst, d_adv, seed=33),
                         plonk.create_proof(ctx, pk, inst, d_adv, seed=34, transcript=plonk.MULTIOPEN_GWC),
                         plonk.create_proof(ctx, pk, inst, d_adv, seed=33)]
            d_adv.free(); pk.free(); params.free()
        assert got["lanes"] == got["serial"] == got["serial_full"] == got["lanes_full"]
        assert got["lanes"][0] == got["lanes"][2]
        opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
        assert got["lanes"][0] == PR.create_proof(opk, c.instances, c.advice, seed=33)
        assert got["lanes"][1] == PR.create_proof(opk, c.instances, c.advice, seed=34, multiopen="gwc")


def test_h_program_pieces_give_the_same_bytes(ctx, pkg, plonk, oracle, monkeypatch):
    """The h(X) program is cut into pieces that run side by side into their own h and are summed (6 by
    default, AMDZK_H_PARTS at keygen): h is linear in the constraint terms, so every cut gives the oracle's bytes — 1 to 8
    pieces, more pieces than a small program has terms included, on a circuit with lookups and on the SquareCircuit."""
    for c in (circuits.lookup_circuit(plonk, 6, seed=13), circuits.square_circuit(plonk, 4, signal=5)):
        opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
        want = PR.create_proof(opk, c.instances, c.advice, seed=41)
        for parts in (1, 2, 3, 5, 8, 64):
            monkeypatch.setenv("AMDZK_H_PARTS", str(parts))
            params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
            assert plonk.create_proof(ctx, pk, inst, d_adv, seed=41) == want, parts
            d_adv.free(); pk.free(); params.free()
        monkeypatch.delenv("AMDZK_H_PARTS")


@pytest.mark.parametrize("power,flags", [(9, 0), (9, 1), (4, 0), (16, 0)])
def test_high_degree_gate_bytes_equal_oracle(ctx, pkg, plonk, oracle, power, flags):
    """Constraint degree 10 (9 quotient pieces: more cosets than the templated recombination kernel's 8 — the generic
    kernel; ADVICE r2: keygen used to refuse it), the same in full-coset mode (16 cosets), degree 5 (4 cosets = the
    whole extended domain) and degree 17 (16 pieces on 16 cosets): byte-equal to the oracle prover, and verified."""
    c = circuits.high_degree_circuit(plonk, 5, power=power)
    assert c.desc["cs_degree"] == power + 1
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c, flags=flags)
    got = plonk.create_proof(ctx, pk, inst, d_adv, seed=17)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
    assert got == PR.create_proof(opk, c.instances, c.advice, seed=17)
    assert PR.verify_proof(opk, c.instances, got)
    d_adv.free(); pk.free(); params.free()


def test_host_wait_modes_give_the_same_bytes(ctx, pkg, plonk, oracle):
    """amdzk_set_host_wait: spinning (hipStreamSynchronize) and blocking (polled completion event) waits are the same
    proof — lanes and serial keys, and the mode can be switched between proofs on one context."""
    c = circuits.lookup_circuit(plonk, 6, seed=5)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
    want = PR.create_proof(opk, c.instances, c.advice, seed=41)
    try:
        for flags in (0, plonk.KEYGEN_SERIAL):
            params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c, flags=flags)
            for block in (True, False, True):
                ctx.set_host_wait(block)
                assert plonk.create_proof(ctx, pk, inst, d_adv, seed=41) == want
            d_adv.free(); pk.free(); params.free()
    finally:
        ctx.set_host_wait(False)
    with pytest.raises(Exception):
        ctx._chk(ctx.L.amdzk_set_host_wait(ctx.h, 7))


def test_keygen_ex_rejects_unknown_flags(ctx, pkg, plonk, oracle):
    c = circuits.square_circuit(plonk, 4, signal=5)
    with pytest.raises(Exception):
        setup(ctx, pkg, plonk, oracle, c, flags=0x80)


def test_lookup_failure_is_reported(ctx, pkg, plonk, oracle):
    c = circuits.lookup_circuit(plonk, 5, seed=4)
    rows = [r for r in range(c.usable) if c.fixed[2][r] == 1]  # fixed[2] = q_rng: range lookup enabled
    c.advice[0][rows[0]] = 9  # not in the 0..7 range table
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    with pytest.raises(pkg.AmdzkError) as e:
        plonk.create_proof(ctx, pk, inst, d_adv, seed=5)
    assert "lookup 0 input not in table" in str(e.value)
    d_adv.free(); pk.free(); params.free()
    # the two-column lookup (index 1, behind a lookup whose table was sorted at keygen)
    c = circuits.lookup_circuit(plonk, 5, seed=4)
    c.advice[2][c.usable - 1] = (c.advice[2][c.usable - 1] + 1) % circuits.R  # (b, c) no longer a (x, x^2) pair of the table
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    with pytest.raises(pkg.AmdzkError) as e:
        plonk.create_proof(ctx, pk, inst, d_adv, seed=5)
    assert "lookup 1 input not in table" in str(e.value)
    d_adv.free(); pk.free(); params.free()


def test_rsa_sha256_shape_small_equals_oracle(ctx, pkg, plonk, oracle):
    c = circuits.rsa_sha256_shape(plonk, k=7, num_advice=5, num_lookup_advice=2, lookup_bits=5, num_spread=2, spread_bits=3)
    circuits.check_satisfied(c)
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    got = plonk.create_proof(ctx, pk, inst, d_adv, seed=7)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
    assert got == PR.create_proof(opk, c.instances, c.advice, seed=7)
    assert PR.verify_proof(vk_from_device(pk, c), c.instances, got)
    d_adv.free(); pk.free(); params.free()


def test_rsa_sha256_shape_k15_bytes_equal_cpu_prover_and_verify(ctx, pkg, plonk, oracle):
    """Full budget of TestRSASignatureWithHashCircuit1 (/root/reference/src/lib.rs:263-274) at the
    reference's k = 15: 112 advice, 24 lookups, 115 permutation columns. The MI355X's proof is byte
    for byte the CPU oracle prover's (oracle/plonk_fast.py, same witness / SRS / seed) and verifies."""
    import plonk_fast as PF

    c = circuits.rsa_sha256_shape(plonk, k=15)
    circuits.check_satisfied(c, rows=range(0, c.usable, 997))
    assert c.desc["num_advice"] == 112 and len(c.desc["lookups"]) == 24 and len(c.desc["permutation_columns"]) == 115
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    proof = plonk.create_proof(ctx, pk, inst, d_adv, seed=2024)
    assert PR.verify_proof(vk_from_device(pk, c), c.instances, proof)
    fpk = PF.FastKey(c.desc, c.fixed, c.assembly.mapping, TAU, 123456789)
    f, p = pk.commitments()
    assert [zu.point_to_ints(x) for x in f] == fpk.fixed_commitments
    assert [zu.point_to_ints(x) for x in p] == fpk.permutation_commitments
    assert proof == PF.create_proof(fpk, c.instances, c.advice, seed=2024)
    d_adv.free(); pk.free(); params.free()


def test_rsa_sha256_shape_k18_bytes_equal_cpu_prover_and_verify(ctx, pkg, plonk, oracle):
    """BASELINE config 2 at its stated k ~ 18: the RSA-SHA256 shape re-configured to k = 18 (same area, 8x fewer gate
    columns: workloads.SHAPES["k18"]; shape source /root/reference/src/lib.rs:263-274). Different window width
    (c = 14), 3-step NTT at 2^20, larger workspaces than any k = 15 test. The SRS is built on the device
    (amdzk_srs_setup, itself checked against the oracle in test_gpu_msm.py) and handed to the CPU oracle prover; one
    seed — the CPU proof at this size takes the better part of a minute."""
    import plonk_fast as PF

    shape = dict(circuits.SHAPES["k18"])
    c = circuits.rsa_sha256_shape(plonk, **shape)
    assert c.k == 18 and c.desc["num_advice"] == 14 and len(c.desc["lookups"]) == 3
    circuits.check_satisfied(c, rows=range(0, c.usable, 7919))
    params = pkg.kzg.ParamsKZG.setup(ctx, c.k, zu.fr_from_int(TAU), want_host_copy=True)
    fixed = np.stack([zu.ints_to_fr(oracle, col) for col in c.fixed])
    pk = plonk.ProvingKey(ctx, params, c.desc, fixed, c.assembly.mapping, zu.fr_from_int(123456789))
    adv = np.stack([zu.ints_to_fr(oracle, col) for col in c.advice])
    d_adv = ctx.alloc(adv.nbytes).upload(adv)
    inst = [zu.ints_to_fr(oracle, col) if col else np.zeros((0, 4), np.uint64) for col in c.instances]
    proof = plonk.create_proof(ctx, pk, inst, d_adv, seed=1818)
    assert len(proof) == plonk.proof_size(ctx, pk)
    fpk = PF.FastKey(c.desc, c.fixed, c.assembly.mapping, TAU, 123456789, msm_bases=(params._g, params._gl))
    f, p = pk.commitments()
    assert [zu.point_to_ints(x) for x in f] == fpk.fixed_commitments
    assert [zu.point_to_ints(x) for x in p] == fpk.permutation_commitments
    assert proof == PF.create_proof(fpk, c.instances, c.advice, seed=1818)
    assert PR.verify_proof(vk_from_device(pk, c), c.instances, proof)
    d_adv.free(); pk.free(); params.free()


def test_full_aadhaar_shape_equals_oracle(ctx, pkg, plonk, oracle):
    """Composite AadhaarQRVerifierCircuit budget (/root/reference/src/aadhaar_verifier_circuit.rs:49-56,
    BASELINE config 3): RSA shape + IdentityCircuit gates + 7 never-queried timestamp columns +
    SquareCircuit. Small k: bytes equal the pure-Python oracle prover for both transcripts; k = 15 with the
    full column budget (141 advice, 118 permutation columns): bytes equal the CPU oracle prover and verify."""
    import plonk_fast as PF

    c = circuits.full_aadhaar_shape(plonk, k=7, num_advice=5, num_lookup_advice=2, lookup_bits=5, num_spread=2, spread_bits=3)
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
    assert plonk.create_proof(ctx, pk, inst, d_adv, seed=9) == PR.create_proof(opk, c.instances, c.advice, seed=9)
    assert plonk.create_proof(ctx, pk, inst, d_adv, seed=9, transcript=plonk.TRANSCRIPT_KECCAK256_EVM) == \
        PR.create_proof(opk, c.instances, c.advice, seed=9, transcript="evm")
    d_adv.free(); pk.free(); params.free()

    c = circuits.full_aadhaar_shape(plonk, k=15)
    assert c.desc["num_advice"] == 141 and len(c.desc["lookups"]) == 24 and len(c.desc["permutation_columns"]) == 118
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    proof = plonk.create_proof(ctx, pk, inst, d_adv, seed=77)
    assert PR.verify_proof(vk_from_device(pk, c), c.instances, proof)
    fpk = PF.FastKey(c.desc, c.fixed, c.assembly.mapping, TAU, 123456789)
    assert proof == PF.create_proof(fpk, c.instances, c.advice, seed=77)
    evm = plonk.create_proof(ctx, pk, inst, d_adv, seed=78, transcript=plonk.TRANSCRIPT_KECCAK256_EVM)
    assert PR.verify_proof(vk_from_device(pk, c), c.instances, evm, transcript="evm")
    assert evm == PF.create_proof(fpk, c.instances, c.advice, seed=78, transcript="evm")
    d_adv.free(); pk.free(); params.free()


def test_gwc_multiopen_equals_oracle(ctx, pkg, plonk, oracle):
    """ProverGWC on the device (AMDZK_MULTIOPEN_GWC): bytes equal the oracle's GWC prover on small circuits
    for both transcripts, amdzk_proof_size predicts the length, and at k = 15 with the composite Aadhaar
    budget (6 opening points: rotations 0..3, -1, -(bf+1)) the proof verifies with the oracle's GWC verifier."""
    gwc = plonk.MULTIOPEN_GWC
    for c in (circuits.square_circuit(plonk, 4, signal=5), circuits.lookup_circuit(plonk, 6, seed=4),
              circuits.full_aadhaar_shape(plonk, k=7, num_advice=5, num_lookup_advice=2, lookup_bits=5, num_spread=2, spread_bits=3)):
        params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
        opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
        for kind, name in ((plonk.TRANSCRIPT_BLAKE2B, "blake2b"), (plonk.TRANSCRIPT_KECCAK256_EVM, "evm")):
            got = plonk.create_proof(ctx, pk, inst, d_adv, seed=12, transcript=kind | gwc)
            assert got == PR.create_proof(opk, c.instances, c.advice, seed=12, transcript=name, multiopen="gwc")
            assert len(got) == plonk.proof_size(ctx, pk, kind | gwc)
            assert len(plonk.create_proof(ctx, pk, inst, d_adv, seed=12, transcript=kind)) == plonk.proof_size(ctx, pk, kind)
        d_adv.free(); pk.free(); params.free()
    c = circuits.full_aadhaar_shape(plonk, k=15)
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    proof = plonk.create_proof(ctx, pk, inst, d_adv, seed=5, transcript=gwc)
    assert len(proof) == plonk.proof_size(ctx, pk, gwc) == plonk.proof_size(ctx, pk, 0) + 4 * 32
    assert PR.verify_proof(vk_from_device(pk, c), c.instances, proof, multiopen="gwc")
    d_adv.free(); pk.free(); params.free()


def test_concurrent_proofs_on_separate_contexts_equal_sequential_ones(pkg, plonk, oracle):
    """What bench.py relies on: four contexts (own stream, SRS handle, proving key and workspace each) driven
    from four host threads at once produce exactly the bytes the same seeds give one at a time — no state
    is shared between contexts (twiddle caches, workspaces, staging buffers are per context / per key).
    Checked on the composite shape at k = 9 so that every kernel family of the prover runs concurrently."""
    import threading

    c = circuits.full_aadhaar_shape(plonk, k=9, num_advice=6, num_lookup_advice=2, lookup_bits=6, num_spread=2, spread_bits=4)
    ctxs = [pkg.Context(0) for _ in range(4)]
    keys = [setup(cx, pkg, plonk, oracle, c) for cx in ctxs]
    seeds = [[1000 * w + i for i in range(3)] for w in range(4)]
    sequential = [[plonk.create_proof(ctxs[w], keys[w][1], keys[w][3], keys[w][2], seed=sd) for sd in seeds[w]] for w in range(4)]
    concurrent = [[None] * 3 for _ in range(4)]
    errors = []

    def work(w):
        try:
            for i, sd in enumerate(seeds[w]):
                concurrent[w][i] = plonk.create_proof(ctxs[w], keys[w][1], keys[w][3], keys[w][2], seed=sd)
        except Exception as e:  # surfaced below: an exception in a thread must fail the test
            errors.append(e)

    th = [threading.Thread(target=work, args=(w,)) for w in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert concurrent == sequential
    # same seed on different contexts: same proof (contexts are interchangeable)
    assert plonk.create_proof(ctxs[1], keys[1][1], keys[1][3], keys[1][2], seed=seeds[0][0]) == sequential[0][0]
    assert PR.verify_proof(vk_from_device(keys[0][1], c), c.instances, sequential[0][0])
    for (params, pk, d_adv, _), cx in zip(keys, ctxs):
        d_adv.free(); pk.free(); params.free()
        cx.close()


def test_evm_proof_accepted_by_reference_solidity_verifier(ctx, pkg, plonk, oracle):
    """Config 1/3 of BASELINE.json in the form the reference can check: the SquareCircuit
    (/root/reference/src/signal.rs) proved on the MI355X with the Keccak256/EVM transcript equals the
    oracle prover's bytes and is accepted by the restated contract.sol (oracle/contract_sol.py)."""
    import contract_sol as CS

    for k in (4, 8):
        c = circuits.square_circuit(plonk, k, signal=5)
        params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c, transcript_repr_int=0xABCDEF)
        got = plonk.create_proof(ctx, pk, inst, d_adv, seed=21, transcript=plonk.TRANSCRIPT_KECCAK256_EVM)
        assert len(got) == 0x0460  # contract.sol:221
        opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=0xABCDEF)
        assert got == PR.create_proof(opk, c.instances, c.advice, seed=21, transcript="evm")
        f, p = pk.commitments()
        vk = CS.vk_words(0xABCDEF, 0, k, [zu.point_to_ints(x) for x in f], [zu.point_to_ints(x) for x in p])
        assert CS.verify_proof(vk, got, [], TAU)
        bad = bytearray(got)
        bad[-1] ^= 1
        assert not CS.verify_proof(vk, bytes(bad), [], TAU)
        d_adv.free(); pk.free(); params.free()


def test_caller_supplied_randomness_equals_seeded_rng(ctx, pkg, plonk, oracle):
    """amdzk_create_proof_scalars with the scalars a ChaCha20Rng would have drawn gives the same bytes as
    the seeded entry point (draw count and order = SURVEY.md Appendix A); too few scalars is an error."""
    c = circuits.lookup_circuit(plonk, 6, seed=9)
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    want = plonk.create_proof(ctx, pk, inst, d_adv, seed=77)
    cnt = plonk.proof_random_count(ctx, pk)
    rng = PR.ChaCha20Rng(77)
    draws = zu.ints_to_fr(oracle, [rng.fr() for _ in range(cnt)])
    assert plonk.create_proof_with_scalars(ctx, pk, inst, d_adv, draws) == want
    with pytest.raises(pkg.AmdzkError):
        plonk.create_proof_with_scalars(ctx, pk, inst, d_adv, draws[:-1])
    d_adv.free(); pk.free(); params.free()


@pytest.mark.parametrize("k,ncirc,transcript", [(7, 2, "blake2b"), (7, 2, "keccak"), (8, 2, "blake2b"), (7, 3, "blake2b")])
def test_several_circuit_instances_in_one_proof(ctx, pkg, plonk, oracle, k, ncirc, transcript):
    """create_proof(params, pk, &[circuit; N], &[instances; N], ..) — upstream's slices: N instances of one
    circuit, different witnesses and public inputs, in ONE proof: bytes equal to the oracle's create_proof_multi (both
    transcripts), accepted by its verifier, N = 1 through the same entry point equals create_proof, a clone alone is a
    complete key, and the order of the instances matters."""
    small = dict(k=k, num_advice=5, num_lookup_advice=2, lookup_bits=min(5, k - 2), num_spread=2, spread_bits=3)
    c = circuits.rsa_sha256_shape(plonk, **small)
    wit = [(c.advice, c.instances)] + [c.witness(300 + j) for j in range(1, ncirc)]
    for a, i in wit[1:]:
        circuits.check_satisfied(c, advice=a, instances=i)
    assert wit[1][0] != wit[0][0]
    tk = plonk.TRANSCRIPT_BLAKE2B if transcript == "blake2b" else plonk.TRANSCRIPT_KECCAK256_EVM
    params, pk, d_adv0, inst0 = setup(ctx, pkg, plonk, oracle, c)
    pks = [pk] + [pk.clone_workspace() for _ in range(1, ncirc)]
    d_adv, inst = [d_adv0], [inst0]
    for a, i in wit[1:]:
        arr = np.stack([zu.ints_to_fr(oracle, col) for col in a])
        d_adv.append(ctx.alloc(arr.nbytes).upload(arr))
        inst.append([zu.ints_to_fr(oracle, col) if col else np.zeros((0, 4), np.uint64) for col in i])
    got = plonk.create_proof_multi(ctx, pks, inst, d_adv, seed=17, transcript=tk)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
    want = PR.create_proof_multi(opk, [i for _, i in wit], [a for a, _ in wit], seed=17, transcript=transcript)
    assert len(got) == plonk.proof_size_multi(ctx, pk, ncirc, tk) == len(want)
    assert got == want
    assert PR.verify_proof_multi(vk_from_device(pk, c), [i for _, i in wit], got, transcript=transcript)
    assert plonk.create_proof_multi(ctx, pks, inst, d_adv, seed=17, transcript=tk) == got  # workspaces reused cleanly
    # one instance through the slice entry point = create_proof; a clone alone is a complete key
    single = plonk.create_proof(ctx, pk, inst0, d_adv0, seed=17, transcript=tk)
    assert plonk.create_proof_multi(ctx, [pk], [inst0], [d_adv0], seed=17, transcript=tk) == single
    assert plonk.create_proof(ctx, pks[1], inst0, d_adv0, seed=17, transcript=tk) == single
    assert single == PR.create_proof(opk, c.instances, c.advice, seed=17, transcript=transcript)
    # the instances' order is part of the statement
    swapped = plonk.create_proof_multi(ctx, pks, inst[::-1], d_adv[::-1], seed=17, transcript=tk)
    assert swapped != got
    with pytest.raises(AssertionError):
        PR.verify_proof_multi(vk_from_device(pk, c), [i for _, i in wit], swapped, transcript=transcript)
    # the same workspace twice is refused
    with pytest.raises(pkg.AmdzkError):
        plonk.create_proof_multi(ctx, [pk, pk], inst[:2], d_adv[:2], seed=17, transcript=tk)
    for d in d_adv:
        d.free()
    for q in pks[1:]:
        q.free()
    pk.free(); params.free()


def test_create_proof_argument_errors_leave_the_context_usable(ctx, pkg, plonk, oracle):
    """The error half of the boundary (`int` return, message via amdzk_last_error, never aborts): every
    refused call names its reason, and the same context and key prove the right bytes afterwards.
      * an instance column longer than the usable rows — upstream's Error::InstanceTooLarge
      * a null advice pointer, an advice stride < n, an unknown transcript kind, a null instance column with a length
      * a proof buffer that is too small: refused, with the needed length reported in *proof_len
      * proof_out = NULL: the length query"""
    import ctypes as C
    c = circuits.lookup_circuit(plonk, 6, seed=4)
    params, pk, d_adv, inst = setup(ctx, pkg, plonk, oracle, c)
    opk = PR.keygen(c.desc, c.fixed, c.assembly.mapping, TAU, transcript_repr=123456789)
    want = PR.create_proof(opk, c.instances, c.advice, seed=3)
    n = c.n

    def good():
        assert plonk.create_proof(ctx, pk, inst, d_adv, seed=3) == want

    good()
    too_long = [np.zeros((c.usable + 1, 4), np.uint64)] + list(inst[1:])
    with pytest.raises(pkg.AmdzkError, match="InstanceTooLarge"):
        plonk.create_proof(ctx, pk, too_long, d_adv, seed=3)
    good()
    with pytest.raises(pkg.AmdzkError, match="null"):
        plonk.create_proof(ctx, pk, inst, None, seed=3)
    with pytest.raises(pkg.AmdzkError, match="stride"):
        plonk.create_proof(ctx, pk, inst, d_adv, seed=3, advice_stride=n - 1)
    with pytest.raises(pkg.AmdzkError, match="transcript"):
        plonk.create_proof(ctx, pk, inst, d_adv, seed=3, transcript=7)
    good()
    # raw calls: a null instance column with a non-zero length; buffer too small; length query
    L = ctx.L
    cols = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in inst]
    ptrs = (C.c_void_p * len(cols))(*[x.ctypes.data if x.size else None for x in cols])
    lens = (C.c_size_t * len(cols))(*[x.shape[0] for x in cols])
    need = C.c_size_t(0)
    buf = (C.c_uint8 * (1 << 16))()
    null_ptrs = (C.c_void_p * len(cols))()
    one = (C.c_size_t * len(cols))(*([1] * len(cols)))
    assert L.amdzk_create_proof_ex(ctx.h, pk.h, null_ptrs, one, d_adv.ptr, n, C.c_uint64(3), 0, buf, len(buf), C.byref(need)) != 0
    assert b"null" in L.amdzk_last_error(ctx.h)
    need.value = 0
    assert L.amdzk_create_proof_ex(ctx.h, pk.h, ptrs, lens, d_adv.ptr, n, C.c_uint64(3), 0, buf, 16, C.byref(need)) != 0
    assert b"too small" in L.amdzk_last_error(ctx.h) and need.value == len(want)
    need.value = 0
    assert L.amdzk_create_proof_ex(ctx.h, pk.h, ptrs, lens, d_adv.ptr, n, C.c_uint64(3), 0, None, 0, C.byref(need)) == 0
    assert need.value == len(want) == plonk.proof_size(ctx, pk)
    assert L.amdzk_create_proof_ex(ctx.h, pk.h, ptrs, lens, d_adv.ptr, n, C.c_uint64(3), 0, buf, len(buf), C.byref(need)) == 0
    assert bytes(buf[: need.value]) == want
    d_adv.free(); pk.free(); params.free()


SWITCHES = [
    {"AMDZK_LATENCY_MODE": "0"}, {"AMDZK_LATENCY_COLS": "64"}, {"AMDZK_FOLD_QUAD": "0"}, {"AMDZK_TAIL_QUAD": "0", "AMDZK_TAIL_TREE": "1"},
    {"AMDZK_MSM_NLEV": "2"}, {"AMDZK_L1_LDS": "4"}, {"AMDZK_L1_LDS": "9"}, {"AMDZK_SERIAL": "1"}, {"AMDZK_FULL_COSETS": "1"},
    {"AMDZK_HOST_WAIT": "block"}, {"AMDZK_H_PARTS": "1"}, {"AMDZK_MSM_T1": "32", "AMDZK_MSM_TL": "4"}, {"AMDZK_MSM_PIPELINE": "0"},
    {"AMDZK_NTT_AFTER_L1": "0"}, {"AMDZK_MSM_C": "12"}]


def test_whole_proofs_under_every_documented_switch():
    """INTEGRATION.md's environment switches — the variants kept behind them (latency mode off / wider, one fold level, level 1
    with the accumulator in LDS or as a persistent grid, serial or full-coset default keys, polling waits, one-piece h(X)
    program, other task sizes and window width, no pipelining) prove the oracle's BYTES: whole-proof tests of this file and a
    few random constraint systems, again, in child processes with the switch forced (switches are read once per process).
    Four children at a time (with this process, five on the GPU: the pool allows six)."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    gp, rc = os.path.join(here, "test_gpu_prover.py"), os.path.join(here, "test_random_circuits.py")
    ids = [gp + "::test_proof_bytes_equal_oracle", gp + "::test_rsa_sha256_shape_small_equals_oracle",
           gp + "::test_several_circuit_instances_in_one_proof[7-2-blake2b]"]  # GWC: the random circuits with seed % 4 == 1
    ids += [rc + "::test_device_proof_bytes_equal_oracle_on_random_circuits[%s]" % c for c in ("5-1", "5-9", "6-44", "6-23", "7-85", "8-112", "9-120")]
    ids += [rc + "::test_device_multi_instance_proofs_equal_oracle_on_random_circuits[6-212-3]"]
    pending, running, failed = list(SWITCHES), [], []
    while pending or running:
        while pending and len(running) < 4:
            env = pending.pop(0)
            e = dict(os.environ)
            e.update(env)
            running.append((env, subprocess.Popen([sys.executable, "-m", "pytest", *ids, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], env=e,
                                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        env, proc = running.pop(0)
        try:
            out, _ = proc.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            proc.kill()
            out = "timed out\n" + proc.communicate()[0]
        if proc.returncode != 0 or " passed" not in out or "failed" in out:
            failed.append((env, out[-2000:]))
    assert not failed, "\n\n".join("%r:\n%s" % f for f in failed)
